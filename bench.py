#!/usr/bin/env python3
"""bench.py -- self-play hot path benchmark (BASELINE.json metric: self-play positions/s and MCTS
node-expansions/s at 800 sims on 8x8).

    python bench.py --gpus N --steps K --warmup W          (N=1: plain python; N>1: torchrun)

A "step" = one lockstep MOVE of all G concurrent games = 1 root evaluation + `sims` simulations,
each simulation being one fused HIP tree kernel + one batched CNN forward over the leaves that need
one, then the move itself (root policy, sampling, rules step, game-ended test, slot refill).
Workload = config[1] of BASELINE.json: 8x8, 800 sims, 4096 concurrent games per GPU, frozen random-init
128x10 net (torch.manual_seed(0)); games start at staggered plies from seeded random legal play so the
batch is in the steady state of continuous self-play.  value = positions/s over all ranks.

The headline leg runs the evaluator at the REFERENCE's precision: `--nn f16x3` = the split-f16 tower
(csrc/yy_tower_g.hip + yy_fc_heads.hip: float32-accurate, identical visit counts to the reference's CPU float32 search on
every golden root at 6x6, 8x8, 12x12 and three non-default shapes -- tests/test_gpu_mcts.py::test_live_gpu_evaluator_search_vs_reference_pi).  At N=1 one more
leg is timed in the same run and reported as an extra key of the same JSON line: `secondary` = the bf16
tower (reduced precision: NOT the headline); `--oversubscribe G2` adds a leg with G2 concurrent games (`oversubscribed`);
`with_evaluation_reuse` is the same workload with the engine's evaluation reuse on (off in the headline).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: bf16 / f16 MFMA ~2.5 PFLOP/s dense
MFMA_F32_PEAK_TFLOPS = 157.3    # same guide: f32-input MFMA = the f32 vector rate
NN_MODES = ["f16x3", "f16x3r", "bf16", "fp16", "fp32", "fp32t"]
FP32_GRADE = {"f16x3": "float32-accurate split-f16 MFMA (hi + lo float16 pairs = 22 significant bits, f32 accumulate)",
              "f16x3r": "float32-accurate split-f16 MFMA, round-2 32x32x16 tower kernel (A/B partner)",
              "fp32t": "exact float32 MFMA", "fp32": "float32 (PyTorch/MIOpen)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--cols", type=int, default=8)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--nn", default="f16x3", choices=NN_MODES,
                    help="evaluator of the headline leg; f16x3 = float32-accurate (the reference evaluates in float32)")
    ap.add_argument("--secondary-nn", default="bf16", choices=NN_MODES + ["none"],
                    help="N=1 only: a second, shorter leg with this evaluator, reported under `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--oversubscribe", type=int, default=0,
                    help="N=1 only: a third leg with this many concurrent games (compacted leaf batch ~ whole workgroup "
                         "rounds); 0 (default) = skip: measured neutral, the launch time is proportional to the live rows")
    ap.add_argument("--oversubscribe-steps", type=int, default=3)
    ap.add_argument("--reuse-evaluations", type=int, default=0,
                    help="1 = the main leg runs with the engine's evaluation reuse (YY_FLAG_REUSE_PASS_VALUE | "
                         "YY_FLAG_REUSE_TRANSPOSITIONS | YY_FLAG_KEEP_EVALUATIONS); default 0: the evaluator is given every row the reference evaluates")
    ap.add_argument("--book-stones", type=int, default=8,
                    help="legs with evaluation reuse also get a shared opening book: every position reachable with at most this "
                         "many stones, evaluated once before the timed region (0 = none; boards of at most 64 cells)")
    ap.add_argument("--reuse-steps", type=int, default=10, help="steps of the extra leg with evaluation reuse on (0 = skip)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --games concurrent games PER GPU; strong: --games in total, split over the ranks "
                         "(SURVEY 8d config 3: 8 x 512)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="HIP streams the rank's games are cut over (self_play.SelfPlayLanes): one lane's evaluator launch fills the "
                         "compute units the other lane's partly empty last round of workgroups leaves idle; 1 = one lockstep batch")
    ap.add_argument("--split-wg", type=int, default=0,
                    help="experiment: live-row count (in small-form workgroups) up to which a compacted evaluator launch takes the "
                         "one-board tower form (network.G_SPLIT_WG; 0 = the shipped value)")
    ap.add_argument("--no-form-hint", action="store_true",
                    help="experiment: never tell the evaluator the rows per step (it then enqueues both tower forms, device-gated)")
    ap.add_argument("--semantics", default="copied", choices=["copied", "aliased"])
    ap.add_argument("--quirks", action="store_true", help="reference_quirks (Q4/Q5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline sample: half per board semantics")
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU baseline processes (0 = one per host core, at most 16)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="run only the CPU baseline leg (no GPU needed) and print it")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank control flow with several ranks sharing ONE GPU (RCCL needs a "
                         "device per rank); the exchange then goes through host memory")
    return ap.parse_args()


def stagger_start(eng, seed):
    """Advance game g by (g mod 48) uniformly random legal plies with the HIP rules kernels, so the
    batch holds openings, middle games and endings like continuous self-play does."""
    if hasattr(eng, "lanes"):                  # SelfPlayLanes: every lane, with its own seed and its own global game indices
        K = len(eng.lanes)
        for k, lane in enumerate(eng.lanes):
            with lane._on_stream():
                stagger_start(lane, seed + 7919 * k)
                lane.game_id = lane.game_id * K + k
        torch.cuda.synchronize()
        return
    from yinyang_game_alphazero_amd import engine as E
    G, dev = eng.G, eng.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    target = (torch.arange(G, device=dev) * 7919) % 48
    boards = torch.zeros((G, eng.R, eng.C), dtype=torch.int8, device=dev)
    players = torch.ones(G, dtype=torch.int8, device=dev)
    ply = torch.zeros(G, dtype=torch.int32, device=dev)
    for _ in range(2 * 48):
        mask = E.valid_mask(boards, players).to(torch.float32)
        has = mask.sum(1) > 0
        adv = (ply < target) & has
        nomove = (ply < target) & ~has
        players = torch.where(nomove, -players, players).contiguous()          # pass
        dist = torch.where(adv[:, None], mask, torch.full_like(mask, 1.0))
        act = torch.multinomial(dist, 1, generator=gen).reshape(-1).to(torch.int32)
        act = torch.where(adv, act, torch.full_like(act, -1)).contiguous()
        old = players.clone()
        E.step_(boards, players, act)
        players = torch.where(adv, players, old).contiguous()
        ply += adv.to(torch.int32)
    ended = E.game_ended(boards, players) != 0
    boards[ended] = 0
    players[ended] = 1
    ply[ended] = 0
    eng.boards, eng.players, eng.ply = boards, players.contiguous(), ply
    eng.alive[:] = True
    eng.n_alive = G
    eng.n_ex[:] = 0
    eng.game_id = torch.arange(G, device=dev, dtype=torch.int64)
    eng.games_started = G
    eng.games_target = 1 << 60          # refill forever


class HipEventTimer:
    """HIP events (hipEventRecord on the stream the kernel is launched on, via the HIP runtime itself,
    not torch.cuda.Event) around every launch of a kernel."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))
        self.i = 0

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def start(self):
        if self.i < len(self.pairs):
            assert self.hip.hipEventRecord(self.pairs[self.i][0], self._stream()) == 0

    def stop(self):
        if self.i < len(self.pairs):
            assert self.hip.hipEventRecord(self.pairs[self.i][1], self._stream()) == 0
            self.i += 1

    def mean_ms(self):
        torch.cuda.synchronize()
        ms, tot = ctypes.c_float(0), 0.0
        for a, b in self.pairs[: self.i]:
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            tot += ms.value
        for a, b in self.pairs:
            self.hip.hipEventDestroy(a)
            self.hip.hipEventDestroy(b)
        return tot / max(self.i, 1), self.i


def tower_launcher(eng, planes):
    """(callable launching the evaluator's dominant kernel ALONE on the leaf batch `planes`, kernel name, algorithmic FLOPs per
    launch, MFMA peak it is priced against) or None when the evaluator has no hand-written tower."""
    from yinyang_game_alphazero_amd import engine as E
    ev, G = eng.evaluator, planes.shape[0]
    cells, blocks = eng.R * eng.C, len(ev.net.res_blocks)
    body = 2 * blocks * (2 * 9 * 128 * 128 * cells)
    heads = 2 * 128 * 64 * cells
    mode = getattr(ev, "mode", "")
    if mode == "f16x3":
        stem = 2 * 9 * 5 * ev.net.conv1.out_channels * cells
        ch = ev.net.conv1.out_channels
        body = 2 * blocks * (2 * 9 * ch * ch * cells)
        heads = 2 * ch * 64 * cells
        if ev.use_h3r:
            form = {8: "k_tower_h3r<8,2,9>", 6: "k_tower_h3r<6,4,3>", 12: "k_tower_h3r<12,1,3>"}[eng.R] + " (round-2 32x32x16 kernel, A/B partner)"

            def launch(rows=None, n_rows=None):
                return E.tower_heads_forward_h3r(planes, ev.h3r_w, ev.h3r_hw, ev.h3_b, ev.h3_layers, ev.h3_exps, rows, n_rows)
        else:
            nb, tb = ev.g_big if G > ev.g_split else ev.g_small
            form = "k_tower_g<%d,%d,9> (%d boards = %d of %d columns per workgroup, v_mfma_f32_16x16x32_f16, wave = 32 output channels, weight stream in registers)" % (
                ch // 32, nb, tb, tb * cells, 16 * nb)

            def launch(rows=None, n_rows=None):
                return E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb, rows, n_rows)
        return (launch,
                form + ": stem + residual tower + 1x1 head convs, split-f16 (3 f16 MFMAs per product term, float32-accurate); "
                       "peak = f16 MFMA dense peak / 3 (nominal, 2.4 GHz; what this board sustains for bare f16 MFMA issue on random "
                       "operands: profiles/r03_mfma_ceiling.json)",
                (stem + body + heads) * G, MFMA_BF16_PEAK_TFLOPS / 3)
    if mode == "fp32t":
        return (lambda: E.tower_forward_f32(planes, ev.f32_w, ev.f32_b, ev.f32_layers),
                "k_tower_f32: stem + residual tower, exact f32 MFMA 32x32x2", (2 * 9 * 8 * 128 * cells + body) * G,
                MFMA_F32_PEAK_TFLOPS)
    if getattr(ev, "tower", False):
        name = {6: "k_towerq<6,8>", 8: "k_tower" if G > 512 else "k_towerq<8,%d>" % (1 if G <= 256 else 2),
                12: "k_towerq<12,2>"}.get(eng.R, "k_tower")
        if getattr(ev, "fused_heads", False):
            return (lambda: E.tower_heads_forward(planes, ev.towerh_w, ev.towerh_b, ev.tower_layers),
                    name + ": stem + residual tower + 1x1 head convs, bf16 MFMA, activations LDS-resident",
                    (2 * 9 * 16 * 128 * cells + body + heads) * G, MFMA_BF16_PEAK_TFLOPS)
        return (lambda: E.tower_forward(planes, ev.tower_w, ev.tower_b, ev.tower_layers),
                name + ": stem + residual tower, bf16 MFMA", (2 * 9 * 16 * 128 * cells + body) * G, MFMA_BF16_PEAK_TFLOPS)
    return None


def roofline_pass(eng):
    """One more move of the SAME workload, launched eagerly (same kernels as the graph replays) with HIP events around every
    fused tree-kernel launch of ONE lane; returns (mean kernel ms, launches, that lane's device counters of exactly those
    launches, mean evaluator-forward ms, move ms, mean dominant-kernel ms or None, live launch or None, lane games).  The
    evaluator and its dominant kernel are timed on the leaf batch of ALL lanes together (= the configuration's G rows)."""
    lanes = getattr(eng, "lanes", [eng])
    lane = lanes[0]
    torch.cuda.synchronize()
    lane.search.use_graph, lane.search.graph = False, None
    lane.ctx.reset_counters()
    timer = HipEventTimer(lane.sims)
    lane.search.timer = timer
    with lane._on_stream():
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        lane.play_move()
        t1.record()
    lane.search.timer = None
    k_ms, n = timer.mean_ms()
    counters = lane.ctx.status()
    move_ms = t0.elapsed_time(t1)
    planes = torch.cat([l.ctx.planes for l in lanes]).contiguous()
    needs = torch.cat([l.ctx.needs_eval for l in lanes]).contiguous()
    # evaluator forward alone (every row), the whole leaf batch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        eng.evaluator(planes)
    e1.record()
    torch.cuda.synchronize()
    tower_ms, live = None, None
    tl = tower_launcher(eng, planes)
    if tl is not None:      # the dominant kernel alone: HIP events on its launch stream
        tt = HipEventTimer(reps)
        for _ in range(reps):
            tt.start()
            tl[0]()
            tt.stop()
        tower_ms, _ = tt.mean_ms()
        if getattr(eng.evaluator, "supports_compaction", False):
            # the same kernel on only the rows of the pending leaf batch that need an evaluation
            from yinyang_game_alphazero_amd import engine as E
            rows, n_rows = E.compact_rows(needs)
            tt = HipEventTimer(reps)
            for _ in range(reps):
                tt.start()
                tl[0](rows, n_rows)
                tt.stop()
            live = (tt.mean_ms()[0], int(n_rows.item()))
            if len(lanes) > 1:      # ... and on ONE lane's compacted rows: the launch a kernel trace of the timed steps lists
                tl1 = tower_launcher(eng, lane.ctx.planes)
                rows1, n1 = E.compact_rows(lane.ctx.needs_eval)
                tt = HipEventTimer(reps)
                for _ in range(reps):
                    tt.start()
                    tl1[0](rows1, n1)
                    tt.stop()
                live = live + (tt.mean_ms()[0], int(n1.item()))
    return k_ms, n, counters, e0.elapsed_time(e1) / reps, move_ms, tower_ms, live, lane.G, planes


def algorithmic_bytes(counters, G, A, n_steps, nw):
    """bytes = 16*sum_levels(k) + 2*Bb + Mb + 8 + 20*A + 4*A + 4 + 16*k_leaf
    + 16*(d+1) + 32 per expansion, from the measured device counters."""
    ev = max(counters["evals"], 1)
    Bb, Mb = 16 * nw, 8 * nw
    per_exp = 2 * Bb + Mb + 8 + 20 * A + 4 * A + 4 + 32
    total = (16 * counters["children_scanned"] + per_exp * counters["evals"] + 16 * counters["children_created"]
             + 16 * (counters["levels"] + n_steps * G))
    return total / n_steps, dict(mean_children_scanned=counters["children_scanned"] / (n_steps * G),
                                 mean_depth=counters["levels"] / (n_steps * G),
                                 mean_k_leaf=counters["children_created"] / ev,
                                 eval_fraction=counters["evals"] / (n_steps * G),
                                 bytes_per_expansion=total / ev)


# ------------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(job):
    """One host process = one core: the CPU restatement's tree (oracle/yy_oracle.c: same float32 PUCT) driving the same
    network under PyTorch CPU at batch 1 like the reference (self_play.py:54-59), ONE `sims`-simulation search from the
    empty board with root noise from the per-game seed 1000 + game index; evaluations are counted until the deadline, after
    which the callback stops calling the network so the process ends quickly."""
    idx, rows, cols, channels, blocks, sims, copied, seconds = job
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from yinyang_game_alphazero_amd.game import YinYangGame
    from yinyang_game_alphazero_amd.network import YinYangNeuralNetwork
    torch.set_num_threads(1)
    torch.manual_seed(0)
    net = YinYangNeuralNetwork(YinYangGame(rows, cols), channels, blocks).eval()
    A = rows * cols
    rng = np.random.RandomState(1000 + idx)
    noise = rng.dirichlet([0.3] * A)            # every cell is legal on the empty board
    state = dict(n=0, t0=None, dt=0.0, last=None)

    def predict(board):
        now = time.perf_counter()
        if state["t0"] is None:
            state["t0"] = now
        if now - state["t0"] >= seconds and state["last"] is not None:
            return state["last"]                # past the deadline: not counted, not evaluated
        x = torch.from_numpy(O.encode_planes(np.ascontiguousarray(board)[None]))
        with torch.no_grad():
            logits, v = net(x)
            p = torch.softmax(logits, 1)
        state["last"] = (p[0].numpy(), float(v[0, 0]))
        state["n"] += 1
        state["dt"] = time.perf_counter() - state["t0"]
        return state["last"]

    O.search_callback(np.zeros((rows, cols), np.int8), 1, sims, copied, predict, noise=noise)
    return state["n"], state["dt"]


def cpu_baseline(args):
    import multiprocessing as mp
    cores = args.cpu_workers or min(16, os.cpu_count() or 1)     # a 1-GPU box's share of the host is 16 cores
    half = args.cpu_seconds / 2
    out = {}
    ctx = mp.get_context("spawn")                                # the parent holds a GPU context: never fork it
    with ctx.Pool(cores) as pool:
        for name, copied in (("copied", 1), ("aliased", 0)):
            jobs = [(i, args.rows, args.cols, args.channels, args.blocks, args.sims, copied, half) for i in range(cores)]
            res = pool.map(_cpu_worker, jobs, chunksize=1)
            out[name] = sum(n / max(dt, 1e-9) for n, dt in res)
    v = out["copied"]
    return {"value": v, "unit": "expansions/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes (one per host core, 1 torch thread each), each the first {half:.0f} s of one "
                      f"{args.sims}-simulation search from the empty {args.rows}x{args.cols} board (per-game seed 1000+i, root "
                      f"noise), batch-1 fp32 {args.channels}x{args.blocks} net on torch CPU, oracle/yy_oracle.c tree; value = "
                      f"copied boards, aliased (literal reference semantics) alongside",
            "aliased_expansions_per_s": out["aliased"],
            "positions_per_s": v / (args.sims + 1),
            "python_reference_anchor": "BASELINE.md section 2: 103 (copied) / 117 (aliased) simulations/s per 8-core process"}


# ------------------------------------------------------------------------------------------- one timed leg
def timed_region(eng, steps, warmup, rank, world, dist, cdev, sims, capacity=None):
    """The contract's timed region for one engine: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier +
    torch.cuda.synchronize() on both sides, the MAX of the elapsed time over ranks, the SUM of positions / evaluator rows
    over ranks, and the path's single exchange (all-gather of the examples produced in the region) after it.  `eng` needs
    play_move() -> positions, collect() -> example dict, ctx.status() / ctx.reset_counters(), G (tests drive this with a
    stand-in engine under gloo, world 2)."""
    from yinyang_game_alphazero_amd.self_play import gather_examples
    on_gpu = torch.cuda.is_available()

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(warmup):
        eng.play_move()
    eng.collect()
    eng.ctx.reset_counters()
    barrier()
    t0 = time.perf_counter()
    positions, marks = 0, [t0]
    for _ in range(steps):
        positions += eng.play_move()
        marks.append(time.perf_counter())           # a move ends with the lanes' host reads: wall time per step, no extra sync
    barrier()
    dt = time.perf_counter() - t0
    counters = eng.ctx.status()
    ex = eng.collect()
    # the single exchange of the path: all-gather the examples produced in the timed region
    tg0 = time.perf_counter()
    dev0 = ex["states"].device
    ex_all = gather_examples(ex if cdev == dev0 else {k: v.to(cdev) for k, v in ex.items()}, capacity=capacity)
    if on_gpu:
        torch.cuda.synchronize()
    gather_s = time.perf_counter() - tg0
    served = counters.get("reused_values", 0) + counters.get("transposition_hits", 0)      # expansions that needed no evaluator row
    tot = torch.tensor([float(positions), float(counters["evals"]), dt, float(served)], dtype=torch.float64, device=cdev)
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx[2])
    return dict(games=eng.G, steps=steps, dt=dt, positions=float(tot[0]), evals=float(tot[1]), served=float(tot[3]), gather_s=gather_s,
                examples=int(ex_all["states"].shape[0]), sims_total=steps * sims * eng.G * world, warmup=warmup,
                step_ms=[round((b - a) * 1e3, 1) for a, b in zip(marks[:-1], marks[1:])],
                eval_fraction=counters["evals"] / max(steps * sims * eng.G, 1))


def run_leg(args, nn, games, steps, warmup, rank, world, dist, cdev, with_roofline, reuse=None):
    import yinyang_game_alphazero_amd as pkg
    from yinyang_game_alphazero_amd.self_play import SelfPlayLanes, example_capacity
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    game = pkg.YinYangGame(args.rows, args.cols)
    net = pkg.YinYangNeuralNetwork(game, args.channels, args.blocks).to(dev).eval()
    evaluator = pkg.BatchedEvaluator(net, nn)
    use_reuse = bool(args.reuse_evaluations if reuse is None else reuse)
    book, book_s = None, 0.0
    if use_reuse and args.book_stones > 0 and args.rows * args.cols <= 64 and getattr(evaluator, "row_independent", False):
        torch.cuda.synchronize()
        tb = time.perf_counter()
        book = pkg.engine.OpeningBook(args.rows, args.cols, evaluator, args.book_stones, device=dev)
        torch.cuda.synchronize()
        book_s = time.perf_counter() - tb
    eng = SelfPlayLanes(game, evaluator, num_simulations=args.sims, concurrent_games=games, opening_book=book, lanes=args.lanes,
                         board_semantics=args.semantics, reference_quirks=args.quirks,
                         use_graph=not args.no_graph, seed=1000, device=dev,
                         first_game_index=rank, game_index_stride=world,
                         reuse_pass_value=bool(args.reuse_evaluations if reuse is None else reuse),
                         reuse_transpositions=bool(args.reuse_evaluations if reuse is None else reuse),
                         keep_evaluations=bool(args.reuse_evaluations if reuse is None else reuse))
    stagger_start(eng, 4242 + rank)
    leg = timed_region(eng, steps, warmup, rank, world, dist, cdev, args.sims,
                       capacity=example_capacity(games * world, world, eng.T))
    leg["nn"], leg["reuse"] = nn, eng.reuse_pass_value or eng.reuse_transpositions
    if book is not None:
        leg["book"] = dict(max_stones=book.max_stones, positions=book.n, stored=book.stored, bytes=book.bytes, build_s=book_s,
                           note="evaluated once with the same evaluator before the timed region; shared by all games")
    if with_roofline and rank == 0:
        leg["roofline"] = make_roofline(args, eng, games)
    eng.close()
    del eng, evaluator, net
    torch.cuda.empty_cache()
    return leg


def pmc_traffic(name, key):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (collected OFFLINE in separate passes, corrected as
    MI355X_MICROARCH.md prescribes) when its recorded configuration equals this run's; else (None, why)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"no offline PMC summary profiles/{name}"
    rec = json.load(open(path))
    if rec.get("config") != key:
        return None, f"profiles/{name} was collected for {rec.get('config')}, not for this configuration"
    return rec["per_launch_bytes"]["traffic_corrected"], f"profiles/{name} (offline rocprofv3 --pmc passes, not measured in this run)"


def make_roofline(args, eng, games):
    k_ms, n_launch, kc, nn_ms, eager_move_ms, tower_ms, live, lane_games, planes = roofline_pass(eng)
    A = args.rows * args.cols
    nw = (A + 63) // 64
    # the counters cover one lane's whole move = n_launch + 1 selections (the first one runs before the timed fused steps)
    bytes_per_launch, shape = algorithmic_bytes(kc, lane_games, A, n_launch + 1, nw)
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    tkey = dict(games=games, rows=args.rows, cols=args.cols, channels=args.channels, blocks=args.blocks)   # the tower launch
    key = dict(tkey, games=lane_games, sims=args.sims, semantics=args.semantics)                           # the tree kernel (one lane's launch)
    traffic, src = pmc_traffic("r03_k_mcts_pmc.json", key)
    roof_tree = {"bound": "hbm", "kernel": "k_mcts (fused expand+backup+select+rules+encode), one lane of %d games per launch" % lane_games,
                 "achieved": achieved,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                 "traffic_source": src, "avg_launch_ms": k_ms, "launches_timed": n_launch,
                 "algorithmic_bytes_per_launch": bytes_per_launch, **shape}
    out = {"roofline": roof_tree}
    tl = tower_launcher(eng, planes)
    if tl is not None and tower_ms is not None:
        _, name, flops, peak = tl
        ach = flops / (tower_ms * 1e-3) / 1e12
        mode = getattr(eng.evaluator, "mode", "")
        traffic, src = pmc_traffic(f"r03_k_tower_{mode}_hbm_pmc.json", tkey)
        out = {"roofline": {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                            "frac": ach / peak, "traffic": traffic, "traffic_source": src, "avg_launch_ms": tower_ms,
                            "algorithmic_flops_per_launch": flops,
                            "timed": "20 launches on every row of the live leaf batch, hipEventRecord on the launch stream"},
               "roofline_tree_kernel": roof_tree}
        if live is not None:      # what the lockstep step launches (and what a rocprofv3 summary of this command averages over)
            lms, lrows = live[:2]
            lach = flops / games * lrows / (lms * 1e-3) / 1e12
            out["roofline"].update(live_rows=lrows, live_launch_ms=lms, live_achieved=lach, live_frac=lach / peak,
                                   live_note="the same kernel on the compacted rows of the pending leaf batch of all lanes (rows that "
                                             "need an evaluation); fewer whole rounds of workgroups than the dense launch")
            if len(live) == 4:
                out["roofline"].update(lane_live_rows=live[3], lane_live_launch_ms=live[2],
                                       lane_live_achieved=flops / games * live[3] / (live[2] * 1e-3) / 1e12,
                                       lane_note="ONE lane's compacted launch alone on the chip: what the timed steps enqueue (two lanes, "
                                                 "two streams); in a kernel trace of the bench command its duration is longer, because "
                                                 "the two lanes' launches overlap and share the compute units")
    Cc = args.channels
    flops_leaf = (2 * 9 * 5 * Cc * A + 2 * args.blocks * (2 * 9 * Cc * Cc * A) + 2 * (2 * Cc * 32 * A)
                  + 2 * 32 * A * A + 2 * 32 * A * 256 + 512)
    out.update(nn_forward_ms=nn_ms, nn_tflops=flops_leaf * games / (nn_ms * 1e-3) / 1e12, tree_kernel_ms=k_ms,
               tree_kernel_only_expansions_per_s=kc["evals"] / max(n_launch, 1) / (k_ms * 1e-3), eager_move_ms=eager_move_ms)
    return out


def leg_summary(leg, note):
    return {"nn": leg["nn"], "games_per_gpu": leg["games"], "steps": leg["steps"], "value": leg["positions"] / leg["dt"],
            "unit": "positions/s", "expansions_per_s": leg["evals"] / leg["dt"], "simulations_per_s": leg["sims_total"] / leg["dt"],
            "ms_per_step": leg["dt"] / leg["steps"] * 1e3, "eval_fraction": leg["eval_fraction"],
            "warmup": leg.get("warmup"), "step_ms": leg.get("step_ms"), "note": note}


def result_line(args, main_leg, world, extra=None, cpub=None):
    """The ONE JSON line of the contract, from the main leg's totals (already summed / maxed over ranks)."""
    roof = main_leg.pop("roofline", {"roofline": None})
    dt = main_leg["dt"]
    return {
        "metric": f"self-play positions/sec (+ MCTS node-expansions/sec) at {args.sims} sims, {args.rows}x{args.cols} board",
        "value": main_leg["positions"] / dt, "unit": "positions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": getattr(args, "scaling", "weak"), "vs_baseline": None, "dtype": args.nn,
        "dtype_note": FP32_GRADE.get(args.nn, "reduced-precision evaluator (the reference evaluates in float32)"),
        "data": "synthetic",
        "expansions_per_s": main_leg["evals"] / dt, "simulations_per_s": main_leg["sims_total"] / dt,
        "config": {"workload": f"{args.rows}x{args.cols} board, {args.sims} sims/move, {args.games} concurrent games "
                               f"{'per GPU' if getattr(args, 'scaling', 'weak') == 'weak' else 'in total (split over the ranks)'}, frozen random-init {args.channels}x{args.blocks} net (seed 0), "
                               f"{args.semantics} boards, reference_quirks={args.quirks}, staggered start plies",
                   "tree_arithmetic": "f32 PUCT + u64 bitboards", "nn_dtype": args.nn,
                   "evaluation_reuse": bool(main_leg.get("reuse", False)),
                   "parallelism": f"episode-sharded x{world}", "lanes_per_gpu": getattr(args, "lanes", 1), "hipgraph": not args.no_graph},
        "cpu_baseline": cpub, "gather_s": main_leg["gather_s"], "examples_gathered": main_leg["examples"],
        **roof, **(extra or {}),
    }


def main():
    args = parse()
    if args.no_form_hint:
        from yinyang_game_alphazero_amd import network as _net
        _net.BatchedEvaluator.rows_hint = lambda self, owner, mean_rows: None
    if args.split_wg > 0:          # (how network.G_SPLIT_WG was chosen)
        from yinyang_game_alphazero_amd import network as _net
        _net.G_SPLIT_WG = args.split_wg
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (the product path has no CPU fallback)"
    if args.dist_backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torchrun the RCCL path is exercised even at N=1
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is reserved for the
        # one JSON line, so park fd 1 on stderr while the communicator is created (eager with device_id) and exercised once
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                dist.barrier()
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo")
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    dev = torch.device("cuda", local)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")      # where collectives run

    if args.scaling == "strong":
        assert args.games % world == 0, "--scaling strong: --games must divide by the number of ranks"
    per_rank = args.games if args.scaling == "weak" else args.games // world
    main_leg = run_leg(args, args.nn, per_rank, args.steps, args.warmup, rank, world, dist, cdev, with_roofline=True)
    extra = {}
    default_workload = (args.games, args.rows, args.cols, args.sims, args.channels, args.blocks) == (4096, 8, 8, 800, 128, 10)
    if world == 1:
        if args.secondary_nn not in ("none", args.nn):
            leg = run_leg(args, args.secondary_nn, args.games, args.secondary_steps, 1, rank, world, dist, cdev, False)
            extra["secondary"] = leg_summary(leg, "same workload, %s evaluator%s" % (
                args.secondary_nn, "" if args.secondary_nn in FP32_GRADE else
                ": REDUCED precision against the reference's float32 (not the headline; parity figures in "
                "tests/test_gpu_mcts.py::test_live_gpu_evaluator_search_vs_reference_pi)"))
        if args.reuse_steps > 0 and not args.reuse_evaluations and args.semantics == "copied":
            leg = run_leg(args, args.nn, args.games, args.reuse_steps, max(1, args.warmup), rank, world, dist, cdev, False, reuse=True)
            s = leg_summary(leg, "same workload and evaluator with the engine's evaluation reuse ON (the SelfPlayEngine default for this "
                                 "evaluator; OFF in the headline so that the evaluator sees the reference's rows one for one): a node "
                                 "without legal moves is evaluated once instead of on every visit (ai/mcts.py:93-95), and a leaf whose "
                                 "position this search or an earlier search of the same game has evaluated takes the cached policy row "
                                 "and value (:385-397); the games are the same move for move (tests/test_gpu_selfplay.py::test_evaluation_reuse_plays_the_same_games). "
                                 "The games enter at staggered plies with EMPTY evaluation caches, which fill during the first moves: the leg takes "
                                 "the main leg's W untimed warm-up moves, step_ms lists the timed moves one by one, and the whole-run figure with "
                                 "every transient inside (book build, cold caches, the draining tail) is profiles/r03_config2_full_run_evaluation_reuse_book.json")
            s["evaluator_rows_per_s"] = s.pop("expansions_per_s")      # rows really evaluated
            s["opening_book"] = leg.get("book")
            if leg.get("book"):
                # the book is evaluated once per network and serves a whole self-play run: config 2 plays its 4096 games to the
                # end in ~53 steps, so these `steps` carry steps/53 of the build; the other two readings are alongside
                b = leg["book"]["build_s"]
                run_steps = 53.0
                s["value_excluding_book_build"] = s["value"]
                s["value_book_build_inside_these_steps"] = leg["positions"] / (leg["dt"] + b)
                s["value"] = leg["positions"] / (leg["dt"] + b * leg["steps"] / run_steps)
                s["value_note"] = ("positions/s with the opening book's build time charged pro rata: a book is built once per network and "
                                   "serves a whole run (config 2 = ~%d steps); these %d steps carry %d/%d of its %.2f s" % (
                                       run_steps, leg["steps"], leg["steps"], run_steps, b))
            # tree nodes expanded (Node.expand calls of the reference, mcts.py:397): by an evaluator row or from the cache
            s["node_expansions_per_s"] = (leg["evals"] + leg["served"]) / leg["dt"]
            extra["with_evaluation_reuse"] = s
            extra["value_with_evaluation_reuse"] = s["value"]          # positions/s of the engine's default configuration
        over = max(args.oversubscribe, 0)
        if over > 0:
            leg = run_leg(args, args.nn, over, args.oversubscribe_steps, 1, rank, world, dist, cdev, False)
            extra["oversubscribed"] = leg_summary(leg, f"headline evaluator with {over} concurrent games: the leaves that need "
                                                       "an evaluation (terminal revisits do not: 5-7 % of the simulations) are compacted, so the "
                                                       "launch carries ~3950-4050 live rows = just under 8 whole rounds of 2-board workgroups "
                                                       "on 256 CUs")
    cpub = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:      # the CPU leg is reported at N=1 only
        cpub = cpu_baseline(args)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result_line(args, main_leg, world, extra, cpub)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

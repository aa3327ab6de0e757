#!/usr/bin/env python3
"""bench.py -- self-play hot path benchmark (BASELINE.json metric: self-play positions/s and MCTS
node-expansions/s at 800 sims on 8x8).

    python bench.py --gpus N --steps K --warmup W          (N=1: plain python; N>1: torchrun)

A "step" = one lockstep MOVE of all G concurrent games = 1 root evaluation + `sims` simulations,
each simulation being one fused HIP tree kernel + one batched CNN forward over G leaves, then the
move itself (root policy, sampling, rules step, game-ended test, slot refill).  Workload = config[1]
of BASELINE.json: 8x8, 800 sims, 4096 concurrent games per GPU, frozen random-init 128x10 net
(torch.manual_seed(0)); games start at staggered plies from seeded random legal play so the batch
is in the steady state of continuous self-play.  value = positions/s over all ranks.
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: bf16 MFMA ~2.5 PFLOP/s dense
MFMA_F32_PEAK_TFLOPS = 157.3    # same guide: f32-input MFMA = the f32 vector rate


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
    ap.add_argument("--nn", default="bf16", choices=["bf16", "fp16", "fp32", "fp32t", "bf16x3"])
    ap.add_argument("--semantics", default="copied", choices=["copied", "aliased"])
    ap.add_argument("--quirks", action="store_true", help="reference_quirks (Q4/Q5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank control flow with several ranks sharing ONE GPU (RCCL needs a "
                         "device per rank); the exchange then goes through host memory")
    return ap.parse_args()


def stagger_start(eng, seed):
    """Advance game g by (g mod 48) uniformly random legal plies with the HIP rules kernels, so the
    batch holds openings, middle games and endings like continuous self-play does."""
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
    eng.n_ex[:] = 0
    eng.game_id = torch.arange(G, device=dev, dtype=torch.int64)
    eng.games_started = G
    eng.games_target = 1 << 60          # refill forever


class HipEventTimer:
    """HIP events (hipEventRecord on the stream the kernel is launched on, via the HIP runtime itself,
    not torch.cuda.Event) around every launch of the fused tree kernel."""

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


def roofline_pass(eng):
    """One more move of the SAME workload, launched eagerly (same kernels as the graph replays) with HIP
    events around every fused tree-kernel launch; returns (mean kernel ms, launches, device counters of
    exactly those launches, mean evaluator-forward ms)."""
    eng.search.use_graph, eng.search.graph = False, None
    eng.ctx.reset_counters()
    timer = HipEventTimer(eng.sims)
    eng.search.timer = timer
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    eng.play_move()
    t1.record()
    eng.search.timer = None
    k_ms, n = timer.mean_ms()
    counters = eng.ctx.status()
    move_ms = t0.elapsed_time(t1)
    # evaluator forward alone, same batch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        eng.evaluator(eng.ctx.planes)
    e1.record()
    torch.cuda.synchronize()
    # the tower kernel alone (HIP events on its launch stream), when the evaluator uses it
    tower_ms = None
    ev = eng.evaluator
    if getattr(ev, "mode", "") in ("fp32t", "bf16x3"):
        from yinyang_game_alphazero_amd import engine as E
        tower = E.tower_forward_f32 if ev.mode == "fp32t" else E.tower_forward_x3
        tt = HipEventTimer(5)
        for _ in range(5):
            tt.start()
            tower(eng.ctx.planes, ev.f32_w, ev.f32_b, ev.f32_layers)
            tt.stop()
        tower_ms, _ = tt.mean_ms()
    if getattr(ev, "tower", False):
        from yinyang_game_alphazero_amd import engine as E
        tt = HipEventTimer(reps)
        for _ in range(reps):
            tt.start()
            if getattr(ev, "fused_heads", False):
                E.tower_heads_forward(eng.ctx.planes, ev.towerh_w, ev.towerh_b, ev.tower_layers)
            else:
                E.tower_forward(eng.ctx.planes, ev.tower_w, ev.tower_b, ev.tower_layers)
            tt.stop()
        tower_ms, _ = tt.mean_ms()
    return k_ms, n, counters, e0.elapsed_time(e1) / reps, move_ms, tower_ms


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


def cpu_baseline(args):
    """The CPU restatement (oracle/yy_oracle.c: same float32 PUCT, copied semantics) driving the same
    network under PyTorch CPU, batch 1 like the reference (self_play.py:54-59), timed on this host for
    a bounded sample: repeated `sims`-simulation searches from the empty board until ~cpu_seconds."""
    import oracle_lib as O
    from yinyang_game_alphazero_amd.game import YinYangGame
    from yinyang_game_alphazero_amd.network import YinYangNeuralNetwork
    torch.manual_seed(0)
    net = YinYangNeuralNetwork(YinYangGame(args.rows, args.cols), args.channels, args.blocks).eval()
    # the GPU box gives one GPU's share of the host: 16 cores (intra-op threads beyond that only add
    # contention for batch-1 convolutions)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cores = torch.get_num_threads()
    enc = O.encode_planes

    def predict(board):
        x = torch.from_numpy(enc(np.ascontiguousarray(board)[None]))
        with torch.no_grad():
            logits, v = net(x)
            p = torch.softmax(logits, 1)
        return p[0].numpy(), float(v[0, 0])

    sims = min(args.sims, 200)
    t0 = time.perf_counter()
    evals = 0
    n = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        r = O.search_callback(np.zeros((args.rows, args.cols), np.int8), 1, sims, 1, predict)
        evals += r.n_evals + 1
        n += 1
    dt = time.perf_counter() - t0
    return {"value": evals / dt, "unit": "expansions/s", "cores": cores, "kind": "port",
            "sample": f"{n} searches x {sims} sims from the empty {args.rows}x{args.cols} board, batch-1 fp32 "
                      f"{args.channels}x{args.blocks} net on torch CPU ({cores} threads), oracle/yy_oracle.c tree",
            "positions_per_s": evals / dt / (args.sims + 1)}


def main():
    args = parse()
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
    import yinyang_game_alphazero_amd as pkg
    from yinyang_game_alphazero_amd.self_play import SelfPlayEngine, gather_examples

    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    game = pkg.YinYangGame(args.rows, args.cols)
    net = pkg.YinYangNeuralNetwork(game, args.channels, args.blocks).to(dev).eval()
    evaluator = pkg.BatchedEvaluator(net, args.nn)
    eng = SelfPlayEngine(game, evaluator, num_simulations=args.sims, concurrent_games=args.games,
                         board_semantics=args.semantics, reference_quirks=args.quirks,
                         use_graph=not args.no_graph, seed=1000 + rank, device=dev,
                         first_game_index=rank, game_index_stride=world)
    stagger_start(eng, 4242 + rank)

    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")      # where collectives run

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.play_move()
    eng.collect()
    eng.ctx.reset_counters()
    barrier()
    t0 = time.perf_counter()
    positions = 0
    for _ in range(args.steps):
        positions += eng.play_move()
    barrier()
    dt = time.perf_counter() - t0
    counters = eng.ctx.status()
    ex = eng.collect()
    # the single exchange of the path: all-gather the examples produced in the timed region
    tg0 = time.perf_counter()
    ex_all = gather_examples(ex if cdev == dev else {k: v.to(cdev) for k, v in ex.items()})
    torch.cuda.synchronize()
    gather_s = time.perf_counter() - tg0
    tot = torch.tensor([float(positions), float(counters["evals"]), dt], dtype=torch.float64, device=cdev)
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx[2])
    positions_all, evals_all = float(tot[0]), float(tot[1])

    roof = cpub = None
    extra = {}
    if rank == 0:
        k_ms, n_launch, kc, nn_ms, eager_move_ms, tower_ms = roofline_pass(eng)
        nw = (args.rows * args.cols + 63) // 64
        bytes_per_launch, shape = algorithmic_bytes(kc, args.games, args.rows * args.cols, n_launch, nw)
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_k_mcts_pmc.json")
        if os.path.exists(pmc) and (args.games, args.rows, args.cols, args.sims) == (4096, 8, 8, 800):
            traffic = json.load(open(pmc))["per_launch_bytes"]["traffic_corrected"]
        roof_tree = {"bound": "hbm", "kernel": "k_mcts (fused expand+backup+select+rules+encode)", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": k_ms, "launches_timed": n_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                "pmc_profile": "profiles/ (rocprofv3 --pmc passes are collected offline; see DESIGN.md)", **shape}
        roof = roof_tree
        if tower_ms is not None:
            # dominant kernel of a step = the LDS-resident MFMA tower (csrc/yy_tower.hip): algorithmic FLOPs per
            # board = stem with K padded to 16 + 2 convs per block, each 2*9*128*128*64
            cells = args.rows * args.cols
            tower_flops = (2 * 9 * 16 * 128 * cells + 2 * args.blocks * (2 * 9 * 128 * 128 * cells)
                           + (2 * 128 * 64 * cells if getattr(eng.evaluator, "fused_heads", False) else 0)) * args.games
            ach = tower_flops / (tower_ms * 1e-3) / 1e12
            f32t = args.nn == "fp32t"
            if f32t:   # exact-f32 kernel: K of the stem padded to 8, no fused heads
                tower_flops = (2 * 9 * 8 * 128 * cells + 2 * args.blocks * (2 * 9 * 128 * 128 * cells)) * args.games
                ach = tower_flops / (tower_ms * 1e-3) / 1e12
            x3 = args.nn == "bf16x3"
            if x3:     # split-bf16 kernel: K of the stem padded to 16, no fused heads
                tower_flops = (2 * 9 * 16 * 128 * cells + 2 * args.blocks * (2 * 9 * 128 * 128 * cells)) * args.games
                ach = tower_flops / (tower_ms * 1e-3) / 1e12
            # split-bf16 issues three bf16 MFMAs per algorithmic multiply-add: its bound is a third of the bf16 peak
            peak = MFMA_F32_PEAK_TFLOPS if f32t else (MFMA_BF16_PEAK_TFLOPS / 3 if x3 else MFMA_BF16_PEAK_TFLOPS)
            name = "k_tower_f32 (stem + residual tower, exact f32 MFMA 32x32x2, activations LDS-resident)" if f32t else \
                "k_tower_x3 (stem + residual tower, split-bf16: 3 bf16 MFMAs per product term, f32-grade accuracy; peak = bf16 MFMA peak / 3)" if x3 else \
                {6: "k_towerq<6,8>", 8: "k_tower" if args.games > 512 else "k_towerq<8,%d>" % (1 if args.games <= 256 else 2), 12: "k_towerq<12,2>"}.get(args.rows, "k_tower") + \
                " (stem + residual tower + 1x1 head convs, bf16 MFMA, activations LDS-resident)"
            ttraffic = None      # HBM bytes per launch from the committed rocprofv3 --pmc passes (same G, same kernel)
            tpmc = os.path.join(ROOT, "profiles", "r01_k_tower_hbm_pmc.json")
            if os.path.exists(tpmc) and not f32t and getattr(eng.evaluator, "fused_heads", False) and \
                    (args.games, args.rows, args.cols, args.channels, args.blocks) == (4096, 8, 8, 128, 10):
                ttraffic = json.load(open(tpmc))["per_launch_bytes"]["traffic_corrected"]
            roof = {"bound": "mfma", "kernel": name,
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": ttraffic, "avg_launch_ms": tower_ms, "algorithmic_flops_per_launch": tower_flops}
            extra["roofline_tree_kernel"] = roof_tree
        A = args.rows * args.cols
        Cc = args.channels
        flops_leaf = (2 * 9 * 5 * Cc * A + 2 * args.blocks * (2 * 9 * Cc * Cc * A) + 2 * (2 * Cc * 32 * A)
                      + 2 * 32 * A * A + 2 * 32 * A * 256 + 512)
        extra = {**extra, "nn_forward_ms": nn_ms, "nn_tflops": flops_leaf * args.games / (nn_ms * 1e-3) / 1e12,
                 "tree_kernel_ms": k_ms, "tree_kernel_only_expansions_per_s": kc["evals"] / max(n_launch, 1) / (k_ms * 1e-3),
                 "eager_move_ms": eager_move_ms, "gather_s": gather_s, "examples_gathered": int(ex_all["states"].shape[0])}
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is reported at N=1 only
            cpub = cpu_baseline(args)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        sims_total = args.steps * args.sims * args.games * world
        line = {
            "metric": "self-play positions/sec (+ MCTS node-expansions/sec) at 800 sims, 8x8 board",
            "value": positions_all / dt, "unit": "positions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.nn, "data": "synthetic",
            "expansions_per_s": evals_all / dt, "simulations_per_s": sims_total / dt,
            "config": {"workload": f"{args.rows}x{args.cols} board, {args.sims} sims/move, {args.games} concurrent games "
                                   f"per GPU, frozen random-init {args.channels}x{args.blocks} net (seed 0), "
                                   f"{args.semantics} boards, reference_quirks={args.quirks}, staggered start plies",
                       "tree_arithmetic": "f32 PUCT + u64 bitboards", "nn_dtype": args.nn,
                       "parallelism": f"episode-sharded x{world}", "hipgraph": not args.no_graph},
            "roofline": roof, "cpu_baseline": cpub, **extra,
        }
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (the reference never travels to the GPU box);
the .npz files it writes are data: seeded inputs and the reference's outputs.

    cd /tmp/somewhere && python /root/repo/tests/golden/make_golden.py [--only g1,g2,...]

Fixture families:
  G1 rules_<R>x<C>.npz     getValidMoves(+1/-1), getGameEnded(+1/-1), count_pieces,
                           getNextState(random action) + placed bit
  G2 planes_<R>x<C>.npz    YinYangNeuralNetwork.board_to_input
  G3 search_<R>x<C>.npz    MCTS.search with the exact dyadic hash evaluator, both board
                           semantics, several simulation counts, optional root noise
     search_net_8x8.npz    MCTS.search with the real seeded 128x10 net, evaluator outputs recorded
  G4 episodes.npz          SelfPlayWorker.play_game transcripts (literal and copied adapter)
  G6 augment.npz           DataProcessor.augment_sample: the 8 symmetric (planes, policy) variants
  G8 symmetries.npz        YinYangGame.getSymmetries: the 8 (board, pi) forms in the reference's order
  G7 ucb.npz               Node.select_child on random child statistics (ties, unvisited children, f32 / python-float sums)

The reference imports create mcts.log / neural_network.log / training.log in the
CWD, so run from a scratch directory.  Nothing is written into the reference tree.
"""
import argparse
import logging
import multiprocessing as mp
import os
import sys

import numpy as np

REF = os.environ.get("YY_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    logging.disable(logging.CRITICAL)
    from src.yin_yang import YinYangGame  # noqa
    from src.yin_yang.yin_yang_logic import YinYangLogic  # noqa
    from src.yin_yang.ai.mcts import MCTS  # noqa
    return YinYangGame, YinYangLogic, MCTS


# --------------------------------------------------------------------------- hash evaluator
# Same integer hash as oracle/yy_oracle.c:yyo_hash_eval and tests/hash_eval.py.
def hash_eval_np(board_arr, pbits, vbits):
    b = np.asarray(board_arr, dtype=np.int8).reshape(-1)
    A = b.size
    h = 0x9E3779B9
    for i in range(A):
        code = int(b[i]) & 3
        h = (((h ^ code) * 16777619) + i) & 0xFFFFFFFF
    a = np.arange(A, dtype=np.uint64)
    x = (np.uint64(h) + a * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(12)
    pol = ((1 + (x & np.uint64((1 << pbits) - 1))).astype(np.float32)
           / np.float32(1 << (pbits + 6))).astype(np.float32)
    y = h ^ (h >> 16)
    y = (y * 0x045D9F3B) & 0xFFFFFFFF
    y ^= y >> 13
    half = 1 << (vbits - 1)
    val = np.float32(((y & ((1 << vbits) - 1)) - half) / half)
    return pol, val


class HashEvaluator:
    """predict(board) -> (np.float32[A], np.float32) exactly like the real net's dtypes."""

    def __init__(self, pbits=10, vbits=11, log=None):
        self.pbits, self.vbits, self.log = pbits, vbits, log

    def predict(self, board):
        arr = board.get_board()
        if self.log is not None:
            self.log.append(arr.copy())
        return hash_eval_np(arr, self.pbits, self.vbits)


class RecordingNet:
    def __init__(self, net):
        self.net, self.pol, self.val, self.boards = net, [], [], []

    def predict(self, board):
        p, v = self.net.predict(board)
        self.boards.append(board.get_board().copy())
        self.pol.append(p.copy())
        self.val.append(np.float32(v))
        return p, v


def make_copied_game(YinYangGame):
    """The 'copied' oracle: unmodified reference classes, driven through a game whose
    getNextState deep-copies first.  Harness code, not reference code."""
    import copy

    class CopiedGame(YinYangGame):
        def getNextState(self, board, player, action):
            return super().getNextState(copy.deepcopy(board), player, action)

    return CopiedGame


# --------------------------------------------------------------------------- board samplers
def random_play_positions(game, rng, n_keep):
    """Play one uniformly random legal game (passes allowed); return n_keep positions sampled
    uniformly over its plies (array copies), with the side to move."""
    b = game.getInitBoard()
    player = 1
    hist = [(b.get_board(), player)]
    passes = 0
    while True:
        v = game.getValidMoves(b, player)
        idx = np.where(v == 1)[0]
        if len(idx) == 0:
            passes += 1
            if passes >= 2:
                break
            player = -player
            continue
        passes = 0
        a = int(rng.choice(idx))
        b, player = game.getNextState(b, player, a)
        hist.append((b.get_board(), player))
    sel = rng.choice(len(hist), size=min(n_keep, len(hist)), replace=False)
    return [hist[i] for i in sel]


def adversarial_board(R, C, rng, kind):
    A = R * C
    if kind == 0:      # iid fill with random density
        pe = rng.uniform(0.05, 0.95)
        pb = rng.uniform(0.2, 0.8)
        u = rng.random(A)
        c = rng.random(A)
        arr = np.where(u < pe, 0, np.where(c < pb, 1, -1))
    elif kind == 1:    # full board
        arr = np.where(rng.random(A) < rng.uniform(0.3, 0.7), 1, -1)
    elif kind == 2:    # exactly one empty cell
        arr = np.where(rng.random(A) < rng.uniform(0.3, 0.7), 1, -1)
        arr[rng.integers(A)] = 0
    elif kind == 3:    # one colour only, sparse
        arr = np.where(rng.random(A) < rng.uniform(0.05, 0.5), rng.choice([1, -1]), 0)
    elif kind == 4:    # blobs grown from seeds: few components per colour
        arr = np.zeros(A, dtype=np.int64)
        grid = arr.reshape(R, C)
        for col in (1, -1):
            for _ in range(rng.integers(1, 4)):
                x, y = rng.integers(R), rng.integers(C)
                for _ in range(rng.integers(1, max(2, A // 4))):
                    if grid[x, y] == 0:
                        grid[x, y] = col
                    d = rng.integers(4)
                    x = min(R - 1, max(0, x + (0, 1, 0, -1)[d]))
                    y = min(C - 1, max(0, y + (1, 0, -1, 0)[d]))
    else:              # stripes / checker-ish patterns with holes
        ii, jj = np.indices((R, C))
        m = rng.integers(0, 4)
        base = [(ii + jj) % 2, ii % 2, jj % 2, (ii // 2 + jj // 2) % 2][m]
        arr = np.where(base == 1, 1, -1).reshape(-1)
        holes = rng.random(A) < rng.uniform(0.05, 0.6)
        arr = np.where(holes, 0, arr)
    return np.asarray(arr, dtype=np.int8).reshape(R, C)


# --------------------------------------------------------------------------- G1 / G2
def _g1_chunk(args):
    R, C, seed, n_play, n_adv = args
    YinYangGame, YinYangLogic, _ = _import_ref()
    game = YinYangGame(R, C)
    rng = np.random.default_rng(seed)
    boards, players = [], []
    while len(boards) < n_play:
        for arr, pl in random_play_positions(game, rng, 8):
            if len(boards) < n_play:
                boards.append(arr)
                players.append(pl)
    for i in range(n_adv):
        boards.append(adversarial_board(R, C, rng, i % 6))
        players.append(int(rng.choice([1, -1])))
    A = R * C
    n = len(boards)
    out = dict(
        boards=np.stack(boards).astype(np.int8),
        players=np.asarray(players, dtype=np.int8),
        mask_p1=np.zeros((n, A), np.uint8), mask_m1=np.zeros((n, A), np.uint8),
        ended_p1=np.zeros(n, np.float64), ended_m1=np.zeros(n, np.float64),
        counts=np.zeros((n, 2), np.int32),
        step_action=np.zeros(n, np.int32), step_placed=np.zeros(n, np.uint8),
        step_board=np.zeros((n, R, C), np.int8), step_player=np.zeros(n, np.int8),
    )
    for i in range(n):
        lb = YinYangLogic(R, C)
        lb.board = out["boards"][i].copy()
        out["mask_p1"][i] = game.getValidMoves(lb, 1).astype(np.uint8)
        out["mask_m1"][i] = game.getValidMoves(lb, -1).astype(np.uint8)
        out["ended_p1"][i] = game.getGameEnded(lb, 1)
        out["ended_m1"][i] = game.getGameEnded(lb, -1)
        out["counts"][i] = lb.count_pieces()
        assert np.array_equal(lb.board, out["boards"][i])
        # step: half the time a legal action (if any), else a uniformly random one
        pl = int(out["players"][i])
        legal = np.where((out["mask_p1"][i] if pl == 1 else out["mask_m1"][i]) == 1)[0]
        if len(legal) and rng.random() < 0.5:
            a = int(rng.choice(legal))
        else:
            a = int(rng.integers(A))
        before = lb.board.copy()
        nb, npl = game.getNextState(lb, pl, a)
        assert nb is lb
        out["step_action"][i] = a
        out["step_board"][i] = nb.board
        out["step_player"][i] = npl
        out["step_placed"][i] = int(not np.array_equal(before, nb.board))
    return out


def gen_g1(sizes, n_total, pool):
    for (R, C) in sizes:
        chunks = 16
        per = n_total // chunks
        jobs = [(R, C, 1000 * R + 10 * C + k, per // 2, per - per // 2) for k in range(chunks)]
        parts = pool.map(_g1_chunk, jobs)
        out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        path = os.path.join(OUT, f"rules_{R}x{C}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, out["boards"].shape, "legal-move density",
              out["mask_p1"].mean(), "ended frac", (out["ended_p1"] != 0).mean(), flush=True)


def gen_g2(sizes, n):
    YinYangGame, YinYangLogic, _ = _import_ref()
    from src.yin_yang.ai.neural_network import YinYangNeuralNetwork
    for (R, C) in sizes:
        game = YinYangGame(R, C)
        net = YinYangNeuralNetwork(game, num_channels=8, num_res_blocks=1)
        rng = np.random.default_rng(77 + R * 31 + C)
        boards = []
        while len(boards) < n // 2:
            boards += [a for a, _ in random_play_positions(game, rng, 8)]
        boards = boards[: n // 2] + [adversarial_board(R, C, rng, i % 6) for i in range(n - n // 2)]
        planes = np.zeros((n, 5, R, C), np.float32)
        for i, arr in enumerate(boards):
            lb = YinYangLogic(R, C)
            lb.board = arr.copy()
            planes[i] = net.board_to_input(lb).numpy()
        path = os.path.join(OUT, f"planes_{R}x{C}.npz")
        np.savez_compressed(path, boards=np.stack(boards).astype(np.int8), planes=planes)
        print("wrote", path, flush=True)


# --------------------------------------------------------------------------- G3
def _search_case(args):
    (R, C, seed, sims, copied, pbits, vbits, noise, root_player, start_kind, keep_leaves) = args
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    Game = make_copied_game(YinYangGame) if copied else YinYangGame
    game = Game(R, C)
    rng = np.random.default_rng(seed)
    # root position: empty, or a random-legal-play position
    if start_kind == 0:
        arr = np.zeros((R, C), np.int8)
    else:
        plain = YinYangGame(R, C)
        arr, _pl = random_play_positions(plain, rng, 1)[0]
    lb = YinYangLogic(R, C)
    lb.board = arr.copy()
    log = []
    ev = HashEvaluator(pbits, vbits, log)
    mcts = MCTS(game, ev, num_simulations=sims, cpuct=1.0, verbose=0)
    A = R * C
    noise_vec = np.zeros(A, np.float64)
    if noise:
        # reproduce the draw the reference will make (same global stream, same call)
        np.random.seed(seed)
        valid = YinYangGame(R, C).getValidMoves(lb, root_player)
        idx = np.where(valid == 1)[0]
        if len(idx) > 0:
            draw = np.random.dirichlet([0.3] * len(idx))
            noise_vec[idx] = draw
        np.random.seed(seed)
    pi, root = mcts.search(lb, root_player, add_exploration_noise=bool(noise))
    counts = np.zeros(A, np.int32)
    cw = np.zeros(A, np.float64)
    cp = np.zeros(A, np.float32)
    for a, ch in root.children.items():
        counts[a] = ch.visits
        cw[a] = float(ch.value_sum)
        cp[a] = np.float32(ch.prior)
    leaves = np.stack(log[1:]) if len(log) > 1 else np.zeros((0, R, C), np.int8)
    return dict(
        root_board=arr.astype(np.int8), root_player=np.int8(root_player), sims=np.int32(sims),
        copied=np.uint8(copied), pbits=np.int32(pbits), vbits=np.int32(vbits),
        noise=noise_vec, has_noise=np.uint8(noise), pi=pi.astype(np.float64), counts=counts,
        child_w=cw, child_p=cp, root_visits=np.int32(root.visits),
        root_w=np.float64(float(root.value_sum)), n_evals=np.int32(len(log) - 1),
        final_board=lb.board.astype(np.int8),
        leaves=leaves.astype(np.int8) if keep_leaves else np.zeros((0, R, C), np.int8),
        n_leaves=np.int32(leaves.shape[0] if keep_leaves else 0),
    )


def gen_g3(sizes, pool):
    for (R, C) in sizes:
        jobs = []
        big = (R * C >= 144)
        sims_list = [25, 200, 800] if not big else [25, 200, 1600]
        seed = 5000 + 100 * R + C
        for copied in (0, 1):
            for sims in sims_list:
                n_roots = {25: 16, 200: 8, 800: 4, 1600: 2}[sims]
                if R * C <= 16:
                    n_roots *= 2
                for k in range(n_roots):
                    seed += 1
                    pbits, vbits = ((10, 11), (2, 2), (6, 4))[k % 3]   # fine / tie-heavy / medium
                    noise = 1 if (k % 4 == 1) else 0
                    root_player = -1 if (k % 5 == 2) else 1
                    start_kind = 0 if (k % 2 == 0) else 1
                    keep = 1 if sims <= 200 else 0
                    jobs.append((R, C, seed, sims, copied, pbits, vbits, noise, root_player,
                                 start_kind, keep))
        res = pool.map(_search_case, jobs, chunksize=1)
        A = R * C
        n = len(res)
        maxleaf = max(int(r["n_leaves"]) for r in res)
        out = {}
        for k in res[0]:
            if k == "leaves":
                continue
            out[k] = np.stack([np.asarray(r[k]) for r in res])
        leaves = np.zeros((n, maxleaf, R, C), np.int8)
        for i, r in enumerate(res):
            leaves[i, : r["leaves"].shape[0]] = r["leaves"]
        out["leaves"] = leaves
        path = os.path.join(OUT, f"search_{R}x{C}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, n, "cases; evals", out["n_evals"].sum(), flush=True)


def _search_net_case(args):
    R, C, seed, sims, copied, noise = args
    import torch
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    from src.yin_yang.ai.neural_network import YinYangNeuralNetwork
    torch.manual_seed(0)
    torch.set_num_threads(1)
    plain = YinYangGame(R, C)
    net = YinYangNeuralNetwork(plain)          # default 128 x 10
    Game = make_copied_game(YinYangGame) if copied else YinYangGame
    game = Game(R, C)
    rng = np.random.default_rng(seed)
    arr = np.zeros((R, C), np.int8) if seed % 2 == 0 else random_play_positions(plain, rng, 1)[0][0]
    lb = YinYangLogic(R, C)
    lb.board = arr.copy()
    rec = RecordingNet(net)
    mcts = MCTS(game, rec, num_simulations=sims, cpuct=1.0, verbose=0)
    A = R * C
    noise_vec = np.zeros(A, np.float64)
    if noise:
        np.random.seed(seed)
        idx = np.where(plain.getValidMoves(lb, 1) == 1)[0]
        if len(idx):
            noise_vec[idx] = np.random.dirichlet([0.3] * len(idx))
        np.random.seed(seed)
    pi, root = mcts.search(lb, 1, add_exploration_noise=bool(noise))
    counts = np.zeros(A, np.int32)
    cw = np.zeros(A, np.float64)
    cp = np.zeros(A, np.float32)
    for a, ch in root.children.items():
        counts[a], cw[a], cp[a] = ch.visits, float(ch.value_sum), np.float32(ch.prior)
    return dict(root_board=arr.astype(np.int8), sims=np.int32(sims), copied=np.uint8(copied),
                noise=noise_vec, has_noise=np.uint8(noise), pi=pi, counts=counts, child_w=cw,
                child_p=cp, root_w=np.float64(float(root.value_sum)),
                rec_policy=np.stack(rec.pol).astype(np.float32),
                rec_value=np.asarray(rec.val, np.float32),
                rec_boards=np.stack(rec.boards).astype(np.int8),
                final_board=lb.board.astype(np.int8))


def gen_g3_net(pool):
    R = C = 8
    sims = 160
    jobs = [(R, C, 900 + k, sims, k % 2, 1 if k >= 2 else 0) for k in range(4)]
    res = pool.map(_search_net_case, jobs, chunksize=1)
    n = max(r["rec_policy"].shape[0] for r in res)
    out = {}
    for k in res[0]:
        if k.startswith("rec_"):
            shp = (len(res), n) + res[0][k].shape[1:]
            arr = np.zeros(shp, res[0][k].dtype)
            for i, r in enumerate(res):
                arr[i, : r[k].shape[0]] = r[k]
            out[k] = arr
        else:
            out[k] = np.stack([np.asarray(r[k]) for r in res])
    out["n_rec"] = np.asarray([r["rec_policy"].shape[0] for r in res], np.int32)
    path = os.path.join(OUT, "search_net_8x8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, flush=True)


def _search_net800_case(args):
    """One 800-simulation search with the real seeded 128x10 net (CPU fp32, as the reference runs it): only the RESULTS are
    kept (pi, counts, child statistics, root value sum, the root evaluation) -- the live GPU evaluators are compared with these."""
    R, C, seed, sims, copied, noise, root_player = args
    r = _search_net_case((R, C, seed, sims, copied, noise)) if root_player == 1 else None
    assert r is not None
    out = {k: v for k, v in r.items() if not k.startswith("rec_")}
    out["root_policy"] = r["rec_policy"][0]
    out["root_value"] = r["rec_value"][0]
    out["n_evals"] = np.int32(r["rec_policy"].shape[0] - 1)
    out["seed"] = np.int32(seed)
    return out


def gen_g3_net800(pool):
    """search_net800_8x8.npz: 64 roots (empty board and random legal-play positions) x 800 simulations, both board semantics,
    with and without root noise (real-net part; ai/mcts.py:275-343 with ai/neural_network.py:125-154)."""
    R = C = 8
    jobs = [(R, C, 2000 + k, 800, k % 2, (k // 2) % 2, 1) for k in range(64)]
    res = pool.map(_search_net800_case, jobs, chunksize=1)
    out = {k: np.stack([np.asarray(r[k]) for r in res]) for k in res[0]}
    path = os.path.join(OUT, "search_net800_8x8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(res), "cases", flush=True)


# --------------------------------------------------------------------------- G4
def _episode_case(args):
    R, C, seed, sims, copied, pbits, vbits = args
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    from src.yin_yang.ai.self_play import SelfPlayWorker
    Game = make_copied_game(YinYangGame) if copied else YinYangGame
    game = Game(R, C)
    worker = SelfPlayWorker(game, "/nonexistent/model.pth.tar", num_simulations=sims)
    worker.mcts.neural_net = HashEvaluator(pbits, vbits)
    A = R * C
    trace = dict(boards=[], pis=[], actions=[], players=[])
    state = {"in_search": False}
    orig_search = worker.mcts.search
    orig_next = game.getNextState

    def search_wrapper(board, player, add_exploration_noise=False):
        trace["boards"].append(board.get_board().copy())      # board handed to search
        state["in_search"] = True
        try:
            pi, root = orig_search(board, player, add_exploration_noise)
        finally:
            state["in_search"] = False
        trace["pis"].append(pi.copy())
        return pi, root

    def traced_next(board, player, action):
        if not state["in_search"]:                            # play_game's own call (self_play.py:163)
            trace["actions"].append(int(action))
            trace["players"].append(int(player))
        return orig_next(board, player, action)

    worker.mcts.search = search_wrapper
    game.getNextState = traced_next
    np.random.seed(seed)
    examples = worker.play_game()
    n = len(examples)
    T = 4 * A + 8
    assert n <= T and n == len(trace["pis"]) == len(trace["actions"])
    out = dict(
        seed=np.int32(seed), sims=np.int32(sims), copied=np.uint8(copied), pbits=np.int32(pbits),
        vbits=np.int32(vbits), n=np.int32(n),
        search_boards=np.zeros((T, R, C), np.int8), pis=np.zeros((T, A), np.float64),
        actions=np.full(T, -1, np.int32), players=np.zeros(T, np.int8),
        z=np.zeros(T, np.float64), example_boards=np.zeros((T, R, C), np.int8),
    )
    for i in range(n):
        out["search_boards"][i] = trace["boards"][i]
        out["pis"][i] = trace["pis"][i]
        out["actions"][i] = trace["actions"][i]
        out["players"][i] = trace["players"][i]
        out["z"][i] = examples[i][2]
        out["example_boards"][i] = examples[i][0].get_board()
        assert np.array_equal(examples[i][1], trace["pis"][i])
    return out


def gen_g4(pool):
    jobs = []
    seed = 300
    for (R, C, sims) in ((6, 6, 25), (8, 8, 40), (4, 4, 30)):
        for copied in (0, 1):
            for k in range(6 if copied else 10):
                seed += 1
                pb, vb = ((10, 11), (2, 2))[k % 2]
                jobs.append((R, C, seed, sims, copied, pb, vb))
    res = pool.map(_episode_case, jobs, chunksize=1)
    # group by board size (arrays have size-dependent shapes)
    for (R, C) in ((6, 6), (8, 8), (4, 4)):
        sel = [r for r, j in zip(res, jobs) if (j[0], j[1]) == (R, C)]
        out = {k: np.stack([np.asarray(r[k]) for r in sel]) for k in sel[0]}
        path = os.path.join(OUT, f"episodes_{R}x{C}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, "games", len(sel), "examples per game", out["n"].tolist(), flush=True)


# --------------------------------------------------------------------------- G6 augmentation
def gen_g6():
    """DataProcessor.augment_sample (data_utils.py:39-134): the 8 (planes, policy) variants, in order."""
    import torch
    YinYangGame, YinYangLogic, _ = _import_ref()
    from src.yin_yang.ai.neural_network import YinYangNeuralNetwork
    from src.yin_yang.ai.data_utils import DataProcessor
    out = {}
    for (R, C) in ((6, 6), (8, 8)):
        game = YinYangGame(R, C)
        net = YinYangNeuralNetwork(game, num_channels=8, num_res_blocks=1)
        proc = DataProcessor(game)
        rng = np.random.default_rng(600 + R)
        boards, pis, aug_planes, aug_pis = [], [], [], []
        for _ in range(8):
            arr, _pl = random_play_positions(game, rng, 1)[0]
            lb = YinYangLogic(R, C)
            lb.board = arr.copy()
            pi = rng.random(R * C) * (rng.random(R * C) < 0.4)
            pi = pi / max(pi.sum(), 1e-9)
            planes = net.board_to_input(lb)
            aug = proc.augment_sample(planes, torch.FloatTensor(pi))
            assert len(aug) == 8
            boards.append(arr)
            pis.append(pi.astype(np.float32))
            aug_planes.append(np.stack([a[0].numpy() for a in aug]))
            aug_pis.append(np.stack([np.asarray(a[1], dtype=np.float32) for a in aug]))
        out[f"boards_{R}"] = np.stack(boards).astype(np.int8)
        out[f"pi_{R}"] = np.stack(pis)
        out[f"aug_planes_{R}"] = np.stack(aug_planes).astype(np.float32)
        out[f"aug_pi_{R}"] = np.stack(aug_pis).astype(np.float32)
    path = os.path.join(OUT, "augment.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, flush=True)


def gen_sym():
    """G8 symmetries.npz: YinYangGame.getSymmetries (yin_yang_game.py:127-166) on random square boards: the 8 (board, pi) forms
    in the reference's order."""
    YinYangGame, YinYangLogic, _ = _import_ref()
    rng = np.random.default_rng(88)
    out = {}
    for (R, C) in ((4, 4), (6, 6), (8, 8)):
        game = YinYangGame(R, C)
        n = 6
        boards = rng.integers(-1, 2, size=(n, R, C)).astype(np.int8)
        pis = rng.dirichlet(np.ones(R * C), size=n)
        sb = np.zeros((n, 8, R, C), np.int8)
        sp = np.zeros((n, 8, R * C), np.float64)
        for i in range(n):
            lb = YinYangLogic(R, C)
            lb.board = boards[i].copy()
            syms = game.getSymmetries(lb, pis[i])
            assert len(syms) == 8
            for k, (b, p) in enumerate(syms):
                sb[i, k], sp[i, k] = b.get_board(), p
        out[f"boards_{R}"], out[f"pis_{R}"], out[f"sym_boards_{R}"], out[f"sym_pis_{R}"] = boards, pis, sb, sp
    path = os.path.join(OUT, "symmetries.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, flush=True)


def gen_ucb():
    """G7 ucb.npz: Node.select_child (mcts.py:97-145) on random child statistics, with ties, unvisited children, python-float
    and float32 value sums."""
    _import_ref()
    from src.yin_yang.ai.mcts import Node
    rng = np.random.default_rng(77)
    n, K = 4000, 12
    out = dict(k=np.zeros(n, np.int32), visits=np.zeros((n, K), np.int32), wsum=np.zeros((n, K), np.float32),
               w_is_f32=np.zeros((n, K), np.uint8), prior=np.zeros((n, K), np.float32), cpuct=np.zeros(n, np.float64),
               chosen=np.zeros(n, np.int32))
    for i in range(n):
        k = int(rng.integers(1, K + 1))
        node = Node(None)
        cp = float(rng.choice([1.0, 0.5, 2.0, 1.25]))
        coarse = rng.random() < 0.5                       # coarse grids make exact ties likely
        for a in range(k):
            ch = Node(None, parent=node, action=a, prior=np.float32(rng.integers(1, 9) / 16.0 if coarse else rng.random()))
            ch.visits = int(rng.integers(0, 4 if coarse else 40))
            if ch.visits:
                w = rng.integers(-ch.visits, ch.visits + 1) / (2.0 if coarse else 1.0) * (1.0 if coarse else rng.random())
                if rng.random() < 0.8:
                    ch.value_sum = np.float32(w)
                    out["w_is_f32"][i, a] = 1
                else:
                    ch.value_sum = float(np.float32(w))   # still a python float: only terminal values were added
            node.children[a] = ch
            out["visits"][i, a], out["wsum"][i, a], out["prior"][i, a] = ch.visits, ch.value_sum, ch.prior
        a_sel, _ = node.select_child(cp)
        out["k"][i], out["cpuct"][i], out["chosen"][i] = k, cp, a_sel
    path = os.path.join(OUT, "ucb.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g1,g2,g3,g3net,g4,g6")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--n-rules", type=int, default=10240)
    args = ap.parse_args()
    only = set(args.only.split(","))
    assert not os.path.abspath(os.getcwd()).startswith(os.path.abspath(REF)), "run from a scratch dir"
    pool = mp.Pool(args.procs)
    if "g1" in only:
        gen_g1([(6, 6), (8, 8), (12, 12)], args.n_rules, pool)
        gen_g1([(3, 3), (4, 4), (5, 7), (9, 4)], 1024, pool)
    if "g2" in only:
        gen_g2([(6, 6), (8, 8), (12, 12), (3, 3), (5, 7)], 1000)
    if "g3" in only:
        gen_g3([(3, 3), (4, 4), (6, 6), (8, 8), (12, 12), (5, 7)], pool)
    if "g3net" in only:
        gen_g3_net(pool)
    if "g3net800" in only:
        gen_g3_net800(pool)
    if "g4" in only:
        gen_g4(pool)
    if "g6" in only:
        gen_g6()
    if "ucb" in only:
        gen_ucb()
    if "sym" in only:
        gen_sym()
    if "edge" in only:      # degenerate and maximum geometries (1 x N, N x 1, 2 x 2, the 192-cell / 16-wide limits)
        gen_g1([(1, 1), (1, 6), (7, 1), (2, 2), (2, 9), (13, 14), (16, 12), (12, 16)], 256, pool)
        gen_g2([(1, 6), (7, 1), (16, 12)], 64)
        gen_g3([(1, 6), (2, 2), (7, 1), (16, 12), (12, 16)], pool)
    pool.close()


if __name__ == "__main__":
    main()

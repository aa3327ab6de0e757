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
    R, C, seed, sims, copied, noise = args[:6]
    net_shape = args[6] if len(args) > 6 else None          # (num_channels, num_res_blocks); None = the default 128 x 10
    import torch
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    from src.yin_yang.ai.neural_network import YinYangNeuralNetwork
    torch.manual_seed(0)
    torch.set_num_threads(1)
    plain = YinYangGame(R, C)
    net = YinYangNeuralNetwork(plain) if net_shape is None else YinYangNeuralNetwork(plain, *net_shape)   # default 128 x 10
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
    R, C, seed, sims, copied, noise, root_player = args[:7]
    assert root_player == 1
    r = _search_net_case((R, C, seed, sims, copied, noise) + tuple(args[7:]))
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


# (file tag, R, C, sims, roots, first seed, net shape): BASELINE configs 4 and 1 with the default 128 x 10 net, and shapes the
# generalised split-f16 tower must cover (non-square boards, other widths / depths: train_alphazero.py:35-36, neural_network.py:39)
NET_SEARCH_SETS = {
    "net1600_12x12": (12, 12, 1600, 24, 3000, None),
    "net800_6x6": (6, 6, 800, 32, 4000, None),
    "net25_6x6": (6, 6, 25, 32, 4100, None),
    "net400_10x10_c64b4": (10, 10, 400, 16, 5000, (64, 4)),
    "net400_5x7_c96b3": (5, 7, 400, 16, 5100, (96, 3)),
    "net200_9x12_c32b2": (9, 12, 200, 16, 5200, (32, 2)),
}


def gen_g3_net_sets(pool, names):
    """search_<tag>.npz: the reference's MCTS.search with its own CPU float32 network (seeded random init) at the other BASELINE
    sizes -- roots = the empty board (even seeds) and random legal-play positions at random plies (odd seeds), both board
    semantics, with and without root noise; results only (ai/mcts.py:275-343 with ai/neural_network.py:125-154)."""
    for name in names:
        R, C, sims, n, seed0, shape = NET_SEARCH_SETS[name]
        jobs = [(R, C, seed0 + k, sims, k % 2, (k // 2) % 2, 1) + ((shape,) if shape else ()) for k in range(n)]
        res = pool.map(_search_net800_case, jobs, chunksize=1)
        out = {k: np.stack([np.asarray(r[k]) for r in res]) for k in res[0]}
        out["net_shape"] = np.asarray(shape if shape else (128, 10), np.int32)
        path = os.path.join(OUT, "search_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote", path, len(res), "cases", flush=True)


# --------------------------------------------------------------------------- row/column rule (browser JS only)
def rowcol_boards(R, C, rng, n):
    """Inputs for the row/column rule: random legal play of the PYTHON reference (which has no such rule: its positions
    contain full single-coloured lines the JS must react to), iid fills, and boards built to stress the rule: lines with
    exactly one empty cell whose other cells are one colour / mixed, full single-coloured lines already on the board."""
    YinYangGame, _, _ = _import_ref()
    game = YinYangGame(R, C)
    boards = []
    while len(boards) < n // 3:
        boards += [a for a, _ in random_play_positions(game, rng, 6)]
    boards = boards[: n // 3]
    while len(boards) < n:
        kind = len(boards) % 4
        if kind == 0:
            arr = adversarial_board(R, C, rng, int(rng.integers(6)))
        else:
            arr = np.where(rng.random((R, C)) < rng.uniform(0.2, 0.9), rng.choice([1, -1], size=(R, C)), 0).astype(np.int8)
            for _ in range(int(rng.integers(1, 3))):
                colour = int(rng.choice([1, -1]))
                line = colour * np.ones(C if kind != 2 else R, np.int8)
                if kind == 3:                                  # mixed line: one stone of the other colour
                    line[rng.integers(len(line))] = -colour
                if rng.random() < 0.8:                         # exactly one empty cell
                    line[rng.integers(len(line))] = 0
                if kind != 2:
                    arr[rng.integers(R), :] = line
                else:
                    arr[:, rng.integers(C)] = line
        boards.append(np.asarray(arr, np.int8))
    return np.stack(boards)


def gen_rowcol():
    """rowcol_<R>x<C>.npz: legal-move masks of the reference's JavaScript game (yin_yang_game.js:187-232, 338-384) for both
    colours -- the pin of the YY_FLAG_ROWCOL rule path (oracle and HIP)."""
    import json
    import subprocess
    js = os.path.join(REF, "src", "gui", "static", "js", "yin_yang_game.js")
    sizes = [(6, 6), (8, 8), (12, 12), (4, 4), (3, 3), (5, 7), (2, 9), (1, 6), (7, 1), (16, 12)]
    sets, inputs = [], []
    for (R, C) in sizes:
        rng = np.random.default_rng(4000 + 100 * R + C)
        b = rowcol_boards(R, C, rng, 2048 if R * C >= 36 else 512)
        sets.append(b)
        inputs.append(dict(rows=R, cols=C, boards=b.reshape(len(b), -1).tolist()))
    res = subprocess.run(["node", os.path.join(OUT, "rowcol_from_js.js"), js], input=json.dumps(inputs).encode(),
                         stdout=subprocess.PIPE, check=True)
    for b, r in zip(sets, json.loads(res.stdout)):
        R, C = r["rows"], r["cols"]
        path = os.path.join(OUT, f"rowcol_{R}x{C}.npz")
        p1, m1 = np.asarray(r["p1"], np.uint8), np.asarray(r["m1"], np.uint8)
        np.savez_compressed(path, boards=b, mask_p1=p1, mask_m1=m1)
        print("wrote", path, b.shape, "legal density", p1.mean(), m1.mean(), flush=True)


# --------------------------------------------------------------------------- arena / select_action
class _HashNetFromPath:
    """Stands in for YinYangNeuralNetwork inside the reference's AlphaZero.evaluate: `load_model(path)` picks the hash
    evaluator from the file NAME (current -> fine, best -> medium), so the reference's own match loop runs unmodified."""

    def __init__(self, game):
        self.pbits, self.vbits = 10, 11

    def load_model(self, path):
        self.pbits, self.vbits = (10, 11) if "current" in os.path.basename(path) else (6, 4)

    def predict(self, board):
        return hash_eval_np(board.get_board(), self.pbits, self.vbits)


def _arena_case(args):
    R, C, sims, copied, num_games = args
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    import src.yin_yang.ai.alphazero as az
    Game = make_copied_game(YinYangGame) if copied else YinYangGame
    game = Game(R, C)
    trace = dict(actions=[[]], players=[[]], results=[])
    orig_next, orig_ended = game.getNextState, game.getGameEnded
    state = {"depth": 0}
    orig_search = MCTS.search

    def traced_search(self, board, player, add_exploration_noise=False):
        state["depth"] += 1
        try:
            return orig_search(self, board, player, add_exploration_noise)
        finally:
            state["depth"] -= 1

    def traced_next(board, player, action):
        if state["depth"] == 0:                                # the match loop's own call (alphazero.py:198)
            trace["actions"][-1].append(int(action))
            trace["players"][-1].append(int(player))
        return orig_next(board, player, action)

    def traced_ended(board, player):
        r = orig_ended(board, player)
        if state["depth"] == 0 and r != 0:                     # alphazero.py:201-202: the game is over
            trace["results"].append(float(r))
            trace["actions"].append([])
            trace["players"].append([])
        return r

    game.getNextState, game.getGameEnded = traced_next, traced_ended
    MCTS.search = traced_search
    az.YinYangNeuralNetwork = _HashNetFromPath
    try:
        a = az.AlphaZero.__new__(az.AlphaZero)
        a.game, a.num_simulations, a.mcts_threads = game, sims, 1
        ratio = a.evaluate("current_model.pth.tar", "best_model.pth.tar", num_games=num_games)
    finally:
        MCTS.search = orig_search
    T = max(len(x) for x in trace["actions"]) + 1
    acts = np.full((num_games, T), -1, np.int32)
    pls = np.zeros((num_games, T), np.int8)
    for i in range(num_games):
        n = len(trace["actions"][i])
        acts[i, :n], pls[i, :n] = trace["actions"][i], trace["players"][i]
    res = np.asarray(trace["results"], np.float64)
    assert len(res) == num_games
    # the reference's own accounting (alphazero.py:204-216)
    cur = best = draws = 0
    for i, r in enumerate(res):
        first_is_current = (i % 2 == 0)
        if r == 1:
            cur, best = (cur + 1, best) if first_is_current else (cur, best + 1)
        elif r == -1:
            cur, best = (cur, best + 1) if first_is_current else (cur + 1, best)
        else:
            draws += 1
    assert abs(ratio - cur / num_games) < 1e-12
    return dict(sims=np.int32(sims), copied=np.uint8(copied), actions=acts, players=pls, results=res,
                n_moves=np.asarray([len(x) for x in trace["actions"][:num_games]], np.int32),
                current_wins=np.int32(cur), best_wins=np.int32(best), draws=np.int32(draws), win_ratio=np.float64(ratio))


def gen_arena(pool):
    """arena_<R>x<C>.npz: the reference's AlphaZero.evaluate (alphazero.py:136-226) run unmodified with two hash evaluators
    ("current" fine, "best" medium): every move of every game, the game results as the loop saw them, and its win / loss / draw
    accounting -- literal (aliased boards) and with the copied-board game adapter."""
    for (R, C, sims, n) in ((4, 4, 30, 8), (6, 6, 40, 8), (8, 8, 50, 6)):
        res = pool.map(_arena_case, [(R, C, sims, copied, n) for copied in (0, 1)], chunksize=1)
        out = {}
        for tag, r in zip(("literal", "copied"), res):
            for k, v in r.items():
                out[f"{tag}_{k}"] = v
        path = os.path.join(OUT, f"arena_{R}x{C}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, {t: (int(out[t + "_current_wins"]), int(out[t + "_best_wins"]), int(out[t + "_draws"]),
                                  out[t + "_n_moves"].tolist()) for t in ("literal", "copied")}, flush=True)


def gen_select_action():
    """select_action.npz: MCTS.select_action (mcts.py:427-479) -- temperature 0 / 1 / 0.5, with and without valid_moves
    (incl. a mask that removes all visited moves and a wrong-length mask), seeded np.random: the search's pi (hash evaluator)
    and the action returned."""
    YinYangGame, YinYangLogic, MCTS = _import_ref()
    rng = np.random.default_rng(91)
    rows = []
    for (R, C, sims) in ((3, 3, 20), (6, 6, 40), (8, 8, 60)):
        plain = YinYangGame(R, C)
        Game = make_copied_game(YinYangGame)
        game = Game(R, C)
        A = R * C
        for k in range(24):
            arr = np.zeros((R, C), np.int8) if k % 3 == 0 else random_play_positions(plain, rng, 1)[0][0]
            player = -1 if k % 4 == 3 else 1
            temp = (0, 1.0, 0.5, None)[k % 4]
            kind = k % 6                      # 0,1: no mask; 2,3: the legal mask; 4: mask removing the searched moves; 5: wrong length
            lb = YinYangLogic(R, C)
            lb.board = arr.copy()
            valid = plain.getValidMoves(lb, player)
            mcts = MCTS(game, HashEvaluator(6, 4), num_simulations=sims, cpuct=1.0, verbose=0)
            seed = 700 + k
            np.random.seed(seed)
            pi, _ = mcts.search(lb, player)
            if kind in (0, 1):
                vm = None
            elif kind in (2, 3):
                vm = valid.copy()
            elif kind == 4:
                vm = ((pi == 0) & (np.arange(A) % 2 == 0)).astype(np.float64)
            else:
                vm = np.ones(A + 3)
            np.random.seed(seed)
            lb2 = YinYangLogic(R, C)
            lb2.board = arr.copy()
            if temp is not None and temp != 0 and vm is None and pi.sum() > 0 and kind in (0, 1) and valid.sum() == 0:
                continue
            action = mcts.select_action(lb2, player, temperature=temp, valid_moves=vm)
            rows.append(dict(R=R, C=C, sims=sims, board=arr, player=player, temp=-1.0 if temp is None else float(temp),
                             has_mask=vm is not None, mask=np.zeros(A + 3) if vm is None else np.pad(vm, (0, A + 3 - len(vm))),
                             mask_len=0 if vm is None else len(vm), seed=seed, pi=pi, action=int(action)))
    out = {}
    for (R, C) in ((3, 3), (6, 6), (8, 8)):
        sel = [r for r in rows if (r["R"], r["C"]) == (R, C)]
        tag = f"{R}x{C}"
        out[f"board_{tag}"] = np.stack([r["board"] for r in sel]).astype(np.int8)
        out[f"pi_{tag}"] = np.stack([r["pi"] for r in sel])
        out[f"mask_{tag}"] = np.stack([r["mask"] for r in sel])
        for k, dt in (("player", np.int8), ("temp", np.float64), ("has_mask", np.uint8), ("mask_len", np.int32),
                      ("seed", np.int32), ("action", np.int32), ("sims", np.int32)):
            out[f"{k}_{tag}"] = np.asarray([r[k] for r in sel], dt)
    path = os.path.join(OUT, "select_action.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("action")}, flush=True)


# --------------------------------------------------------------------------- trainer
def gen_trainer():
    """trainer.npz: the reference's AlphaZeroTrainer.train (trainer.py:67-161) for 2 epochs on a 16-channel / 1-block net,
    batch 8, no augmentation, fixed seeds: the examples, the batch order its DataLoader used, the policy / value loss of
    every step and every parameter / buffer after the last step."""
    import tempfile
    import torch
    YinYangGame, YinYangLogic, _ = _import_ref()
    from src.yin_yang.ai.neural_network import YinYangNeuralNetwork
    from src.yin_yang.ai.trainer import AlphaZeroTrainer
    torch.set_num_threads(1)
    R = C = 6
    game = YinYangGame(R, C)
    rng = np.random.default_rng(55)
    boards, pis, zs, examples = [], [], [], []
    while len(boards) < 44:
        for arr, pl in random_play_positions(game, rng, 4):
            lb = YinYangLogic(R, C)
            lb.board = arr.copy()
            valid = game.getValidMoves(lb, pl)
            if valid.sum() == 0:
                continue                                   # keep every policy row unique: the batch order is recovered from them
            pi = rng.random(R * C) * valid
            pi = pi / pi.sum()
            z = float(rng.choice([1.0, -1.0, 1e-4]))
            boards.append(arr.astype(np.int8))
            pis.append(pi)
            zs.append(z)
            examples.append((lb, pi, z))
    boards, pis, zs, examples = boards[:44], pis[:44], zs[:44], examples[:44]
    with tempfile.TemporaryDirectory() as tmp:
        trainer = AlphaZeroTrainer(game, model_dir=tmp, lr=0.001, batch_size=8, device=torch.device("cpu"))
        torch.manual_seed(5)
        trainer.nnet = YinYangNeuralNetwork(game, num_channels=16, num_res_blocks=1)     # first thing after the seed
        trainer.optimizer = torch.optim.Adam(trainer.nnet.parameters(), lr=0.001, weight_decay=1e-4)
        init = {k: v.detach().clone().numpy() for k, v in trainer.nnet.state_dict().items()}
        log = dict(p=[], v=[], order=[])
        pol_fn, val_fn = trainer.policy_loss_fn, trainer.value_loss_fn
        table = np.stack(pis).astype(np.float32)

        def policy_loss(logits, policies):
            out = pol_fn(logits, policies)
            log["p"].append(float(out.item()))
            rows = policies.detach().numpy()
            match = [np.flatnonzero((table == r).all(1)) for r in rows]
            assert all(len(m_) == 1 for m_ in match)
            log["order"].append([int(m_[0]) for m_ in match])
            return out

        def value_loss(pred, target):
            out = val_fn(pred, target)
            log["v"].append(float(out.item()))
            return out

        trainer.policy_loss_fn, trainer.value_loss_fn = policy_loss, value_loss
        torch.manual_seed(6)
        metrics = trainer.train(examples, epochs=2, augment=False)
        final = {k: v.detach().clone().numpy() for k, v in trainer.nnet.state_dict().items()}
    steps_per_epoch = (44 + 7) // 8
    order = np.full((2, steps_per_epoch, 8), -1, np.int32)
    for s_, idx in enumerate(log["order"]):
        order[s_ // steps_per_epoch, s_ % steps_per_epoch, : len(idx)] = idx
    out = dict(boards=np.stack(boards), policies=np.stack(pis), values=np.asarray(zs, np.float64), order=order,
               policy_loss=np.asarray(log["p"], np.float64), value_loss=np.asarray(log["v"], np.float64),
               epoch_policy_loss=np.asarray(metrics["policy_loss"]), epoch_value_loss=np.asarray(metrics["value_loss"]),
               epoch_total_loss=np.asarray(metrics["total_loss"]))
    for k, v in init.items():
        out["init/" + k] = v
    for k, v in final.items():
        out["final/" + k] = v
    path = os.path.join(OUT, "trainer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "steps", len(log["p"]), "losses", metrics["total_loss"], flush=True)


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
    if "trainer" in only:
        gen_trainer()
    if "arena" in only:
        gen_arena(pool)
    if "selact" in only:
        gen_select_action()
    if "rowcol" in only:
        gen_rowcol()
    if "g3net800" in only:
        gen_g3_net800(pool)
    sets = [k for k in NET_SEARCH_SETS if k in only or "netsets" in only]
    if sets:
        gen_g3_net_sets(pool, sets)
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

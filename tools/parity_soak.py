#!/usr/bin/env python3
"""Randomised parity soak: for SECONDS, draw a board shape, a simulation count, a board semantics, a batch of random
legal-play root positions (both root players, Dirichlet noise on some), run the HIP search (graph replay for device
evaluators) with the exact hash evaluator and compare visit counts, float32 value sums, priors and final boards with the CPU
oracle, game by game.  Prints one summary line; exits non-zero on the first mismatch.
    python tools/parity_soak.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
import oracle_lib as O
from hash_eval import hash_eval_torch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SHAPES = [(8, 8), (6, 6), (12, 12), (3, 3), (4, 4), (5, 7), (9, 4), (1, 6), (7, 1), (2, 2), (2, 9), (13, 14), (16, 12), (12, 16), (10, 10)]
t0 = time.time()
rounds = games = sims_total = 0
last = t0
while time.time() - t0 < budget:
    R, C = SHAPES[rng.integers(len(SHAPES))]
    A = R * C
    sims = int(rng.choice([1, 2, 3, 7, 25, 64, 150, 400]) if A > 100 else rng.choice([1, 2, 5, 16, 50, 200, 800]))
    copied = int(rng.random() < 0.7)
    G = int(rng.choice([1, 5, 32, 64]))
    pb, vb = [(10, 11), (2, 2), (6, 4)][rng.integers(3)]
    boards = np.zeros((G, R, C), np.int8)
    pl = np.ones(G, np.int8)
    stop = rng.random(G)                                   # per-game stopping probability per ply: roots at all depths
    for ply in range(A):
        m = O.valid_mask(boards, pl)
        go = (rng.random(G) > stop * 0.2) & m.any(1)
        act = np.array([rng.choice(np.flatnonzero(r)) if r.any() else 0 for r in m], np.int32)
        nb, npl, _ = O.next_state(boards, pl, act)
        boards[go], pl[go] = nb[go], npl[go]
    if rng.random() < 0.15:                                # arbitrary (possibly illegal-play) fills as roots too
        boards = rng.integers(-1, 2, size=(G, R, C)).astype(np.int8)
    players = rng.choice(np.array([1, -1], np.int8), size=G)
    use_noise = rng.random() < 0.5
    noise = np.zeros((G, A))
    if use_noise:
        valid = O.valid_mask(boards, players)
        for g in range(G):
            k = int(valid[g].sum())
            if k:
                noise[g, valid[g] == 1] = rng.dirichlet([0.3] * k)
    ctx = pkg.engine.BatchedMCTS(G, R, C, sims, aliased=not copied)
    ev = lambda planes: hash_eval_torch(planes, pb, vb)
    from yinyang_game_alphazero_amd.self_play import LockstepSearch
    ls = LockstepSearch(ctx, ev, use_graph=bool(rng.random() < 0.5))
    ls.run(torch.from_numpy(boards).cuda(), torch.from_numpy(players).cuda(), sims,
           noise=torch.from_numpy(noise).cuda() if use_noise else None)
    c, cw, cp = (t.cpu().numpy() for t in ctx.root_counts(with_children=True))
    fb = ctx.boards().cpu().numpy()
    visits, wsum = ctx.root_stats()
    ctx.status()
    for g in range(G):
        r = O.search_hash(boards[g], int(players[g]), sims, copied, pb, vb, noise=noise[g] if use_noise else None)
        ok = (np.array_equal(c[g], r.counts) and np.array_equal(cw[g].astype(np.float64), r.child_w)
              and np.array_equal(cp[g], r.child_p) and np.array_equal(fb[g], r.final_board)
              and int(visits[g]) == r.root_visits and float(wsum[g]) == r.root_w)
        if not ok:
            print("MISMATCH", dict(shape=(R, C), sims=sims, copied=copied, G=G, g=g, pbits=pb, noise=use_noise), flush=True)
            np.savez("gpurun_out/parity_soak_fail.npz", board=boards[g], player=players[g], noise=noise[g])
            sys.exit(1)
    ctx.close()
    rounds += 1
    games += G
    sims_total += G * sims
    if time.time() - last > 45:
        last = time.time()
        print("[soak] %.0fs: %d batches, %d searches, %d simulations, all equal" % (last - t0, rounds, games, sims_total), flush=True)
print("parity soak ok: %d batches, %d searches, %d simulations compared with the oracle in %.0f s, 0 mismatches"
      % (rounds, games, sims_total, time.time() - t0))

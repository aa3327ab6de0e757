#!/usr/bin/env python3
"""Latency of the single-board reference-API wrappers (host numpy board in, host result out: two PCIe copies + one launch per
call) -- the PCIe-inclusive side of the boundary; the batched engine never takes this path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import yinyang_game_alphazero_amd as pkg
for R in (8, 12):
    game = pkg.YinYangGame(R, R)
    b = game.getInitBoard()
    rng = np.random.default_rng(0)
    for _ in range(R * R // 3):
        v = game.getValidMoves(b, 1)
        idx = np.flatnonzero(v)
        if len(idx) == 0:
            break
        b, _ = game.getNextState(b, 1, int(rng.choice(idx)))
    res = {}
    for name, fn in (("getValidMoves", lambda: game.getValidMoves(b, 1)), ("getGameEnded", lambda: game.getGameEnded(b, 1)),
                     ("getNextState(illegal: no-op)", lambda: game.getNextState(b, 1, 0))):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        res[name] = (time.perf_counter() - t0) / 300 * 1e6
    print(f"{R}x{R}: " + ", ".join(f"{k} {v:.0f} us" for k, v in res.items()))

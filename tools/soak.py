#!/usr/bin/env python3
"""Soak: play N full 8x8 games at G=4096 with the bf16 tower evaluator and check invariants on every example."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
quirks = len(sys.argv) > 3 and sys.argv[3] == "quirks"
torch.manual_seed(0)
game = pkg.YinYangGame(8, 8)
net = pkg.YinYangNeuralNetwork(game).cuda().eval()
eng = pkg.SelfPlayEngine(game, pkg.BatchedEvaluator(net, "bf16"), num_simulations=sims, concurrent_games=4096, seed=3,
                         board_semantics="aliased" if quirks else "copied", reference_quirks=quirks)
t0 = time.perf_counter()
ex = eng.run(N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st, pi, z, gid = (ex[k] for k in ("states", "policies", "values", "game_id"))
n = st.shape[0]
E = pkg.engine
ones = torch.ones(n, dtype=torch.int8, device="cuda")
m1, m2 = E.valid_mask(st, ones), E.valid_mask(st, -ones)
assert eng.games_finished == N and len(torch.unique(gid)) == N
assert torch.allclose(pi.sum(1), torch.ones(n, device="cuda"), atol=1e-5)
if not quirks:
    assert bool((((pi > 0) & ~((m1 | m2) > 0)).sum() == 0))
    s = st.to(torch.int32)
    blk = (s[:, :-1, :-1] == s[:, 1:, :-1]) & (s[:, :-1, :-1] == s[:, :-1, 1:]) & (s[:, :-1, :-1] == s[:, 1:, 1:]) & (s[:, :-1, :-1] != 0)
    assert not bool(blk.any())
vals = torch.unique(z.abs()).tolist()
assert all(min(abs(v - 1), abs(v - 1e-4)) < 1e-6 for v in vals), vals
c = eng.ctx.status()
print(f"soak ok: {N} games, {n} examples ({n/N:.1f}/game), {dt:.1f}s, {n/dt:.0f} positions/s, {c['evals']/dt/1e6:.2f}M expansions/s, "
      f"z mean {float(z.mean()):+.3f}, draws {float((z.abs() < 0.5).float().mean()):.3f}")

#!/usr/bin/env python3
"""Tree-kernel timing split: the unfused select and expand+backup launches (and the fused step) on live trees of
G concurrent 8x8 games, hash evaluator on the device, 800 simulations; torch events around each launch kind."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
from hash_eval import hash_eval_torch
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = 800
ctx = pkg.engine.BatchedMCTS(G, 8, 8, sims)
boards = torch.zeros((G, 8, 8), dtype=torch.int8, device="cuda")
players = torch.ones(G, dtype=torch.int8, device="cuda")
ev = lambda planes: hash_eval_torch(planes, 10, 11)
ctx.begin(boards, players, None)
p, _ = ev(ctx.planes)
ctx.expand_root(p, None, 0.25)
ctx.select()
E = lambda: torch.cuda.Event(enable_timing=True)
t_sel = t_exp = t_fused = 0.0
n_sel = n_exp = n_fused = 0
for s in range(sims - 1):
    p, v = ev(ctx.planes)
    if s % 2 == 0:
        a, b, c = E(), E(), E()
        a.record(); ctx.expand_backup(p, v); b.record(); ctx.select(); c.record()
        torch.cuda.synchronize()
        if s > 20:
            t_exp += a.elapsed_time(b); t_sel += b.elapsed_time(c); n_exp += 1; n_sel += 1
    else:
        a, b = E(), E()
        a.record(); ctx.step(p, v); b.record()
        torch.cuda.synchronize()
        if s > 20:
            t_fused += a.elapsed_time(b); n_fused += 1
c = ctx.status()
print("G=%d: expand+backup %.1f us, select %.1f us, fused step %.1f us (mean over %d/%d launches; mean depth %.2f)" %
      (G, t_exp / n_exp * 1e3, t_sel / n_sel * 1e3, t_fused / n_fused * 1e3, n_exp, n_fused, c["levels"] / max(1, c["evals"] + c["terminal_revisits"])))

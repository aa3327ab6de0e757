#!/usr/bin/env python3
"""Per-launch time of the 8x8 tower+heads kernel for small batches.  YY_TOWER_TB=1|2|4 forces the boards-per-workgroup
variant (read once per process), so run once per variant:  YY_TOWER_TB=2 python tools/tower_small.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
ev = pkg.BatchedEvaluator(net, "bf16")
rng = np.random.default_rng(0)
out = []
for G in (1, 40, 128, 256, 384, 512, 768, 1024, 2048):
    planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
    f = lambda: pkg.engine.tower_heads_forward(planes, ev.towerh_w, ev.towerh_b, ev.tower_layers)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        f()
    t1.record(); torch.cuda.synchronize()
    out.append("G=%d %.1fus" % (G, t0.elapsed_time(t1) / 50 * 1e3))
print("YY_TOWER_TB=%s  " % os.environ.get("YY_TOWER_TB", "auto") + "  ".join(out))

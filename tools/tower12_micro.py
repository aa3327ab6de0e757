#!/usr/bin/env python3
"""12x12 tower+heads per-launch time; YY_TOWER12_Q=1 selects the cout-quarter form (yy_towerq.hip <12,2>)."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(12, 12)).cuda().eval()
ev = pkg.BatchedEvaluator(net, "bf16")
rng = np.random.default_rng(0)
planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 12, 12)).astype(np.int8)).cuda())
f = lambda: pkg.engine.tower_heads_forward(planes, ev.towerh_w, ev.towerh_b, ev.tower_layers)
for _ in range(3):
    o = f()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    f()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 10
fl = (2 * 9 * 16 * 128 * 144 + 20 * 2 * 9 * 128 * 128 * 144 + 2 * 128 * 64 * 144) * G
x = pkg.engine.tower_forward(planes, ev.tower_w, ev.tower_b, ev.tower_layers)
h = hashlib.sha1(o.view(torch.int16).cpu().numpy().tobytes() + x.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
print("YY_TOWER12_Q=%s G=%d: %.1f us/launch, %.1f TFLOP/s, output sha1 %s" % (os.environ.get("YY_TOWER12_Q", "0"), G, ms * 1e3, fl / ms / 1e9, h))

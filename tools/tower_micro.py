#!/usr/bin/env python3
"""Micro-driver for the tower kernel: G boards, random-init 128x10 net, N launches (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
HEADS = len(sys.argv) > 3 and sys.argv[3] == "heads"      # the shipped variant: tower + fused 1x1 head convs
X3 = len(sys.argv) > 3 and sys.argv[3] == "x3"            # split-bf16 tower (f32-grade accuracy)
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
ev = pkg.BatchedEvaluator(net, "bf16")
rng = np.random.default_rng(0)
planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
evx = pkg.BatchedEvaluator(net, "bf16x3") if X3 else None
def launch():
    if X3:
        return pkg.engine.tower_forward_x3(planes, evx.f32_w, evx.f32_b, evx.f32_layers)
    if HEADS:
        return pkg.engine.tower_heads_forward(planes, ev.towerh_w, ev.towerh_b, ev.tower_layers)
    return pkg.engine.tower_forward(planes, ev.tower_w, ev.tower_b, ev.tower_layers)
for _ in range(3):
    launch()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(N):
    launch()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / N
fl = (2 * 9 * 16 * 128 * 64 + 20 * 2 * 9 * 128 * 128 * 64 + (2 * 128 * 64 * 64 if HEADS else 0)) * G
print(f"tower G={G}: {ms*1e3:.1f} us/launch, {fl/ms/1e9:.1f} TFLOP/s")

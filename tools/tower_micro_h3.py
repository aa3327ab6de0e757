#!/usr/bin/env python3
"""Micro-driver for the split-f16 tower kernel (the headline evaluator): G boards, random-init 128x10 net, N launches of
tower + fused head convs on every row (for rocprofv3 --pmc / --kernel-trace).  python tools/tower_micro_h3.py [G] [N] [form] [R] [nb] [boards]
form: g = the general 16x16x32 kernel (csrc/yy_tower_g.hip; nb column blocks, boards per workgroup: default = the large-batch form),
r = the 32x32x16 register-ring kernel (yy_tower_h3r.hip), q = the LDS-ring kernel (yy_tower_h3q.hip)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
form = sys.argv[3] if len(sys.argv) > 3 else "g"
R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, R)).cuda().eval()
ev = pkg.BatchedEvaluator(net, "f16x3r" if form == "r" else "f16x3")
rng = np.random.default_rng(0)
planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, R)).astype(np.int8)).cuda())
if form == "g":
    from yinyang_game_alphazero_amd import network as NN
    (nb, tb), _ = NN.tower_g_forms(R * R, 128)
    if len(sys.argv) > 6:
        nb, tb = int(sys.argv[5]), int(sys.argv[6])
    wq, bq, kw = NN.pack_tower_g(net.cpu()); hw, hb, kh = NN.pack_heads_g(net.cpu())
    wq, bq, hw, hb = wq.cuda(), bq.cuda(), hw.cuda(), hb.cuda()
    fo = torch.empty((G, 2, 32 * R * R), dtype=torch.float32, device="cuda")
    launch = lambda: pkg.engine.tower_g(planes, wq, bq, 21, (kw, kh, NN.ACT_EXP), nb, tb, hw, hb, out=fo)
    form = "g nb=%d boards=%d" % (nb, tb)
elif form == "r":
    launch = lambda: pkg.engine.tower_heads_forward_h3r(planes, ev.h3r_w, ev.h3r_hw, ev.h3_b, ev.h3_layers, ev.h3_exps)
else:
    raise SystemExit("forms: g, r")
for _ in range(3):
    launch()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(N):
    launch()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / N
cells = R * R
fl = (2 * 9 * 5 * 128 * cells + 20 * 2 * 9 * 128 * 128 * cells + 2 * 128 * 64 * cells) * G
print(f"split-f16 tower [{form}] {R}x{R} G={G}: {ms*1e3:.1f} us/launch, {fl/ms/1e9:.1f} TFLOP/s algorithmic ({3*fl/ms/1e9:.0f} issued)")

#!/usr/bin/env python3
"""Evaluator forward time per mode at G boards (8x8, random-init 128x10 net): whole forward (tower + heads + finish) and
the tower launch alone, HIP-event timed on the launch stream, modes interleaved in one process.
    python tools/eval_micro.py [G] [reps] [modes,comma,separated]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16", "f16x3", "bf16x3", "fp32t"]
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
rng = np.random.default_rng(0)
planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
evs = {m: pkg.BatchedEvaluator(net, m) for m in modes}
flags = torch.from_numpy((rng.random(G) < 0.937).astype(np.uint8)).cuda()
def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
conv_flops = (2 * 9 * 16 * 128 * 64 + 20 * 2 * 9 * 128 * 128 * 64) * G
for rnd in range(3):
    for m, ev in evs.items():
        ms = timed(lambda: ev(planes), N)
        line = f"round {rnd} {m:7s} G={G}: forward {ms*1e3:8.1f} us"
        if m == "f16x3":
            tw = timed(lambda: pkg.engine.tower_heads_forward_h3(planes, ev.h3_w, ev.h3_b, ev.h3_layers, ev.h3_exps), N)
            line += f"  [LDS-ring form: tower+headconv {tw*1e3:8.1f} us]"
            tr = timed(lambda: pkg.engine.tower_heads_forward_h3r(planes, ev.h3r_w, ev.h3r_hw, ev.h3_b, ev.h3_layers, ev.h3_exps), N)
            line += f"  [register-ring: tower+headconv {tr*1e3:8.1f} us]"
            mc = timed(lambda: ev(planes, needs_eval=flags), N)
            line += f"  ({conv_flops/tr/1e9:.0f} TFLOP/s algorithmic)  compacted(0.937) forward {mc*1e3:8.1f} us"
        print(line, flush=True)

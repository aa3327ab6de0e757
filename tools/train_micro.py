#!/usr/bin/env python3
"""Micro-driver for the trainer: N optimiser steps of the 128x10 network at batch 64 on synthetic 8x8 examples
(for rocprofv3 --stats).  argv: steps [benchmark 0|1] [graph 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.backends.cudnn.benchmark = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
rng = np.random.default_rng(0)
n = 64 * steps // 8
ex = dict(states=torch.from_numpy(rng.integers(-1, 2, size=(n, 8, 8)).astype(np.int8)),
          policies=torch.from_numpy(rng.dirichlet(np.ones(64), size=n).astype(np.float32)),
          values=torch.from_numpy(rng.choice([-1.0, 1.0], size=n).astype(np.float32)))
torch.manual_seed(0)
tr = pkg.AlphaZeroTrainer(pkg.YinYangGame(8, 8), model_dir="/tmp/yy_train_micro", device="cuda", graph_step=graph)
tr.train(ex, epochs=1)          # warm-up epoch (graph capture, MIOpen find)
torch.cuda.synchronize(); t0 = time.perf_counter()
m = tr.train(ex, epochs=1)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("benchmark=%s graph=%s: %d steps, %.3f ms/step, loss %.4f" % (torch.backends.cudnn.benchmark, graph, steps, dt / steps * 1e3, m["total_loss"][-1]))

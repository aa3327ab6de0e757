#!/usr/bin/env python3
"""Self-play moves of a mid-size batch (default 512 games, 800 simulations, evaluation reuse on): seconds per move with the
kernel form of the split-f16 tower chosen on the device per launch (BatchedEvaluator.auto_form) and with the register-ring form
only.  python tools/small_batch_step.py [games] [moves]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yinyang_game_alphazero_amd as pkg
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.manual_seed(0)
game = pkg.YinYangGame(8, 8)
net = pkg.YinYangNeuralNetwork(game).cuda().eval()
for auto in (True, False, True, False):
    ev = pkg.BatchedEvaluator(net)
    ev.auto_form = auto
    eng = pkg.SelfPlayEngine(game, ev, num_simulations=800, concurrent_games=G, seed=1000)
    bench.stagger_start(eng, 4242)
    for _ in range(2):
        eng.play_move()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = sum(eng.play_move() for _ in range(moves))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = eng.ctx.status()
    print("games %d, form chosen on the device %s: %.3f s per move, %.0f positions/s" % (G, auto, dt / moves, n / dt), flush=True)
    eng.close()

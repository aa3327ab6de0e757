#!/usr/bin/env python3
"""How many evaluator rows of the benchmark workload are RE-expansions of pass nodes (a non-terminal node without legal moves is
re-evaluated on every visit, mcts.py:93-95) -- i.e. what caching their value would save in copied mode."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yinyang_game_alphazero_amd as pkg
import bench
torch.manual_seed(0)
game = pkg.YinYangGame(8, 8)
net = pkg.YinYangNeuralNetwork(game).cuda().eval()
eng = pkg.SelfPlayEngine(game, pkg.BatchedEvaluator(net, "bf16"), num_simulations=800, concurrent_games=4096, seed=1000)
bench.stagger_start(eng, 4242)
eng.play_move()
eng.ctx.reset_counters()
for _ in range(3):
    eng.play_move()
c = eng.ctx.status()
print(c)
print("evals %d, first expansions (nodes created) %d, re-expansions %d = %.2f %% of the evaluator rows; terminal revisits %d"
      % (c["evals"], c["nodes"], c["evals"] - c["nodes"], 100.0 * (c["evals"] - c["nodes"]) / c["evals"], c["terminal_revisits"]))

#!/usr/bin/env python3
"""Wall time of the reference-API call MCTS.search(board, player) (one board, 800 simulations, 8x8) for the
device-resident evaluators, hipGraph replay on/off."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
game = pkg.YinYangGame(8, 8)
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(game).cuda().eval()
for name, ev in (("fp32 nn.Module", net), ("BatchedEvaluator bf16", pkg.BatchedEvaluator(net, "bf16")),
                 ("BatchedEvaluator fp32t", pkg.BatchedEvaluator(net, "fp32t"))):
    for graph in (False, True):
        m = pkg.MCTS(game, ev, num_simulations=800, board_semantics="copied", dirichlet_noise=False)
        m.use_graph = graph
        b = game.getInitBoard()
        m.search(b, 1)                       # warm-up (context, capture)
        t0 = time.perf_counter()
        for _ in range(3):
            pi, root = m.search(b, 1)
        dt = (time.perf_counter() - t0) / 3
        print("%-24s graph=%-5s %.3f s per 800-sim search (%.0f sims/s), root visits %d, argmax %d" %
              (name, graph, dt, 800 / dt, root.visits, int(np.argmax(pi))), flush=True)

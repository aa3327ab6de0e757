import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
from test_gpu_network import _randomized_net, _module_f64
game = pkg.YinYangGame(8, 8)
rng = np.random.default_rng(6)
for rand in (0, 1):
    for blocks in (1, 2, 3, 10):
        for G in (3, 8):
            if rand:
                net = _randomized_net(pkg, game, blocks, 2)
            else:
                torch.manual_seed(0); net = pkg.YinYangNeuralNetwork(game, 128, blocks).cuda().eval()
            planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
            ev = pkg.BatchedEvaluator(net, "f16x3")
            nt = 9 + 36 * (ev.h3_layers - 1)
            x_t = pkg.engine.tower_forward_h3(planes, ev.h3_w[:nt].contiguous(), ev.h3_b[:ev.h3_layers].contiguous(), ev.h3_layers)
            x64, p64, v64 = _module_f64(net, planes)
            p, v = ev(planes)
            evx = pkg.BatchedEvaluator(net, "bf16x3")
            x_x = pkg.engine.tower_forward_x3(planes, evx.f32_w, evx.f32_b, evx.f32_layers)
            d = (x_t.double() - x64).abs()
            print(f"rand {rand} blocks {blocks} G {G}: out-path err {float(d.max()):.3e} (per board {[round(float(d[g].max()),4) for g in range(G)]}) x3 {float((x_x.double()-x64).abs().max()):.3e} "
                  f"scale {float(x64.abs().max()):.3f}  heads-path policy err {float((p.double()-p64).abs().max()):.3e} value err {float((v.double()-v64).abs().max()):.3e}", flush=True)

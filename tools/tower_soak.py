#!/usr/bin/env python3
"""Evaluator soak: for SECONDS, random board size (6/8/12), depth, batch size and BatchNorm statistics; the LDS-resident tower
kernels against the same folded bf16 weights run through torch convolutions (2 bf16 ulps of the layer scale per layer, as in
tests/test_gpu_network.py), the fused head features against torch 1x1 convolutions on the kernel's own activations (2 ulps), and
batch-independence (a board's output bits do not depend on the batch it sits in).  Exits non-zero on the first violation."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); rounds = boards_total = 0; last = t0
while time.time() - t0 < budget:
    R = int(rng.choice([6, 8, 8, 8, 12]))
    blocks = int(rng.choice([1, 2, 3, 5, 10]))
    G = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 31, 64, 65, 127, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1025, 1500]))
    if R == 12:
        G = min(G, 600)
    torch.manual_seed(int(rng.integers(1 << 30)))
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, R), 128, blocks).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.Conv2d):
                m.bias.normal_(0, 0.05)
    boards = torch.from_numpy(rng.integers(-1, 2, size=(G, R, R)).astype(np.int8)).cuda()
    planes = pkg.engine.encode_planes(boards)
    ref = pkg.BatchedEvaluator(net, "bf16", tower=False)
    tow = pkg.BatchedEvaluator(net, "bf16", tower=True, fused_heads=True)
    x_t = pkg.engine.tower_forward(planes, tow.tower_w, tow.tower_b, tow.tower_layers)
    x = planes.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = ref._conv(x, ref.stem, 1)
    for (c1, c2) in ref.blocks:
        y = ref._conv(x, c1, 1)
        x = ref._conv(y, c2, 1, residual=x)
    scale = float(x.float().abs().max()); err = float((x_t.float() - x.float()).abs().max())
    ok = err <= scale * 2.0 ** -7 * (1 + blocks)
    feats = pkg.engine.tower_heads_forward(planes, tow.towerh_w, tow.towerh_b, tow.tower_layers)
    pf = ref._conv(x_t, ref.phead, 0).contiguous().flatten(1).float(); vf = ref._conv(x_t, ref.vhead, 0).contiguous().flatten(1).float()
    fs = max(float(pf.abs().max()), float(vf.abs().max()), 1e-6)
    e_pf, e_vf = float((feats[:, 0].float() - pf).abs().max()), float((feats[:, 1].float() - vf).abs().max())
    ulp = 2.0 ** (int(np.floor(np.log2(fs))) - 7)              # one bf16 ulp at the feature scale
    ok_feats = e_pf <= 2 * ulp and e_vf <= 2 * ulp           # other summation order + one rounding: 2 ulps
    ok &= ok_feats
    g2 = max(1, G // 3)                                      # a sub-batch (usually another workgroup shape): same bits
    f2 = pkg.engine.tower_heads_forward(planes[:g2].contiguous(), tow.towerh_w, tow.towerh_b, tow.tower_layers)
    ok_same = torch.equal(f2.view(torch.int16), feats[:g2].view(torch.int16))
    p, v = tow(planes)
    ok_pv = bool(torch.isfinite(p).all()) and bool(torch.isfinite(v).all()) and float((p.sum(1) - 1).abs().max()) < 1e-4
    ok &= ok_same and ok_pv
    if not ok:
        print("VIOLATION", dict(R=R, blocks=blocks, G=G, err=err, scale=scale, e_pf=e_pf, e_vf=e_vf, fs=fs, ok_feats=ok_feats,
                                ok_same=ok_same, ok_pv=ok_pv, psum=float((p.sum(1) - 1).abs().max())), flush=True)
        sys.exit(1)
    rounds += 1; boards_total += G
    if time.time() - last > 45:
        last = time.time(); print("[tower soak] %.0fs: %d configurations, %d boards, all within bounds" % (last - t0, rounds, boards_total), flush=True)
print("tower soak ok: %d random configurations, %d boards in %.0f s, 0 violations" % (rounds, boards_total, time.time() - t0))

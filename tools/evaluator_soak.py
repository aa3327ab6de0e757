#!/usr/bin/env python3
"""Soak of the float32-accurate evaluator (csrc/yy_tower_g.hip + yy_fc_heads.hip): for SECONDS, random board shape (any R x C up to
144 cells), width (32 / 64 / 96 / 128), depth, batch size, BatchNorm statistics and compaction pattern:
  (1) every instantiated kernel form writes the same head features, bit for bit;
  (2) evaluating the same rows inside a different batch (other size, other neighbours, compacted or dense) returns the same
      (policy, value) bits -- the property the engine's evaluation reuse rests on;
  (3) policy / value against the float32 nn.Module within 1e-5 (north-star bound) on every row.
Exits non-zero on the first violation.   python tools/evaluator_soak.py [seconds] [seed] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
E = pkg.engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/evaluator_soak.json"
rng = np.random.default_rng(seed)
shapes = [(8, 8)] * 4 + [(6, 6), (12, 12), (10, 10), (5, 7), (9, 12), (7, 7), (3, 3), (4, 9), (11, 13), (1, 6), (12, 11)]
t0 = time.time(); rounds = rows_total = 0; last = t0; worst_p = worst_v = 0.0
while time.time() - t0 < budget:
    R, C = shapes[int(rng.integers(len(shapes)))]
    if R * C > 144:
        continue
    ch = int(rng.choice([32, 64, 96, 128, 128, 128]))
    blocks = int(rng.choice([1, 2, 3, 5, 10]))
    G = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 31, 64, 65, 127, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1025, 1500, 2049, 3000]))
    if R * C > 100:
        G = min(G, 700)
    torch.manual_seed(int(rng.integers(1 << 30)))
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, C), ch, blocks).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.1)
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
                m.bias.normal_(0, 0.05)
    ev = pkg.BatchedEvaluator(net)
    assert ev.mode == "f16x3"
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, C)).astype(np.int8)).cuda())
    p, v = (t.clone() for t in ev(planes))
    # (3) against the float32 module
    p32, v32 = net.predict_batch(planes)
    ep, evv = float((p - p32).abs().max()), float((v - v32).abs().max())
    worst_p, worst_v = max(worst_p, ep), max(worst_v, evv)
    ok = ep <= 1e-5 and evv <= 1e-5 and bool(torch.isfinite(p).all())
    # (1) every form
    ref = None
    for nb in E.tower_g_available(ch):
        tb = (16 * nb) // (R * C)
        if tb < 1:
            continue
        f = E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb)
        ref = f if ref is None else ref
        ok = ok and torch.equal(f, ref)
    # (2) the same rows in another batch
    M = int(rng.choice([1, 2, 17, 64, 320, 513, 1024, 2049]))
    idx = torch.from_numpy(rng.integers(0, G, size=M)).cuda()
    flags = torch.from_numpy((rng.random(M) < rng.choice([0.2, 0.9, 1.0])).astype(np.uint8)).cuda()
    p2, v2 = ev(planes[idx].contiguous(), needs_eval=flags, static=bool(rng.integers(2)))
    live = flags.bool()
    ok = ok and torch.equal(p2[live], p[idx][live]) and torch.equal(v2[live], v[idx][live])
    rounds += 1; rows_total += G + M
    if not ok:
        print("VIOLATION", dict(R=R, C=C, ch=ch, blocks=blocks, G=G, M=M, ep=ep, ev=evv), flush=True)
        sys.exit(1)
    if time.time() - last > 30:
        last = time.time()
        print("[soak] %.0f s: %d rounds, %d rows, worst |dp| %.2e |dv| %.2e" % (last - t0, rounds, rows_total, worst_p, worst_v), flush=True)
rec = dict(seconds=time.time() - t0, rounds=rounds, rows=rows_total, worst_policy_err_vs_fp32_module=worst_p, worst_value_err_vs_fp32_module=worst_v,
           violations=0, seed=seed, device=torch.cuda.get_device_name(0))
print(json.dumps(rec))
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(rec, open(out_path, "w"), indent=1)

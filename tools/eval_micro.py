#!/usr/bin/env python3
"""Evaluator forward time per mode at G boards (random-init 128x10 net): the whole forward and its pieces (tower + head convs,
FC heads, finish), dense and with a compacted row list, modes interleaved in ONE process.
    python tools/eval_micro.py [G] [reps] [R] [modes,comma,separated] [live fraction]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
E = pkg.engine
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f16x3", "f16x3r", "bf16"]
frac = float(sys.argv[5]) if len(sys.argv) > 5 else 0.937
torch.manual_seed(0)
net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, R)).cuda().eval()
rng = np.random.default_rng(0)
planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, R)).astype(np.int8)).cuda())
evs = {m: pkg.BatchedEvaluator(net, m) for m in modes}
flags = torch.from_numpy((rng.random(G) < frac).astype(np.uint8)).cuda()


def timed(fn, n=REPS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3          # us


cells = R * R
flops = (2 * 9 * 5 * 128 * cells + 20 * 2 * 9 * 128 * 128 * cells + 2 * 128 * 64 * cells) * G
out = {}
for rnd in range(3):
    for m, ev in evs.items():
        rec = out.setdefault(m, {})
        rec.setdefault("forward_us", []).append(timed(lambda: ev(planes, static=True)))
        if getattr(ev, "supports_compaction", False):
            rec.setdefault("forward_compacted_us", []).append(timed(lambda: ev(planes, needs_eval=flags, static=True)))
        if m == "f16x3":
            nb, tb = ev.g_big if G > ev.g_split else ev.g_small
            feats = torch.empty((G, 2, 32 * cells), dtype=torch.float32, device="cuda")
            rec.setdefault("tower_us", []).append(timed(lambda: E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb, out=feats)))
            lg = torch.empty((G, ev.n_actions), dtype=torch.float32, device="cuda"); hd = torch.empty((G, ev.n_hidden), dtype=torch.float32, device="cuda")
            rec.setdefault("fc_heads_us", []).append(timed(lambda: E.fc_heads(feats, ev.fc_w, ev.fc_b, ev.fc_jobs, ev.n_actions, ev.n_hidden, ev.fc_exps, None, lg, hd)))
            n4 = torch.tensor([max(G // 4, 1)], dtype=torch.int32, device="cuda")
            rec.setdefault("fc_heads_quarter_rows_us", []).append(timed(lambda: E.fc_heads(feats, ev.fc_w, ev.fc_b, ev.fc_jobs, ev.n_actions, ev.n_hidden, ev.fc_exps, n4, lg, hd)))
            pol = torch.empty((G, ev.n_actions), dtype=torch.float32, device="cuda"); val = torch.empty(G, dtype=torch.float32, device="cuda")
            rec.setdefault("head_finish_us", []).append(timed(lambda: E.head_finish_f32(lg, hd, ev.fc2_w, ev.fc2_b, None, None, pol, val)))
            # what the FC heads cost as two library GEMMs (round 2)
            wp, wv = net.policy_fc.weight.t().contiguous(), net.value_fc1.weight.t().contiguous()
            rec.setdefault("fc_heads_as_torch_addmm_us", []).append(timed(lambda: (torch.addmm(net.policy_fc.bias, feats[:, 0], wp), torch.addmm(net.value_fc1.bias, feats[:, 1], wv))))
res = {m: {k: float(np.median(v)) for k, v in rec.items()} for m, rec in out.items()}
for m, rec in res.items():
    if "tower_us" in rec:
        rec["tower_tflops_algorithmic"] = flops / rec["tower_us"] / 1e6
print(json.dumps({"G": G, "R": R, "live_fraction": frac, "median_us": res}, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"G": G, "R": R, "live_fraction": frac, "median_us": res}, open("gpurun_out/eval_micro_%dx%d_%d.json" % (R, R, G), "w"), indent=1)

#!/usr/bin/env python3
"""General split-f16 tower (csrc/yy_tower_g.hip): accuracy against a float64 evaluation of the same network, bit equality between
kernel forms, and an interleaved A/B timing against the 32x32x16 register-ring kernel (yy_tower_h3r.hip) in ONE process.
python tools/tower_g_check.py [G] [rounds]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
from yinyang_game_alphazero_amd import network as N, engine as E

out = {}


def features64(net, planes):
    n64 = pkg.YinYangNeuralNetwork(net.game, net.conv1.out_channels, len(net.res_blocks)).double().eval()
    n64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    with torch.no_grad():
        x = planes.double().cpu()
        x = torch.relu(n64.bn1(n64.conv1(x)))
        for blk in n64.res_blocks:
            x = blk(x)
        p = torch.relu(n64.policy_bn(n64.policy_conv(x))).flatten(1)
        v = torch.relu(n64.value_bn(n64.value_conv(x))).flatten(1)
    return x, torch.stack([p, v], 1)


def check(R, C, ch, blocks, g=96):
    torch.manual_seed(1)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, C), ch, blocks).eval()
    for m in net.modules():                       # non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    rng = np.random.default_rng(R * 100 + C)
    boards = torch.from_numpy(rng.integers(-1, 2, size=(g, R, C)).astype(np.int8)).cuda()
    planes = E.encode_planes(boards)
    wq, bq, kw = N.pack_tower_g(net)
    hw, hb, kh = N.pack_heads_g(net)
    wq, bq, hw, hb = wq.cuda(), bq.cuda(), hw.cuda(), hb.cuda()
    L = 1 + 2 * blocks
    exps = (kw, kh, N.ACT_EXP)
    x64, f64 = features64(net, planes)
    big, small = N.tower_g_forms(R * C, ch)
    res = {}
    feats = []
    for nb, tb in sorted({big, small}):
        act = E.tower_g(planes, wq, bq, L, exps, nb, tb).double().cpu()
        ft = E.tower_g(planes, wq, bq, L, exps, nb, tb, hw, hb)
        torch.cuda.synchronize()
        feats.append(ft)
        scale = float(x64.abs().max())
        res["nb%d_tb%d" % (nb, tb)] = dict(act_err=float((act - x64).abs().max()) / scale, scale=scale,
                                          feat_err=float((ft.double().cpu() - f64).abs().max()) / max(1.0, float(f64.abs().max())))
    res["forms_bit_equal"] = bool(all(torch.equal(feats[0], f) for f in feats[1:]))
    # gather + gate
    rows = torch.from_numpy(rng.permutation(g).astype(np.int32)).cuda()
    n = torch.tensor([g // 2 + 1], dtype=torch.int32).cuda()
    nb, tb = big
    ft2 = E.tower_g(planes, wq, bq, L, exps, nb, tb, hw, hb, rows=rows, n_rows=n)
    torch.cuda.synchronize()
    k = int(n[0])
    res["gather_bit_equal"] = bool(torch.equal(ft2[:k], feats[0][rows[:k].long()]))
    return res


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for shape in ((8, 8, 128, 10), (6, 6, 128, 10), (12, 12, 128, 10), (10, 10, 128, 3), (5, 7, 96, 3), (9, 12, 32, 2), (7, 7, 64, 4), (3, 3, 128, 1), (1, 6, 64, 1)):
        try:
            out["%dx%d_c%d_b%d" % shape] = check(*shape)
        except Exception as e:                                            # noqa
            out["%dx%d_c%d_b%d" % shape] = "ERROR " + repr(e)
        print(shape, out["%dx%d_c%d_b%d" % shape], flush=True)

    # ---- A/B timing at 8x8 (128 x 10), same process, interleaved rounds
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "f16x3")
    wq, bq, kw = N.pack_tower_g(net)
    hw, hb, kh = N.pack_heads_g(net)
    wq, bq, hw, hb = wq.cuda(), bq.cuda(), hw.cuda(), hb.cuda()
    rng = np.random.default_rng(0)
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
    fo = torch.empty((G, 2, 2048), dtype=torch.float32, device="cuda")
    variants = {
        "h3r_32x32x16": lambda: E.tower_heads_forward_h3r(planes, ev.h3r_w, ev.h3r_hw, ev.h3_b, ev.h3_layers, ev.h3_exps, out=fo),
        "g_nb8_tb2": lambda: E.tower_g(planes, wq, bq, 21, (kw, kh, N.ACT_EXP), 8, 2, hw, hb, out=fo),
        "g_nb4_tb1": lambda: E.tower_g(planes, wq, bq, 21, (kw, kh, N.ACT_EXP), 4, 1, hw, hb, out=fo),
    }
    times = {k: [] for k in variants}
    for k, f in variants.items():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for k, f in variants.items():
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                f()
            t1.record(); torch.cuda.synchronize()
            times[k].append(t0.elapsed_time(t1) / 10)
    fl = (2 * 9 * 5 * 128 * 64 + 20 * 2 * 9 * 128 * 128 * 64 + 2 * 128 * 64 * 64) * G
    out["timing_ms"] = {k: dict(median=float(np.median(v)), min=float(np.min(v)), tflops_alg=fl / np.median(v) / 1e9) for k, v in times.items()}
    print(json.dumps(out["timing_ms"], indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/tower_g_check.json", "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Build / run compile-time variants of csrc/yy_tower_g.hip side by side (debugging and A/B timing in ONE process).
  python tools/tg_variants.py build            (CPU container: hipcc, one .so per variant under tools/tgv/)
  python tools/tg_variants.py run [G]          (GPU box)"""
import ctypes as ct, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
# name -> extra hipcc flags; add an entry per experiment macro placed in csrc/yy_tower_g.hip (the round-3 experiments -- full fragment
# window, lo-before-hi reads, unrolled channel groups, ring depth 3, epilogue fused into a block-major last group, one dual-form
# launch -- were built this way, measured against "base" in one process and removed again: DESIGN.md section 3)
VARIANTS = {"base": []}
STUB = os.path.join(HERE, "tgv", "stub.cpp")


def build():
    os.makedirs(os.path.join(HERE, "tgv"), exist_ok=True)
    open(STUB, "w").write('#include <stdio.h>\nextern "C" int yy_tower_set_err(int c, const char *m) { fprintf(stderr, "tower error %d: %s\\n", c, m); return c; }\n')
    procs = []
    for name, flags in VARIANTS.items():
        so = os.path.join(HERE, "tgv", "libtg_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
               "-o", so, os.path.join(ROOT, "yinyang-game-alphazero_amd", "csrc", "yy_tower_g.hip"), STUB] + flags
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        assert p.wait() == 0, name
        print("built", name)


def run():
    import numpy as np, torch
    import yinyang_game_alphazero_amd as pkg
    from yinyang_game_alphazero_amd import network as N, engine as E
    from tower_g_check import features64
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    vp = ct.c_void_p
    libs = {}
    for name in VARIANTS:
        L = ct.CDLL(os.path.join(HERE, "tgv", "libtg_%s.so" % name))
        L.yy_nn_tower_g.argtypes = [vp] * 9 + [ct.c_int] * 12 + [vp]
        libs[name] = L

    def call(L, planes, wq, hw, bq, hb, out, heads, R, C, ch, nl, exps, nb, tb):
        G_ = planes.shape[0]
        p = lambda t: None if t is None else vp(t.data_ptr())
        rc = L.yy_nn_tower_g(p(planes), p(wq), p(hw), p(bq), p(hb), None if heads else p(out), p(out) if heads else None, None, None,
                             G_, R, C, ch, nl, exps[0], exps[1], exps[2], nb, tb, -1, 0x7FFFFFFF, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    res = {}
    # correctness: 8x8, 128 ch, 1 block, forms nb 4..9 (tb = 16*nb // 64)
    torch.manual_seed(1)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8), 128, 1).eval()
    rng = np.random.default_rng(5)
    g = 24
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(g, 8, 8)).astype(np.int8)).cuda())
    wq, bq, kw = N.pack_tower_g(net); hw, hb, kh = N.pack_heads_g(net)
    wq, bq, hw, hb = wq.cuda(), bq.cuda(), hw.cuda(), hb.cuda()
    x64, f64 = features64(net, planes)
    for name, L in libs.items():
        for nb in (4, 5, 6, 7, 8, 9):
            tb = (16 * nb) // 64
            out = torch.zeros((g, 8, 8, 128), dtype=torch.float32, device="cuda")
            call(L, planes, wq, None, bq, None, out, False, 8, 8, 128, 3, (kw, kh, N.ACT_EXP), nb, tb)
            torch.cuda.synchronize()
            err = (out.permute(0, 3, 1, 2).double().cpu() - x64).abs()          # [g, ch, 8, 8]
            per_board = err.amax(dim=(1, 2, 3))
            per_cell = err.amax(dim=(0, 1)).reshape(-1)
            res["%s_nb%d" % (name, nb)] = dict(max=float(err.max()), boards_bad=[int(i) for i in torch.nonzero(per_board > 1e-4).reshape(-1)[:12]],
                                               cells_bad=[int(i) for i in torch.nonzero(per_cell > 1e-4).reshape(-1)[:64]])
            print(name, nb, tb, res["%s_nb%d" % (name, nb)], flush=True)
    # timing: 8x8 128x10, nb 8 / tb 2
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).eval()
    wq, bq, kw = N.pack_tower_g(net); hw, hb, kh = N.pack_heads_g(net)
    wq, bq, hw, hb = wq.cuda(), bq.cuda(), hw.cuda(), hb.cuda()
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
    fo = torch.empty((G, 2, 2048), dtype=torch.float32, device="cuda")
    times = {k: [] for k in libs}
    for r in range(6):
        for name, L in libs.items():
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                call(L, planes, wq, hw, bq, hb, fo, True, 8, 8, 128, 21, (kw, kh, N.ACT_EXP), 8, 2)
            t1.record(); torch.cuda.synchronize()
            if r:
                times[name].append(t0.elapsed_time(t1) / 10)
    res["timing_ms"] = {k: dict(median=float(np.median(v)), min=float(np.min(v))) for k, v in times.items()}
    print(json.dumps(res["timing_ms"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tg_variants.json"), "w"), indent=1)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()

"""Duration of ONE evaluator tower launch against the number of rows it carries, for the one-board and the two-board form of
k_tower_g (8x8, 128x10): alone on the chip and with the same launch on a second stream (what two lanes do).
    python tools/tower_rows_sweep.py [out.json]
How network.G_SPLIT_WG was looked at: the per-row cost of the two forms and the size of a round of workgroups."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import yinyang_game_alphazero_amd as pkg
from yinyang_game_alphazero_amd import engine


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    game = pkg.YinYangGame(8, 8)
    net = pkg.YinYangNeuralNetwork(game, 128, 10).to(dev).eval()
    ev = pkg.BatchedEvaluator(net, "f16x3")
    G = 4096
    planes = (torch.rand(G, 5, 8, 8, device=dev) < 0.4).float()
    rows = torch.arange(G, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = [torch.empty((G, 2, 32 * 64), dtype=torch.float32, device=dev) for _ in streams]
    res = {"forms": {"small": ev.g_small, "big": ev.g_big}, "us": {}}

    def launch(form, n_dev, k):
        engine.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, form[0], form[1],
                       head_w=ev.g_hw, head_bias=ev.g_hb, rows=rows, n_rows=n_dev, out=outs[k])

    for n in (64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1536, 2048, 4096):
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        rec = {}
        for name, form in (("small", ev.g_small), ("big", ev.g_big)):
            for two in (False, True):
                ts = []
                for it in range(8):
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(streams[0]):
                        e0.record()
                        for _ in range(4):
                            launch(form, n_dev, 0)
                        e1.record()
                    if two:
                        with torch.cuda.stream(streams[1]):
                            for _ in range(4):
                                launch(form, n_dev, 1)
                    torch.cuda.synchronize()
                    if it >= 2:
                        ts.append(e0.elapsed_time(e1) / 4 * 1e3)
                ts.sort()
                rec[name + ("_two_streams" if two else "")] = round(ts[len(ts) // 2], 1)
        res["us"][n] = rec
        print(n, rec, flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Rows per evaluator launch on the evaluation-reuse leg of bench.py (8x8, 800 sims, 4096 games, two lanes, 8-stone book):
plays warm-up moves with the captured graphs, then ONE move per lane eagerly with every compacted row count recorded.
    python tools/reuse_rows_hist.py [out.json] [warm moves]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import yinyang_game_alphazero_amd as pkg
from yinyang_game_alphazero_amd import engine
from yinyang_game_alphazero_amd.self_play import SelfPlayLanes


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    game = pkg.YinYangGame(8, 8)
    net = pkg.YinYangNeuralNetwork(game, 128, 10).to(dev).eval()
    ev = pkg.BatchedEvaluator(net, "f16x3")
    book = pkg.engine.OpeningBook(8, 8, ev, 8, device=dev)
    eng = SelfPlayLanes(game, ev, num_simulations=800, concurrent_games=4096, opening_book=book, lanes=2, seed=1000, device=dev,
                        reuse_pass_value=True, reuse_transpositions=True, keep_evaluations=True)
    bench.stagger_start(eng, 4242)
    for _ in range(warm):
        eng.play_move()
    seen = []
    real = engine.compact_rows

    def spy(needs_eval, rows=None, n=None):
        r = real(needs_eval, rows, n)
        seen.append(r[1].clone())
        return r

    engine.compact_rows = spy
    out = {}
    for k, lane in enumerate(eng.lanes):
        lane.search.use_graph = False
        seen.clear()
        with lane._on_stream():
            lane.play_move()
        torch.cuda.synchronize()
        v = torch.cat(seen).cpu().numpy().astype(np.int64)
        edges = [0, 1, 64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1536, 2049]
        hist, _ = np.histogram(v, bins=edges)
        out["lane%d" % k] = {"launches": int(v.size), "mean_rows": float(v.mean()), "max": int(v.max()),
                             "hist_edges": edges, "hist": hist.tolist(),
                             "mean_rows_by_100_sims": [float(v[i:i + 100].mean()) for i in range(0, v.size, 100)],
                             "share_of_rows_by_bucket": (np.histogram(v, bins=edges, weights=v)[0] / max(v.sum(), 1)).round(3).tolist()}
        print(k, out["lane%d" % k], flush=True)
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()

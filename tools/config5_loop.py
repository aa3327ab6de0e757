#!/usr/bin/env python3
"""BASELINE.json config 5: the full loop (GPU self-play -> PyTorch-ROCm training -> arena -> promote at >= 0.6) for N
iterations on one GPU, per-phase wall time written after every iteration.

    python tools/config5_loop.py --iterations 10 --episodes 512 --out gpurun_out/config5.json

Trainer settings are the reference's (Adam 1e-3, wd 1e-4, batch 64, 10 epochs, 8-fold augmentation, 10 000 sampled
examples per iteration); models/data go to a scratch directory, the history JSON and the final best model to --out's
directory."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--cols", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--episodes", type=int, default=512)
    ap.add_argument("--simulations", type=int, default=800)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--arena-games", type=int, default=40)
    ap.add_argument("--nn", default="auto", help="auto = float32-accurate (f16x3); bf16 = reduced precision")
    ap.add_argument("--out", default="gpurun_out/config5.json")
    a = ap.parse_args()
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    work = tempfile.mkdtemp(prefix="yy_cfg5_")
    game = pkg.YinYangGame(a.rows, a.cols)
    az = pkg.AlphaZero(game, model_dir=os.path.join(work, "models"), data_dir=os.path.join(work, "data"),
                       num_iterations=1, num_episodes=a.episodes, num_simulations=a.simulations, num_epochs=a.epochs,
                       arena_games=a.arena_games, nn_mode=a.nn, concurrent_games=min(4096, a.episodes),
                       batch_size=a.batch_size)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    t0 = time.perf_counter()
    for it in range(a.iterations):
        az.run()                                              # one iteration; history accumulates in az.history
        h = az.history[-1]
        h["iteration"] = it + 1
        n_ex = len(pkg.training.load_examples(h["data_file"])["values"])
        h["examples"] = int(n_ex)
        h["data_file"] = os.path.basename(h["data_file"])
        print("[config5] it %d  self-play %.1fs (%d examples, %.0f pos/s)  train %.1fs (loss %.4f -> %.4f)  arena %.1fs "
              "(win ratio %.3f, promoted %s)  total %.0fs" % (it + 1, h["self_play_s"], n_ex, n_ex / h["self_play_s"],
              h["train_s"], h["losses"][0], h["losses"][-1], h["arena_s"], h["win_ratio"], h["promoted"],
              time.perf_counter() - t0), flush=True)
        with open(a.out, "w") as f:
            json.dump(dict(config=vars(a), device=torch.cuda.get_device_name(0), wall_s=time.perf_counter() - t0,
                           iterations=az.history), f, indent=1)
    shutil.copy(az.best_model_path, os.path.join(os.path.dirname(os.path.abspath(a.out)), "config5_best_model.pth.tar"))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()

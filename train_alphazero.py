#!/usr/bin/env python3
"""Same-flag entry point as the reference's train_alphazero.py (train_alphazero.py:30-61), running
`--mode self-play` on the MI355X engine.

    python train_alphazero.py --mode self-play --rows 8 --cols 8 --simulations 800 --episodes 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_alphazero.py --mode self-play ...

Flags added to the reference's set: --concurrent-games, --lanes, --board-semantics {copied,aliased}, --reference-quirks,
--nn {auto,f16x3,bf16,fp32,fp32t}, --evaluation-reuse, --opening-book-stones, --seed, --arena-games, --channels, --blocks, --fresh,
--dist-backend, --reference-format.  `--mode train` runs
the iteration loop (GPU self-play -> PyTorch-ROCm training -> batched arena -> promote at 0.6) and
`--mode evaluate` plays 10 games against RandomPlayer, like the reference's modes.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Yin-Yang AlphaZero self-play on MI355X")
    p.add_argument("--rows", type=int, default=8)
    p.add_argument("--cols", type=int, default=8)
    p.add_argument("--iterations", type=int, default=100)
    p.add_argument("--episodes", type=int, default=100)
    p.add_argument("--simulations", type=int, default=800)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--workers", type=int, default=1)
    p.add_argument("--mcts-threads", type=int, default=1)
    p.add_argument("--model-dir", type=str, default="models")
    p.add_argument("--data-dir", type=str, default="data")
    p.add_argument("--resume", action="store_true",
                   help="accepted for compatibility and inert, like the reference's (train_alphazero.py:55 parses it, nothing reads it): "
                        "models and checkpoint_<n> files found in --model-dir are ALWAYS continued from (alphazero.py:66-73, "
                        "training_pipeline.py:171-190); use --fresh to start over")
    p.add_argument("--fresh", action="store_true",
                   help="train: remove current_model / best_model / checkpoint_<n> from --model-dir first (start from a new random network)")
    p.add_argument("--mode", choices=["train", "self-play", "evaluate"], default="train")
    p.add_argument("--output-model", type=str, default="best_model.pth.tar")
    # engine flags
    p.add_argument("--concurrent-games", type=int, default=4096)
    p.add_argument("--board-semantics", choices=["copied", "aliased"], default="copied")
    p.add_argument("--reference-quirks", action="store_true", help="reproduce Q4/Q5 of the reference's play_game")
    p.add_argument("--nn", choices=["auto", "f16x3", "bf16", "fp32", "fp32t"], default="auto",
                   help="evaluator: auto = float32-accurate (f16x3 split-f16 tower where the kernels cover the shape, else the fp32 module; the "
                        "reference evaluates in float32); bf16 = reduced-precision fast tower; fp32t exact-f32 tower (8x8, 128 channels)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--evaluation-reuse", choices=["auto", "off"], default="auto",
                   help="auto: the network is asked once per position of a game (pass values + per-game evaluation cache; "
                        "identical games); off: once per leaf, like the reference")
    p.add_argument("--reference-format", action="store_true",
                   help="self-play: also store the reference's pickled board objects so its own training pipeline reads the file")
    p.add_argument("--opening-book-stones", type=int, default=None,
                   help="shared opening book: every position reachable with at most N stones is evaluated once before play "
                        "(default: 8 when evaluation reuse is on, the board has at most 64 cells and the rank plays >= 1024 games; 0 = none)")
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="collectives of a multi-rank launch: nccl = RCCL over xGMI (one rank per GPU); gloo = over the host, "
                        "which also allows several ranks to share one GPU (rehearsals on a 1-GPU box)")
    p.add_argument("--lanes", type=int, default=0,
                   help="self-play: HIP streams the rank's concurrent games are cut over (0 = automatic: 2 from 512 games on); a lane's "
                        "evaluator launch fills the compute units the other lane's last round of workgroups leaves idle; same games")
    p.add_argument("--arena-games", type=int, default=40)
    p.add_argument("--channels", type=int, default=128)
    p.add_argument("--blocks", type=int, default=10)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("train_alphazero.py: no ROCm device -- the self-play engine has no CPU fallback")
    if args.dist_backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(args.rows, args.cols)
    for d in (args.model_dir, args.data_dir):
        os.makedirs(d, exist_ok=True)
    if args.mode == "train" and args.fresh:
        import glob
        if rank == 0:
            for f in glob.glob(os.path.join(args.model_dir, "checkpoint*.pth.tar")) + [os.path.join(args.model_dir, n) for n in
                                                                                     ("current_model.pth.tar", "best_model.pth.tar")]:
                if os.path.exists(f):
                    os.remove(f)
        if world > 1:
            dist.barrier()
    if args.mode == "train":                     # train_alphazero.py:84-101
        az = pkg.AlphaZero(game, args.model_dir, args.data_dir, num_iterations=args.iterations, num_episodes=args.episodes,
                           num_simulations=args.simulations, num_epochs=args.epochs, num_workers=args.workers,
                           mcts_threads=args.mcts_threads, nn_mode=args.nn, concurrent_games=args.concurrent_games,
                           arena_games=args.arena_games, num_channels=args.channels, num_res_blocks=args.blocks,
                           lr=args.lr, batch_size=args.batch_size)
        hist = az.run()
        if rank == 0:
            print(json.dumps({"iterations": hist}))
        if world > 1:
            dist.destroy_process_group()
        return
    model_path = os.path.join(args.model_dir, args.output_model)
    if args.mode == "evaluate":                  # train_alphazero.py:124-243: 10 games against RandomPlayer
        if not os.path.exists(model_path):
            sys.exit(f"Model file not found: {model_path}")
        res = pkg.evaluate_vs_random(game, model_path, num_games=10, num_simulations=args.simulations, nn_mode=args.nn,
                                     num_channels=args.channels, num_res_blocks=args.blocks)
        print(json.dumps(res))
        return
    if not os.path.exists(model_path):           # same contract as the reference (train_alphazero.py:107-109)
        sys.exit(f"Model file not found: {model_path}")
    path = pkg.generate_self_play_data(game, model_path, args.data_dir, num_games=args.episodes,
                                       num_workers=args.workers, num_simulations=args.simulations,
                                       concurrent_games=args.concurrent_games, board_semantics=args.board_semantics,
                                       reference_quirks=args.reference_quirks, nn_mode=args.nn, seed=args.seed,
                                       num_channels=args.channels, num_res_blocks=args.blocks,
                                       reference_format=args.reference_format,
                                       evaluation_reuse=None if args.evaluation_reuse == "auto" else False,
                                       opening_book_stones=args.opening_book_stones, lanes=args.lanes or None)
    if rank == 0:
        st = pkg.generate_self_play_data.last_stats
        st = dict(st, positions_per_s=st["positions"] / st["seconds"], expansions_per_s=st["evals"] / st["seconds"])
        print(json.dumps({"data_file": path, "rank0_stats": st}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

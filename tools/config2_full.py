#!/usr/bin/env python3
"""BASELINE.json config 2 run TO COMPLETION: 8x8, 800 sims, 4096 concurrent games, frozen seeded 128x10 net, every
game played to its end (~53 plies each).  Whole-run wall time includes the ragged tail (games end at
different plies, the batch empties).  Also a 2x oversubscribed run (8192 games through 4096 slots) where finished
slots are refilled, the way a production iteration keeps the batch full.
    python tools/config2_full.py [--games 4096] [--nn bf16] [--out gpurun_out/config2_full.json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yinyang_game_alphazero_amd as pkg
from yinyang_game_alphazero_amd.self_play import SelfPlayLanes

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--slots", type=int, default=4096)
ap.add_argument("--sims", type=int, default=800)
ap.add_argument("--nn", default="auto", help="auto = float32-accurate (f16x3); bf16 = reduced precision")
ap.add_argument("--rows", type=int, default=8)
ap.add_argument("--single", action="store_true", help="only the plain run (no 2x refilled run)")
ap.add_argument("--reuse", type=int, default=1, help="1 = the engine's evaluation reuse (its default); 0 = the evaluator gets every row the reference evaluates")
ap.add_argument("--book", type=int, default=0, help="shared opening book: positions with at most this many stones (0 = none)")
ap.add_argument("--lanes", type=int, default=2, help="HIP streams the slots are cut over (self_play.SelfPlayLanes)")
ap.add_argument("--cols", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/config2_full.json")
ap.add_argument("--split-wg", type=int, default=0, help="override network.G_SPLIT_WG (A/B)")
ap.add_argument("--hint-mult", type=float, default=0, help="override network.G_HINT_BIG_ONLY (A/B)")
ap.add_argument("--form", default="auto", choices=["auto", "big"], help="big = the evaluator always launches the large tower form only (A/B)")
ap.add_argument("--free-running", action="store_true", help="each lane's next move is enqueued as soon as its last one was read back, instead of all lanes starting every move together as SelfPlayLanes.run() does (A/B: how run() was decided)")
a = ap.parse_args()
if a.hint_mult:
    from yinyang_game_alphazero_amd import network as _net
    _net.G_HINT_BIG_ONLY = a.hint_mult
if a.form == "big":
    from yinyang_game_alphazero_amd import network as _net
    _real_hint = _net.BatchedEvaluator.rows_hint
    _net.BatchedEvaluator.rows_hint = lambda self, owner, mean_rows: _real_hint(self, owner, 1e9)
if a.split_wg:
    from yinyang_game_alphazero_amd import network as _net
    _net.G_SPLIT_WG = a.split_wg
torch.manual_seed(0)
game = pkg.YinYangGame(a.rows, a.cols or a.rows)
net = pkg.YinYangNeuralNetwork(game).cuda().eval()
ev = pkg.BatchedEvaluator(net, a.nn)
res = []
for total in ((a.games,) if a.single else (a.games, 2 * a.games)):
    kw = {} if a.reuse else dict(reuse_pass_value=False, reuse_transpositions=False, keep_evaluations=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()          # the book's build time is inside the wall time
    eng = SelfPlayLanes(game, ev, num_simulations=a.sims, concurrent_games=a.slots, lanes=a.lanes, seed=1000, opening_book=a.book, **kw)
    last = [t0]

    def progress(e):
        if time.perf_counter() - last[0] > 30:
            last[0] = time.perf_counter()
            print("[config2] %d games, %.0f s, %d slots alive" % (total, last[0] - t0, e.n_alive), flush=True)
    if a.free_running:
        from yinyang_game_alphazero_amd.self_play import shard_games
        for k, l in enumerate(eng.lanes):
            l.begin_run(shard_games(total, k, len(eng.lanes))[0])
        live = [l for l in eng.lanes if l.n_alive > 0]
        for l in live:
            l.enqueue_move()
        while live:
            for l in list(live):
                l.finish_move()
                if l.n_alive > 0:
                    l.enqueue_move()
                else:
                    live.remove(l)
            progress(eng)
        eng.ctx.status()
        ex = eng.collect()
    else:
        ex = eng.run(total, progress=progress)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = eng.ctx.status()
    n = int(ex["values"].shape[0])
    plies = torch.bincount(ex["game_id"] - ex["game_id"].min()).float()
    r = dict(board=f"{a.rows}x{a.cols or a.rows}", games=total, slots=a.slots, lanes=a.lanes, sims=a.sims, nn=a.nn, wall_s=dt, positions=n, positions_per_s=n / dt,
             evaluation_reuse=bool(a.reuse), opening_book_stones=a.book, book_positions=(eng.book.n if eng.book is not None else 0), evaluator_rows=int(c["evals"]), evaluator_rows_per_s=c["evals"] / dt,
             pass_values_reused=int(c["reused_values"]), cache_hits=int(c["transposition_hits"]),
             simulations_per_s=n * a.sims / dt, plies_mean=float(plies.mean()),
             plies_min=int(plies.min()), plies_max=int(plies.max()),
             z_counts={str(v): int((ex["values"] == v).sum()) for v in (1.0, -1.0)} | {"draw": int((ex["values"].abs() < 0.5).sum())})
    print(json.dumps(r), flush=True)
    res.append(r)
    eng.close()
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(dict(device=torch.cuda.get_device_name(0), runs=res), open(a.out, "w"), indent=1)

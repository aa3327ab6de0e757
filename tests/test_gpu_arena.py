"""GPU: arena and the iteration loop on the batched engine."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_arena_accounting_and_symmetry():
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(game, 16, 1).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "fp32")
    # a network against itself: every game is decided, colours alternate, totals add up
    res = pkg.Arena(game, ev, ev, num_simulations=12, seed=1).play(24)
    assert res["a_wins"] + res["b_wins"] + res["draws"] == 24 == res["games"]
    # against the uniformly random player
    res2 = pkg.Arena(game, ev, "random", num_simulations=12, seed=2).play(16)
    assert res2["a_wins"] + res2["b_wins"] + res2["draws"] == 16
    # the reference's literal attribution is available as a switch; totals still add up
    res4 = pkg.Arena(game, ev, "random", num_simulations=12, seed=2, reference_scoring=True).play(16)
    assert res4["a_wins"] + res4["b_wins"] + res4["draws"] == 16
    # random vs random needs no search at all
    res3 = pkg.Arena(game, "random", "random", num_simulations=1, seed=3).play(32)
    assert res3["a_wins"] + res3["b_wins"] + res3["draws"] == 32 and res3["a_wins"] > 0 and res3["b_wins"] > 0


def test_arena_graph_replay_equals_routed_eager():
    """A small match runs both players on all rows and replays the simulation step from a hipGraph; a large one routes
    rows to their own network eagerly.  With evaluators whose output depends on the row only (exact hash evaluators)
    both must play the same games."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    from yinyang_game_alphazero_amd import arena as arena_mod
    from hash_eval import hash_eval_torch
    game = pkg.YinYangGame(6, 6)
    ev_a = lambda planes: hash_eval_torch(planes, 10, 11)
    ev_b = lambda planes: hash_eval_torch(planes, 6, 4)
    res_graph = pkg.Arena(game, ev_a, ev_b, num_simulations=30, seed=5).play(10)

    class Routed(arena_mod.DualEvaluator):
        def __init__(self, *a, **k):
            k["dense"] = False
            super().__init__(*a, **k)

    class EagerSearch(arena_mod.LockstepSearch):
        def __init__(self, ctx, ev, use_graph=True):
            super().__init__(ctx, ev, use_graph=False)

    saved = arena_mod.DualEvaluator, arena_mod.LockstepSearch
    arena_mod.DualEvaluator, arena_mod.LockstepSearch = Routed, EagerSearch
    try:
        res_eager = pkg.Arena(game, ev_a, ev_b, num_simulations=30, seed=5).play(10)
    finally:
        arena_mod.DualEvaluator, arena_mod.LockstepSearch = saved
    assert res_graph == res_eager and res_graph["games"] == 10


def test_trainer_graph_step_matches_eager():
    """The hipGraph-replayed optimiser step (single process, full batches) against the eager loop: same data, same
    permutations, same initial weights -> same losses (tolerance 2e-3 relative: the captured step may use other MIOpen
    convolution algorithms, i.e. another summation order, than the eager one)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(6, 6)
    rng = np.random.default_rng(0)
    n = 96
    ex = dict(states=torch.from_numpy(rng.integers(-1, 2, size=(n, 6, 6)).astype(np.int8)),
              policies=torch.from_numpy(rng.dirichlet(np.ones(36), size=n).astype(np.float32)),
              values=torch.from_numpy(rng.choice([-1.0, 1.0, 1e-4], size=n).astype(np.float32)))
    losses = []
    for graph in (True, False):
        torch.manual_seed(3)
        tr = pkg.AlphaZeroTrainer(game, model_dir="/tmp/yy_graph_step_test", batch_size=32, device="cuda",
                                  num_channels=16, num_res_blocks=2, graph_step=graph)
        torch.manual_seed(4)                      # the epoch permutations
        m = tr.train(ex, epochs=3, augment=True)  # 768 samples = 24 full batches per epoch
        assert (tr._graph is not None) == graph
        losses.append(m["total_loss"])
    a, b = np.asarray(losses[0]), np.asarray(losses[1])
    assert np.all(np.abs(a - b) <= 2e-3 * np.abs(b)), (a, b)
    assert a[-1] < a[0]


def test_alphazero_one_iteration(tmp_path):
    """config 5 in miniature: GPU self-play -> training -> arena -> promote decision; files keep the
    reference's names and checkpoint format."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(0)
    az = pkg.AlphaZero(game, model_dir=str(tmp_path / "models"), data_dir=str(tmp_path / "data"), num_iterations=1,
                       num_episodes=12, num_simulations=10, num_epochs=2, arena_games=8, nn_mode="fp32",
                       num_channels=16, num_res_blocks=1, concurrent_games=12)
    hist = az.run()
    assert len(hist) == 1 and 0.0 <= hist[0]["win_ratio"] <= 1.0
    for name in ("current_model.pth.tar", "best_model.pth.tar", "checkpoint_1.pth.tar"):
        ck = torch.load(str(tmp_path / "models" / name), map_location="cpu", weights_only=True)
        assert set(ck) == {"state_dict", "board_size", "action_size"}
    z = np.load(hist[0]["data_file"])
    assert z["states"].shape[0] == z["policies"].shape[0] == z["values"].shape[0] > 12 * 5
    assert len(hist[0]["losses"]) == 2
    # evaluate mode: the trained model against RandomPlayer
    r = pkg.evaluate_vs_random(game, str(tmp_path / "models" / "best_model.pth.tar"), num_games=6, num_simulations=8,
                               nn_mode="fp32", num_channels=16, num_res_blocks=1)
    assert r["alphazero_wins"] + r["random_wins"] + r["draws"] == 6


def test_alphazero_iteration_at_config5_size(tmp_path):
    """BASELINE config 5 at its real size for ONE iteration with few episodes: 8x8 board, 800 simulations per move, the 128x10
    network, the default (float32-accurate) evaluator with the engine's evaluation reuse, 10 000-example sampling cut to what the
    games produce: GPU self-play -> training -> arena (800 simulations) -> promotion rule.  Checks what a run must hold: every
    game contributes its positions with pi summing to 1 and z in {-1, 1, +-1e-4}, the loss falls, the arena accounts for every
    game, the promotion follows the 0.6 rule, the checkpoints load into the reference's layout."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(8, 8)
    torch.manual_seed(0)
    az = pkg.AlphaZero(game, model_dir=str(tmp_path / "models"), data_dir=str(tmp_path / "data"), num_iterations=1,
                       num_episodes=32, num_simulations=800, num_epochs=2, arena_games=4, concurrent_games=32)
    hist = az.run()
    h = hist[0]
    z = np.load(h["data_file"])
    n = z["states"].shape[0]
    assert len(np.unique(z["game_id"])) == 32 and 32 * 30 <= n <= 32 * 64
    assert np.allclose(z["policies"].sum(1), 1.0, atol=1e-6) and set(np.unique(np.round(z["values"], 4))) <= {-1.0, 1.0, 0.0001, -0.0001}
    assert (np.abs(z["states"]) <= 1).all() and (np.count_nonzero(z["states"].reshape(n, -1), axis=1) <= z["ply"]).all()   # a ply places at most one stone
    assert len(h["losses"]) == 2 and h["losses"][1] < h["losses"][0] and np.isfinite(h["losses"]).all()
    a = h["arena"]
    assert a["a_wins"] + a["b_wins"] + a["draws"] == a["games"] == 4
    assert h["promoted"] == (h["win_ratio"] >= 0.6)
    for name in ("current_model.pth.tar", "best_model.pth.tar", "checkpoint_1.pth.tar"):
        ck = torch.load(str(tmp_path / "models" / name), map_location="cpu", weights_only=True)
        assert set(ck) == {"state_dict", "board_size", "action_size"} and tuple(ck["board_size"]) == (8, 8)
    print("config-5-size iteration: self-play %.1f s (%d examples), train %.1f s (loss %.3f -> %.3f), arena %.1f s, win ratio %.2f" % (
        h["self_play_s"], n, h["train_s"], h["losses"][0], h["losses"][1], h["arena_s"], h["win_ratio"]))


def test_arena_with_evaluation_reuse_plays_the_same_match():
    """Two different 128x2 networks, the float32-accurate evaluator, 8x8, 6 games of 800 simulations: with the arena's
    within-search evaluation reuse (pass values + positions reached by another move order; never across searches, the two
    networks alternate) every move of every game and the result are the same as without it."""
    import time
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(8, 8)
    evs = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        evs.append(pkg.BatchedEvaluator(pkg.YinYangNeuralNetwork(game, 128, 2).cuda().eval()))
    out = []
    for reuse in (True, False):
        arena = pkg.Arena(game, evs[0], evs[1], num_simulations=800, evaluation_reuse=reuse)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = arena.play(6, record=True)
        torch.cuda.synchronize()
        out.append((res, arena.transcript, time.perf_counter() - t0))
    (r1, t1, s1), (r0, t0_, s0) = out
    assert r1 == r0 and r1["games"] == 6
    for k in t1:
        assert np.array_equal(t1[k], t0_[k]), k
    print("arena 6 games x 800 sims: %.2f s with evaluation reuse, %.2f s without" % (s1, s0))


def test_alphazero_player_api():
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(4, 4)
    rp = pkg.RandomPlayer(game)
    b = game.getInitBoard()
    np.random.seed(0)
    a = rp.play(b, 1)
    assert 0 <= a < 16
    full = game.getInitBoard()
    full.board[...] = np.array([[1, -1, 1, -1], [-1, 1, -1, 1], [1, -1, 1, -1], [-1, 1, -1, 1]], np.int8)
    assert rp.play(full, 1) == -1


def test_cli_self_play_config1(tmp_path):
    """BASELINE config[0] plumbing: `train_alphazero.py --mode self-play --rows 6 --cols 6 --simulations 25
    --episodes 4 --workers 1` (train_alphazero.py:103-122): refuses to run without the model file, writes
    data/self_play_data_<ts>.npz with boards/policies/values of matching length."""
    import json
    import subprocess
    import sys
    import torch
    import yinyang_game_alphazero_amd as pkg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir, ddir = tmp_path / "models", tmp_path / "data"
    cmd = [sys.executable, os.path.join(root, "train_alphazero.py"), "--mode", "self-play", "--rows", "6", "--cols", "6",
           "--simulations", "25", "--episodes", "4", "--workers", "1", "--model-dir", str(mdir), "--data-dir", str(ddir)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Model file not found" in (r.stderr + r.stdout)
    torch.manual_seed(0)
    pkg.YinYangNeuralNetwork(pkg.YinYangGame(6, 6)).save_model(str(mdir / "best_model.pth.tar"))
    # The batched engine draws noise and moves from per-game counter streams (csrc/yy_selfplay.hip), not from numpy's global
    # stream, so the number of examples differs from one run of the reference (BASELINE.md measured 4 = one example per game
    # for ITS stream in literal mode; the reference-API SelfPlayWorker, which does use numpy's stream, is pinned move by move
    # by the G4 transcripts).  What the file-level test can hold exactly: the run is a pure function of (seed, model) -- two
    # invocations write identical arrays -- and every game contributes between 1 and 36 (+ passes) examples.
    for extra, lo, hi in (([], 40, 37 * 4), (["--board-semantics", "aliased", "--reference-quirks"], 4, 60)):
        files = []
        for rep in range(2):
            r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            info = json.loads(r.stdout.strip().splitlines()[-1])
            z = np.load(info["data_file"])
            files.append({k: z[k] for k in z.files})
            os.remove(info["data_file"])
        a, b = files
        n = a["boards"].shape[0]
        print("config 1 CLI,", extra or "default engine semantics", "->", n, "examples")
        assert a["boards"].shape == (n, 6, 6) and a["policies"].shape == (n, 36) and a["values"].shape == (n,)
        assert lo <= n <= hi and len(np.unique(a["game_id"])) == 4
        assert np.allclose(a["policies"].sum(1), 1.0, atol=1e-6)
        for k in a:
            assert np.array_equal(a[k], b[k]), k          # same seed, same model -> the same file, bit for bit


def test_cli_self_play_two_ranks_equals_one_rank(tmp_path):
    """`train_alphazero.py --mode self-play` under the distributed launcher with two ranks (sharing this box's GPU, collectives
    over gloo) against the single-process run: rank r plays games r, r+2, ..., ONE exchange collects the examples, rank 0
    writes the file behind a barrier -- and the file holds the SAME examples (a game's noise and moves are keyed by its global
    index, not by the rank or the slot it ran in): states, pi, z, game ids, plies, row for row after sorting."""
    import json
    import subprocess
    import sys
    import torch
    import yinyang_game_alphazero_amd as pkg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir = tmp_path / "models"
    os.makedirs(mdir)
    torch.manual_seed(0)
    pkg.YinYangNeuralNetwork(pkg.YinYangGame(6, 6), 128, 2).save_model(str(mdir / "best_model.pth.tar"))
    base = [os.path.join(root, "train_alphazero.py"), "--mode", "self-play", "--rows", "6", "--cols", "6", "--simulations", "20",
            "--episodes", "10", "--workers", "1", "--model-dir", str(mdir), "--blocks", "2", "--seed", "3"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {}
    for world in (1, 2):
        ddir = tmp_path / f"data{world}"
        cmd = ([sys.executable] if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", "29655"]) + base + ["--data-dir", str(ddir)] + (["--dist-backend", "gloo"] if world > 1 else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                        # rank 0 alone reports
        z = np.load(json.loads(lines[0])["data_file"])
        order = np.lexsort((z["ply"], z["game_id"]))
        out[world] = {k: z[k][order] for k in z.files}
    assert sorted(set(out[1]["game_id"].tolist())) == list(range(10))
    for k in out[1]:
        assert np.array_equal(out[1][k], out[2][k]), k


def test_bench_json_contract(tmp_path):
    """bench.py prints ONE JSON line with the driver's contract keys, the roofline and cpu_baseline objects."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--games", "256", "--sims", "40", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["expansions_per_s"] > 0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["unit"] in ("GB/s", "TFLOP/s")
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["aliased_expansions_per_s"] > 0 and "one per host core" in cb["sample"]
    # the headline leg evaluates at the reference's precision; the reduced-precision figure is a labelled extra key
    assert d["dtype"] == "f16x3" and "float32" in d["dtype_note"] and d["config"]["nn_dtype"] == "f16x3"
    assert "k_tower_g" in rf["kernel"] and rf["bound"] == "mfma" and "traffic_source" in rf
    assert "40 sims" in d["metric"] and "8x8" in d["metric"]
    sec = d["secondary"]
    assert sec["nn"] == "bf16" and sec["value"] > 0 and "REDUCED" in sec["note"]
    tree = d["roofline_tree_kernel"]
    assert tree["bound"] == "hbm" and "traffic_source" in tree and 0 < tree["eval_fraction"] <= 1
    # the launch as the timed steps issue it (compacted live rows), beside the dense one
    assert 0 < rf["live_rows"] <= 256 and rf["live_launch_ms"] > 0 and abs(rf["live_frac"] - rf["live_achieved"] / rf["peak"]) < 1e-9
    # the headline leg gives the evaluator every row the reference evaluates; the engine's evaluation reuse is a labelled extra leg
    assert d["config"]["evaluation_reuse"] is False
    ru = d["with_evaluation_reuse"]
    assert ru["nn"] == "f16x3" and ru["value"] > 0 and ru["evaluator_rows_per_s"] > 0 and ru["eval_fraction"] <= tree["eval_fraction"] + 0.02
    assert ru["node_expansions_per_s"] >= ru["evaluator_rows_per_s"]
    assert ru["opening_book"]["positions"] == ru["opening_book"]["stored"] == 769880 and ru["opening_book"]["build_s"] > 0


@pytest.mark.parametrize("nproc,backend,scaling", [(2, "gloo", "weak"), (2, "gloo", "strong"), (1, "nccl", "weak")],
                         ids=["2 ranks on one GPU, gloo", "2 ranks, gloo, strong scaling", "1 rank, RCCL"])
def test_bench_under_the_distributed_launcher(nproc, backend, scaling):
    """The command the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    with the REAL engine: two ranks sharing this box's one GPU (collectives over gloo: RCCL refuses two ranks on one device), and
    one rank with the RCCL communicator (the example exchange is then a real RCCL all_gather_into_tensor on device buffers).
    Rank r plays games r, r+N, ...; the line is the whole job: n_gpus = N, positions of all ranks, one gathered example set."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + nproc), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--games", "320", "--sims", "24", "--dist-backend", backend, "--no-cpu-baseline", "--scaling", scaling]
    per_rank = 320 if scaling == "weak" else 320 // nproc      # strong: --games in total, split over the ranks
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == nproc and d["steps"] == 2 and d["scaling"] == scaling and d["dtype"] == "f16x3"
    if nproc > 1:                                              # the CPU leg and the extra legs are N = 1 only
        assert d["cpu_baseline"] is None and "secondary" not in d and "with_evaluation_reuse" not in d
    assert f"episode-sharded x{nproc}" == d["config"]["parallelism"]
    # staggered starts: some slots are between games in a given move, most are searching
    assert 0.5 * 2 * per_rank * nproc <= d["value"] * d["ms_per_step"] * 2 / 1e3 <= 2 * per_rank * nproc
    assert d["examples_gathered"] > 0 and d["gather_s"] >= 0


# ---- arena against the reference's own AlphaZero.evaluate (alphazero.py:136-226), run unmodified with two hash evaluators
# by tests/golden/make_golden.py (gen_arena): every move of every game, the results the loop saw, its win/loss/draw accounting
ARENA = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arena_*.npz")))


@pytest.mark.parametrize("path", ARENA, ids=[os.path.basename(p) for p in ARENA])
@pytest.mark.parametrize("tag", ["literal", "copied"])
def test_arena_equals_reference_evaluate(path, tag):
    """literal: Arena(literal=True) = the reference's match loop as is (aliased boards mutated by every search, no pass
    handling, reference scoring).  copied: the default arena (copied boards, passes) under reference_scoring=True against the
    reference loop driven through the copied-board game adapter.  Same moves (np.argmax of pi: lowest index among the most
    visited), same players, same game lengths, same getGameEnded values, same current/best/draw counts."""
    import yinyang_game_alphazero_amd as pkg
    from hash_eval import hash_eval_torch
    z = np.load(path)
    R, C = (int(x) for x in os.path.basename(path)[6:-4].split("x"))
    game = pkg.YinYangGame(R, C)
    ev_cur = lambda planes: hash_eval_torch(planes, 10, 11)      # "current" = A
    ev_best = lambda planes: hash_eval_torch(planes, 6, 4)       # "best" = B
    sims = int(z[tag + "_sims"])
    n = z[tag + "_results"].shape[0]
    arena = pkg.Arena(game, ev_cur, ev_best, num_simulations=sims, literal=(tag == "literal"), reference_scoring=True)
    res = arena.play(n, record=True)
    t = arena.transcript
    assert np.array_equal(t["n_moves"], z[tag + "_n_moves"]), (t["n_moves"], z[tag + "_n_moves"])
    for i in range(n):
        k = int(t["n_moves"][i])
        assert np.array_equal(t["actions"][i, :k], z[tag + "_actions"][i, :k]), (tag, i)
        assert np.array_equal(t["players"][i, :k], z[tag + "_players"][i, :k]), (tag, i)
    assert np.array_equal(t["results"], z[tag + "_results"])
    assert (res["a_wins"], res["b_wins"], res["draws"]) == (int(z[tag + "_current_wins"]), int(z[tag + "_best_wins"]),
                                                            int(z[tag + "_draws"]))


def test_select_action_equals_reference():
    """MCTS.select_action (mcts.py:427-479) against the reference's recorded actions: temperature 0 (np.argmax), 1, 0.5 and
    the instance default, with and without valid_moves (the legal mask, a mask that removes every searched move -> arg-max
    fallback, a wrong-length mask -> ignored), the global numpy stream seeded as the generator seeded it."""
    import yinyang_game_alphazero_amd as pkg
    from hash_eval import hash_eval_np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "select_action.npz"))

    class HashNet:
        def predict(self, board):
            return hash_eval_np(board.get_board(), 6, 4)

    checked = 0
    for (R, C) in ((3, 3), (6, 6), (8, 8)):
        tag = f"{R}x{C}"
        game = pkg.YinYangGame(R, C)
        A = R * C
        for i in range(z["action_" + tag].shape[0]):
            m = pkg.MCTS(game, HashNet(), num_simulations=int(z["sims_" + tag][i]), board_semantics="copied", verbose=0)
            board = game.getInitBoard()
            board.board[...] = z["board_" + tag][i]
            player = int(z["player_" + tag][i])
            temp = float(z["temp_" + tag][i])
            vm = z["mask_" + tag][i][: int(z["mask_len_" + tag][i])] if z["has_mask_" + tag][i] else None
            np.random.seed(int(z["seed_" + tag][i]))
            pi, _ = m.search(board, player)
            assert np.array_equal(pi, z["pi_" + tag][i]), (tag, i)
            np.random.seed(int(z["seed_" + tag][i]))
            a = m.select_action(board, player, temperature=None if temp < 0 else temp, valid_moves=vm)
            assert int(a) == int(z["action_" + tag][i]), (tag, i, temp, vm is not None)
            m.close()
            checked += 1
    assert checked >= 60


@pytest.mark.parametrize("graph_step", [False, True], ids=["eager", "hipgraph_step"])
def test_trainer_steps_equal_reference_trainer_gpu(graph_step):
    """The same fixture on the GPU (stock PyTorch-ROCm training step, eager and replayed from a hipGraph): per-step losses
    within 1e-5 relative of the reference's CPU run, parameters and BatchNorm buffers within 1e-4 abs after the 12 Adam
    steps (float32 on another device: summation order differs).  One class of parameters is held to a different bound: the
    bias of a convolution that feeds a BatchNorm has an exactly-zero true gradient (the normalisation removes any constant),
    so what it receives is rounding noise, which Adam's g / sqrt(v) normalisation turns into steps of +-lr whatever its size;
    those biases may differ by up to steps * lr and do not influence the network (checked through the losses)."""
    from test_training_host import _trainer_fixture_run
    z, log, final, m = _trainer_fixture_run("cuda", graph_step=graph_step)
    assert log.shape == (12, 2)
    dl = max(np.abs(log[:, 0] / z["policy_loss"] - 1).max(), np.abs(log[:, 1] - z["value_loss"]).max())
    # ... and so may the running mean of the BatchNorm behind such a bias (it tracks mean(conv output) = ... + bias)
    dead = lambda k: k.endswith(("conv1.bias", "conv2.bias", "policy_conv.bias", "value_conv.bias", "running_mean"))
    dev = {k: float(np.abs(v.astype(np.float64) - z["final/" + k].astype(np.float64)).max()) for k, v in final.items()
           if "num_batches" not in k}
    worst = max(v for k, v in dev.items() if not dead(k))
    worst_dead = max(v for k, v in dev.items() if dead(k))
    print("trainer vs reference (GPU, graph=%s): max loss deviation %.3e, max |dparam| %.3e (zero-gradient conv biases %.3e)"
          % (graph_step, dl, worst, worst_dead))
    top = sorted(dev.items(), key=lambda kv: -kv[1])[:6]
    assert dl < 1e-5 and worst < 1e-4 and worst_dead <= 12 * 0.001 * 2.01, top
    assert np.allclose(m["total_loss"], z["epoch_total_loss"], rtol=1e-5)


def test_augmentation_and_example_format_on_device_tensors(tmp_path):
    """f1 on the GPU: the batched 8-fold augmentation on device tensors equals the reference's DataProcessor.augment_sample
    fixture (exact), and examples written / read back through the .npz formats keep their bits (tensor-native and the
    reference's object layout)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    from yinyang_game_alphazero_amd import training as T
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))
    for R in (6, 8):
        boards = torch.from_numpy(z[f"boards_{R}"]).cuda()
        planes = pkg.engine.encode_planes(boards)
        pi = torch.from_numpy(z[f"pi_{R}"]).cuda()
        ap, apol = T.augment_batch(planes, pi)
        n = boards.shape[0]
        assert ap.is_cuda and ap.shape[0] == 8 * n
        # the reference lists the 8 variants per sample; augment_batch stacks variant-major
        got_p = ap.reshape(8, n, *ap.shape[1:]).permute(1, 0, 2, 3, 4).cpu().numpy()
        got_pi = apol.reshape(8, n, -1).permute(1, 0, 2).cpu().numpy()
        assert np.array_equal(got_p, z[f"aug_planes_{R}"]) and np.array_equal(got_pi, z[f"aug_pi_{R}"])
        # format round trip from device tensors
        ex = dict(states=boards, policies=pi, values=torch.linspace(-1, 1, n, device="cuda"))
        path = str(tmp_path / f"ex_{R}.npz")
        np.savez(path, states=ex["states"].cpu().numpy(), policies=ex["policies"].cpu().numpy().astype(np.float64),
                 values=ex["values"].cpu().numpy().astype(np.float64))
        back = T.load_examples(path)
        assert np.array_equal(np.asarray(back["states"]), boards.cpu().numpy())
        assert np.allclose(np.asarray(back["policies"]), pi.cpu().numpy()) and np.allclose(np.asarray(back["values"]), ex["values"].cpu().numpy())
        ref_path = str(tmp_path / f"ref_{R}.npz")
        T.save_examples_reference_format(ref_path, boards.cpu().numpy(), pi.cpu().numpy(), ex["values"].cpu().numpy())
        back2 = T.load_examples(ref_path, allow_reference_objects=True)
        assert np.array_equal(np.asarray(back2["states"]), boards.cpu().numpy())
        q = T.TrainingDataQueue(max_size=100)
        q.push_file(ref_path, allow_reference_objects=True)
        assert len(q) == n


def test_cli_train_mode_two_ranks(tmp_path):
    """`train_alphazero.py --mode train` under the distributed launcher with two ranks (sharing this box's GPU, collectives over
    gloo): one whole AlphaZero.run iteration -- self-play sharded over the ranks + ONE example exchange, the published file read by
    both ranks, DistributedDataParallel training, the sharded arena match with its win counts all-reduced, promotion decided
    once, checkpoints written by rank 0 behind barriers (ai/alphazero.py:249-270, ai/self_play.py:288-335).  Both ranks finish,
    report ONE line, hold bit-identical weights after training, and leave exactly one set of files."""
    import glob
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir, ddir = tmp_path / "models", tmp_path / "data"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "train_alphazero.py"), "--mode", "train", "--rows", "6", "--cols", "6",
           "--iterations", "1", "--episodes", "12", "--simulations", "16", "--epochs", "2", "--batch-size", "16", "--channels", "32",
           "--blocks", "1", "--arena-games", "6", "--concurrent-games", "8", "--model-dir", str(mdir), "--data-dir", str(ddir),
           "--dist-backend", "gloo"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-5000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                            # rank 0 alone reports
    it = json.loads(lines[0])["iterations"]
    assert len(it) == 1
    h = it[0]
    assert os.path.exists(h["data_file"]) and h["examples"] > 0 and len(h["losses"]) == 2
    assert len(h["param_checksums"]) == 2 and h["param_checksums"][0] == h["param_checksums"][1]      # DDP: identical weights on both ranks
    assert h["arena"]["games"] == 6 and h["arena"]["a_wins"] + h["arena"]["b_wins"] + h["arena"]["draws"] == 6
    assert abs(h["win_ratio"] - h["arena"]["a_wins"] / 6.0) < 1e-12 and h["promoted"] == (h["win_ratio"] >= 0.6)
    z = np.load(h["data_file"])
    assert sorted(set(z["game_id"].tolist())) == list(range(12))         # both ranks' games, gathered once
    assert len(glob.glob(str(ddir / "self_play_data_*.npz"))) == 1 and not glob.glob(str(ddir / "*.tmp*"))
    names = sorted(os.path.basename(f) for f in glob.glob(str(mdir / "*.pth.tar")))
    assert names == ["best_model.pth.tar", "checkpoint_1.pth.tar", "current_model.pth.tar"]

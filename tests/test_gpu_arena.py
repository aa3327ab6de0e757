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
           "--simulations", "25", "--episodes", "4", "--workers", "1", "--model-dir", str(mdir), "--data-dir", str(ddir),
           "--nn", "fp32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Model file not found" in (r.stderr + r.stdout)
    torch.manual_seed(0)
    pkg.YinYangNeuralNetwork(pkg.YinYangGame(6, 6)).save_model(str(mdir / "best_model.pth.tar"))
    for extra, lo, hi in (([], 40, 37 * 4), (["--board-semantics", "aliased", "--reference-quirks"], 4, 60)):
        r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        info = json.loads(r.stdout.strip().splitlines()[-1])
        z = np.load(info["data_file"])
        n = z["boards"].shape[0]
        assert z["boards"].shape == (n, 6, 6) and z["policies"].shape == (n, 36) and z["values"].shape == (n,)
        assert lo <= n <= hi and len(np.unique(z["game_id"])) == 4
        assert np.allclose(z["policies"].sum(1), 1.0, atol=1e-6)
        os.remove(info["data_file"])


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

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
    # random vs random needs no search at all
    res3 = pkg.Arena(game, "random", "random", num_simulations=1, seed=3).play(32)
    assert res3["a_wins"] + res3["b_wins"] + res3["draws"] == 32 and res3["a_wins"] > 0 and res3["b_wins"] > 0


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

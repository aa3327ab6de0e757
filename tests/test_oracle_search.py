"""The CPU oracle's MCTS (oracle/yy_oracle.c) against MCTS.search of the imported reference (G3
fixtures made by tests/golden/make_golden.py): identical visit counts, float32 value sums, priors,
evaluator call order (leaf boards) and, in aliased mode, the mutated caller board."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEARCH = sorted(glob.glob(os.path.join(GOLDEN, "search_[0-9]*.npz")))


@pytest.mark.parametrize("path", SEARCH, ids=[os.path.basename(p) for p in SEARCH])
def test_search_hash_evaluator(path):
    z = np.load(path)
    n = z["counts"].shape[0]
    assert n > 0
    for i in range(n):
        noise = z["noise"][i] if z["has_noise"][i] else None
        nl = int(z["n_leaves"][i])
        r = O.search_hash(z["root_board"][i], int(z["root_player"][i]), int(z["sims"][i]),
                          int(z["copied"][i]), int(z["pbits"][i]), int(z["vbits"][i]), noise=noise,
                          leaf_cap=nl)
        tag = f"{os.path.basename(path)} case {i}"
        assert np.array_equal(r.counts, z["counts"][i]), tag
        assert np.array_equal(r.child_p, z["child_p"][i]), tag
        assert np.array_equal(r.child_w, z["child_w"][i]), tag
        assert r.root_visits == int(z["root_visits"][i]) == int(z["sims"][i]), tag
        assert r.root_w == float(z["root_w"][i]), tag
        assert r.n_evals == int(z["n_evals"][i]), tag
        assert np.array_equal(r.pi, z["pi"][i]), tag
        assert np.array_equal(r.final_board, z["final_board"][i]), tag
        if nl:
            assert np.array_equal(r.leaves, z["leaves"][i, :nl]), tag


def test_search_recorded_network():
    """Real seeded 128x10 net: the reference's evaluator outputs were recorded and are replayed."""
    z = np.load(os.path.join(GOLDEN, "search_net_8x8.npz"))
    for i in range(z["counts"].shape[0]):
        n = int(z["n_rec"][i])
        noise = z["noise"][i] if z["has_noise"][i] else None
        r = O.search_replay(z["root_board"][i], 1, int(z["sims"][i]), int(z["copied"][i]),
                            z["rec_policy"][i, :n], z["rec_value"][i, :n], noise=noise, leaf_cap=n)
        assert r.n_evals == n - 1
        assert np.array_equal(r.counts, z["counts"][i])
        assert np.array_equal(r.child_w, z["child_w"][i])
        assert np.array_equal(r.child_p, z["child_p"][i])
        assert r.root_w == float(z["root_w"][i])
        assert np.array_equal(r.leaves, z["rec_boards"][i, 1:n])
        assert np.array_equal(r.final_board, z["final_board"][i])
        assert np.max(np.abs(r.pi - z["pi"][i])) == 0.0

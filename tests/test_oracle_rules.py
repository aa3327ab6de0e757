"""The CPU oracle (oracle/yy_oracle.c) against the golden vectors captured from the imported
reference (tests/golden/make_golden.py): G1 rules and G2 planes.  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RULES = sorted(glob.glob(os.path.join(GOLDEN, "rules_*.npz")))
PLANES = sorted(glob.glob(os.path.join(GOLDEN, "planes_*.npz")))


def test_fixtures_present():
    assert len(RULES) >= 3 and len(PLANES) >= 3


@pytest.mark.parametrize("path", RULES, ids=[os.path.basename(p) for p in RULES])
def test_rules_against_reference(path):
    z = np.load(path)
    b = z["boards"]
    assert np.array_equal(O.valid_mask(b, 1), z["mask_p1"])
    assert np.array_equal(O.valid_mask(b, -1), z["mask_m1"])
    assert np.array_equal(O.game_ended(b, 1), z["ended_p1"])
    assert np.array_equal(O.game_ended(b, -1), z["ended_m1"])
    assert np.array_equal((b == 1).sum((1, 2)), z["counts"][:, 0])
    assert np.array_equal((b == -1).sum((1, 2)), z["counts"][:, 1])
    nb, npl, placed = O.next_state(b, z["players"], z["step_action"])
    assert np.array_equal(nb, z["step_board"])
    assert np.array_equal(npl, z["step_player"])
    assert np.array_equal(placed, z["step_placed"])


@pytest.mark.parametrize("path", PLANES, ids=[os.path.basename(p) for p in PLANES])
def test_planes_against_reference(path):
    z = np.load(path)
    got = O.encode_planes(z["boards"])
    assert got.dtype == np.float32
    assert np.array_equal(got, z["planes"])


def test_hash_eval_matches_generator():
    """The C hash evaluator equals the numpy one used when the goldens were made."""
    from hash_eval import hash_eval_np
    rng = np.random.default_rng(0)
    for (R, C) in ((3, 3), (6, 6), (8, 8), (12, 12), (5, 7)):
        for pb, vb in ((10, 11), (2, 2), (6, 4)):
            b = rng.integers(-1, 2, size=(R, C)).astype(np.int8)
            p0, v0 = O.hash_eval(b, pb, vb)
            p1, v1 = hash_eval_np(b, pb, vb)
            assert np.array_equal(p0, p1) and v0 == v1


ROWCOL = sorted(glob.glob(os.path.join(GOLDEN, "rowcol_*.npz")))


@pytest.mark.parametrize("path", ROWCOL, ids=[os.path.basename(p) for p in ROWCOL])
def test_rowcol_rule_against_reference_javascript(path):
    """The full-row/column rule exists only in the reference's browser game (src/gui/static/js/yin_yang_game.js:187-232,
    338-384); tests/golden/rowcol_from_js.js ran that class under node and recorded its isValidMove masks for both colours
    (positions from the Python reference's own random play + boards built to stress the rule).  The oracle's flags=1 path must
    reproduce them bit for bit -- this pins YY_FLAG_ROWCOL."""
    z = np.load(path)
    b = z["boards"]
    assert np.array_equal(O.valid_mask(b, 1, flags=1), z["mask_p1"])
    assert np.array_equal(O.valid_mask(b, -1, flags=1), z["mask_m1"])
    if b.shape[1] > 1 and b.shape[2] > 1:
        # the rule really bites in the fixture: some masks differ from the rule-less ones
        assert (O.valid_mask(b, 1, flags=0) != z["mask_p1"]).any()


def test_rowcol_fixtures_present():
    assert len(ROWCOL) >= 8

"""G4: full play_game transcripts of the imported reference (literal = aliased boards, and the
copied-board adapter; both with the reference's Q4/Q5 quirks) against the oracle-driven restatement
of the episode loop.  Identical seeds => identical boards searched, pi, actions, labels."""
import glob
import os

import numpy as np
import pytest

from oracle_episode import play_game

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPIS = sorted(glob.glob(os.path.join(GOLDEN, "episodes_*.npz")))


def check_transcript(z, i, t):
    n = int(z["n"][i])
    assert t["n"] == n
    assert np.array_equal(np.stack(t["search_boards"]), z["search_boards"][i, :n])
    assert np.array_equal(np.stack(t["pis"]), z["pis"][i, :n])
    assert t["actions"] == z["actions"][i, :n].tolist()
    assert t["players"] == z["players"][i, :n].tolist()
    assert np.array_equal(np.asarray(t["z"], np.float64), z["z"][i, :n])
    assert np.array_equal(np.stack(t["example_boards"]), z["example_boards"][i, :n])


@pytest.mark.parametrize("path", EPIS, ids=[os.path.basename(p) for p in EPIS])
def test_episode_transcripts(path):
    z = np.load(path)
    R, C = z["search_boards"].shape[2:]
    for i in range(z["n"].shape[0]):
        t = play_game(R, C, int(z["seed"][i]), int(z["sims"][i]), int(z["copied"][i]), int(z["pbits"][i]),
                      int(z["vbits"][i]), quirks=True)
        check_transcript(z, i, t)

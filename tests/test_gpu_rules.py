"""GPU parity, rules path: the HIP kernels (through the C ABI) against the golden vectors captured
from the imported reference AND against the CPU oracle on fresh seeded boards.  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RULES = sorted(glob.glob(os.path.join(GOLDEN, "rules_*.npz")))
PLANES = sorted(glob.glob(os.path.join(GOLDEN, "planes_*.npz")))


@pytest.fixture(scope="module")
def E():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import yinyang_game_alphazero_amd as pkg
    return pkg.engine


def _t(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("path", RULES, ids=[os.path.basename(p) for p in RULES])
def test_rules_golden(E, path):
    import torch
    z = np.load(path)
    b = _t(z["boards"])
    G = b.shape[0]
    p1 = torch.ones(G, dtype=torch.int8, device="cuda")
    m1 = -p1
    assert np.array_equal(E.valid_mask(b, p1).cpu().numpy(), z["mask_p1"])
    assert np.array_equal(E.valid_mask(b, m1).cpu().numpy(), z["mask_m1"])
    e1, cnt = E.game_ended(b, p1, with_counts=True)
    assert np.array_equal(e1.cpu().numpy(), z["ended_p1"])
    assert np.array_equal(E.game_ended(b, m1).cpu().numpy(), z["ended_m1"])
    assert np.array_equal(cnt.cpu().numpy(), z["counts"])
    nb, pl = b.clone(), _t(z["players"])
    placed = E.step_(nb, pl, _t(z["step_action"]))
    assert np.array_equal(nb.cpu().numpy(), z["step_board"])
    assert np.array_equal(pl.cpu().numpy(), z["step_player"])
    assert np.array_equal(placed.cpu().numpy(), z["step_placed"])
    # packed bitboard form of the same rules
    R, C = z["boards"].shape[1:]
    bl, wh = E.pack_boards(b)
    assert np.array_equal(E.unpack_boards(bl, wh, R, C).cpu().numpy(), z["boards"])
    k1, k2, res = E.mask_terminal_bb(bl, wh, R, C)
    A = R * C
    bits = lambda m: ((m.cpu().numpy().astype(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)) \
        .transpose(1, 0, 2).reshape(G, -1)[:, :A].astype(np.uint8)
    assert np.array_equal(bits(k1), z["mask_p1"])
    assert np.array_equal(bits(k2), z["mask_m1"])
    want = np.where(z["ended_p1"] == 0.0001, 2, z["ended_p1"]).astype(np.int8)
    assert np.array_equal(res.cpu().numpy(), want)


@pytest.mark.parametrize("path", PLANES, ids=[os.path.basename(p) for p in PLANES])
def test_planes_golden(E, path):
    z = np.load(path)
    got = E.encode_planes(_t(z["boards"])).cpu().numpy()
    assert np.array_equal(got, z["planes"])


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (7, 1), (2, 2), (3, 3), (6, 6), (8, 8), (11, 5), (12, 12),
                                   (16, 12), (12, 16), (13, 13), (16, 4)])
@pytest.mark.parametrize("rowcol", [False, True])
def test_rules_vs_oracle_random(E, shape, rowcol):
    """fresh seeded boards (not in the fixtures), incl. degenerate and maximum sizes and the optional
    browser-only row/column rule (pinned by the oracle only: the Python reference has no such rule)."""
    import torch
    R, C = shape
    rng = np.random.default_rng(R * 100 + C)
    G = 777
    dens = rng.uniform(0.05, 0.95, size=(G, 1, 1))
    u = rng.random((G, R, C))
    col = rng.random((G, R, C)) < 0.5
    boards = np.where(u < dens, 0, np.where(col, 1, -1)).astype(np.int8)
    # a share of real-play-like boards: grow legal positions with the oracle
    cur = np.zeros((G // 3, R, C), np.int8)
    pl = np.ones(G // 3, np.int8)
    for _ in range(rng.integers(1, R * C + 1)):
        m = O.valid_mask(cur, pl, flags=int(rowcol))
        act = np.array([rng.choice(np.flatnonzero(r)) if r.any() else 0 for r in m], np.int32)
        cur, pl, _ = O.next_state(cur, pl, act, flags=int(rowcol))
    boards[: G // 3] = cur
    players = rng.choice(np.array([1, -1], np.int8), size=G)
    f = int(rowcol)
    b, p = _t(boards), _t(players)
    assert np.array_equal(E.valid_mask(b, p, rowcol).cpu().numpy(), O.valid_mask(boards, players, f))
    assert np.array_equal(E.game_ended(b, p, rowcol).cpu().numpy(), O.game_ended(boards, players, f))
    acts = rng.integers(0, R * C, size=G).astype(np.int32)
    nb, npl = b.clone(), p.clone()
    placed = E.step_(nb, npl, _t(acts), rowcol)
    ob, opl, oplaced = O.next_state(boards, players, acts, f)
    assert np.array_equal(nb.cpu().numpy(), ob) and np.array_equal(npl.cpu().numpy(), opl)
    assert np.array_equal(placed.cpu().numpy(), oplaced)
    assert np.array_equal(E.encode_planes(b).cpu().numpy(), O.encode_planes(boards))


def test_empty_and_errors(E):
    import torch
    import yinyang_game_alphazero_amd as pkg
    z = torch.zeros((0, 8, 8), dtype=torch.int8, device="cuda")
    assert E.valid_mask(z, torch.zeros(0, dtype=torch.int8, device="cuda")).shape == (0, 64)
    with pytest.raises(pkg.YYError):
        E.valid_mask(torch.zeros((2, 17, 8), dtype=torch.int8, device="cuda"), torch.ones(2, dtype=torch.int8, device="cuda"))
    with pytest.raises(pkg.YYError):
        E.valid_mask(torch.zeros((2, 8, 8), dtype=torch.int8), torch.ones(2, dtype=torch.int8))   # CPU tensors: no fallback


ROWCOL = sorted(glob.glob(os.path.join(GOLDEN, "rowcol_*.npz")))


@pytest.mark.parametrize("path", ROWCOL, ids=[os.path.basename(p) for p in ROWCOL])
def test_rowcol_rule_golden_from_reference_javascript(E, path):
    """YY_FLAG_ROWCOL in the HIP rules kernels against the masks the reference's own JavaScript game produced under node
    (yin_yang_game.js:187-232, 338-384; generator tests/golden/rowcol_from_js.js).  Bit-exact, both colours; the step
    kernel places exactly on the legal cells."""
    import torch
    z = np.load(path)
    b = _t(z["boards"])
    G, R, C = z["boards"].shape
    p1 = torch.ones(G, dtype=torch.int8, device="cuda")
    assert np.array_equal(E.valid_mask(b, p1, True).cpu().numpy(), z["mask_p1"])
    assert np.array_equal(E.valid_mask(b, -p1, True).cpu().numpy(), z["mask_m1"])
    bl, wh = E.pack_boards(b)
    k1, k2, _ = E.mask_terminal_bb(bl, wh, R, C, rowcol=True)
    A = R * C
    bits = lambda m: ((m.cpu().numpy().astype(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)) \
        .transpose(1, 0, 2).reshape(G, -1)[:, :A].astype(np.uint8)
    assert np.array_equal(bits(k1), z["mask_p1"]) and np.array_equal(bits(k2), z["mask_m1"])
    rng = np.random.default_rng(R * 31 + C)
    acts = rng.integers(0, A, size=G).astype(np.int32)
    nb, pl = b.clone(), p1.clone()
    placed = E.step_(nb, pl, _t(acts), True).cpu().numpy()
    assert np.array_equal(placed, z["mask_p1"][np.arange(G), acts])

"""GPU parity, search path: BatchedMCTS (HIP kernels through the C ABI) against MCTS.search of the
imported reference (G3 fixtures) with the exact dyadic hash evaluator computed on the host from the
planes the kernel wrote.  Identical visit counts / float32 value sums / priors / evaluator call
order / mutated boards; pi identical (tolerance 0, well inside the 1e-5 of the north star)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from hash_eval import hash_eval_batch, planes_to_boards

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEARCH = sorted(glob.glob(os.path.join(GOLDEN, "search_[0-9]*.npz")))


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available()
    import yinyang_game_alphazero_amd as pkg
    return pkg


class HostHashEvaluator:
    """policy/value from the planes the select kernel wrote; logs the evaluated boards per game."""

    def __init__(self, pbits, vbits, needs_eval=None):
        self.pbits, self.vbits = np.asarray(pbits), np.asarray(vbits)
        self.calls = 0
        self.logs = [[] for _ in self.pbits]
        self.needs_eval = needs_eval

    def __call__(self, planes):
        import torch
        b = planes_to_boards(planes.cpu().numpy())
        G, A = b.shape[0], b.shape[1] * b.shape[2]
        pol = np.zeros((G, A), np.float32)
        val = np.zeros(G, np.float32)
        for pb, vb in set(zip(self.pbits.tolist(), self.vbits.tolist())):
            idx = np.flatnonzero((self.pbits == pb) & (self.vbits == vb))
            pol[idx], val[idx] = hash_eval_batch(b[idx], pb, vb)
        if self.calls > 0 and self.needs_eval is not None:
            ne = self.needs_eval.cpu().numpy()
            for g in np.flatnonzero(ne):
                self.logs[g].append(b[g].copy())
        self.calls += 1
        return torch.from_numpy(pol).cuda(), torch.from_numpy(val).cuda()


def _run_group(pkg, z, idx, fused, reuse=False):
    import torch
    R, C = z["root_board"].shape[1:]
    sims, copied = int(z["sims"][idx[0]]), int(z["copied"][idx[0]])
    G = len(idx)
    kw = dict(reuse_pass_value=bool(copied), reuse_transpositions=True, keep_evaluations=True) if reuse else {}
    m = pkg.engine.BatchedMCTS(G, R, C, sims, cpuct=1.0, aliased=not copied, **kw)
    ev = HostHashEvaluator(z["pbits"][idx], z["vbits"][idx], m.needs_eval)
    boards = torch.from_numpy(z["root_board"][idx]).cuda()
    players = torch.from_numpy(z["root_player"][idx]).cuda()
    noise = torch.from_numpy(z["noise"][idx]).cuda()     # zeros where has_noise == 0 -> p = f32(.75p + 0) != p
    has = z["has_noise"][idx].astype(bool)
    # games without noise must keep the raw prior: run them in a separate context call
    assert has.all() or (~has).all()
    counts = m.search(boards, players, ev, sims, noise=noise if has.all() else None, fused=fused)
    c2, cw, cp = m.root_counts(with_children=True)
    visits, wsum = m.root_stats()
    pi = m.root_policy()
    fb = m.boards()
    m.status()
    torch.cuda.synchronize()
    for j, i in enumerate(idx):
        tag = f"case {i} sims {sims} copied {copied}"
        assert np.array_equal(counts[j].cpu().numpy(), z["counts"][i]), tag
        assert np.array_equal(c2[j].cpu().numpy(), z["counts"][i]), tag
        assert np.array_equal(cp[j].cpu().numpy(), z["child_p"][i]), tag
        assert np.array_equal(cw[j].cpu().numpy().astype(np.float64), z["child_w"][i]), tag
        assert int(visits[j]) == int(z["root_visits"][i]), tag
        assert float(wsum[j]) == float(z["root_w"][i]), tag
        assert np.array_equal(pi[j].cpu().numpy(), z["pi"][i]), tag
        assert np.array_equal(fb[j].cpu().numpy(), z["final_board"][i]), tag
        if reuse:       # the evaluator is asked for a subsequence of the reference's leaves: every position once
            assert len(ev.logs[j]) <= int(z["n_evals"][i]), tag
            assert len({a.tobytes() for a in ev.logs[j]}) == len(ev.logs[j]), tag
            continue
        assert len(ev.logs[j]) == int(z["n_evals"][i]), tag
        nl = int(z["n_leaves"][i])
        if nl:
            assert np.array_equal(np.stack(ev.logs[j]), z["leaves"][i, :nl]), tag
    m.close()


@pytest.mark.parametrize("path", SEARCH, ids=[os.path.basename(p) for p in SEARCH])
def test_search_golden_with_evaluation_reuse(pkg, path):
    """The reference's own results (G3 fixtures: visit counts, float32 value sums, priors, pi, mutated boards; copied and
    aliased boards, with and without noise, every geometry) with the evaluation reuse ON: pass values (copied boards) and the
    evaluation cache.  Nothing changes but the rows the evaluator is asked for -- no position twice."""
    z = np.load(path)
    groups = {}
    for i in range(z["counts"].shape[0]):
        groups.setdefault((int(z["sims"][i]), int(z["copied"][i]), int(z["has_noise"][i])), []).append(i)
    for key, idx in sorted(groups.items()):
        _run_group(pkg, z, np.asarray(idx), True, reuse=True)


@pytest.mark.parametrize("fused", [True, False], ids=["fused_step", "select+expand"])
@pytest.mark.parametrize("path", SEARCH, ids=[os.path.basename(p) for p in SEARCH])
def test_search_golden(pkg, path, fused):
    z = np.load(path)
    n = z["counts"].shape[0]
    groups = {}
    for i in range(n):
        groups.setdefault((int(z["sims"][i]), int(z["copied"][i]), int(z["has_noise"][i])), []).append(i)
    for key, idx in sorted(groups.items()):
        if not fused and key[0] > 200:
            continue
        _run_group(pkg, z, np.asarray(idx), fused)


def test_search_recorded_network(pkg):
    """Real 128x10 net recorded from the reference run: replay its outputs row by row."""
    import torch
    z = np.load(os.path.join(GOLDEN, "search_net_8x8.npz"))
    for i in range(z["counts"].shape[0]):
        n, sims, copied = int(z["n_rec"][i]), int(z["sims"][i]), int(z["copied"][i])
        m = pkg.engine.BatchedMCTS(1, 8, 8, sims, aliased=not copied)
        state = {"k": 0}
        recp, recv = z["rec_policy"][i], z["rec_value"][i]
        seen = []

        def ev(planes):
            k = state["k"]
            if k > 0 and int(m.needs_eval[0]) == 0:
                return torch.zeros((1, 64), device="cuda"), torch.zeros(1, device="cuda")
            seen.append(planes_to_boards(planes.cpu().numpy())[0])
            state["k"] = k + 1
            return torch.from_numpy(recp[k:k + 1]).cuda(), torch.from_numpy(recv[k:k + 1]).cuda()

        noise = torch.from_numpy(z["noise"][i:i + 1]).cuda() if z["has_noise"][i] else None
        counts = m.search(torch.from_numpy(z["root_board"][i:i + 1]).cuda(), torch.ones(1, dtype=torch.int8, device="cuda"),
                          ev, sims, noise=noise)
        assert state["k"] == n
        assert np.array_equal(np.stack(seen), z["rec_boards"][i, :n])
        assert np.array_equal(counts[0].cpu().numpy(), z["counts"][i])
        assert np.array_equal(m.root_policy()[0].cpu().numpy(), z["pi"][i])
        assert np.array_equal(m.boards()[0].cpu().numpy(), z["final_board"][i])
        m.close()


@pytest.mark.parametrize("shape,sims,copied", [((8, 8), 800, 1), ((8, 8), 800, 0), ((12, 12), 400, 1), ((6, 6), 200, 1),
                                               ((13, 13), 100, 1), ((16, 12), 60, 1), ((3, 3), 64, 1)])
def test_search_vs_oracle_batch(pkg, shape, sims, copied):
    """A batch of fresh seeded roots (random legal-play positions, both root players, coarse and fine
    evaluators, with noise) against the CPU oracle."""
    import torch
    R, C = shape
    A = R * C
    G = 48
    rng = np.random.default_rng(sims + R)
    boards = np.zeros((G, R, C), np.int8)
    pl = np.ones(G, np.int8)
    for ply in range(A):
        m = O.valid_mask(boards, pl)
        go = (rng.random(G) < 0.93) & m.any(1)
        act = np.array([rng.choice(np.flatnonzero(r)) if r.any() else 0 for r in m], np.int32)
        nb, npl, _ = O.next_state(boards, pl, act)
        boards[go], pl[go] = nb[go], npl[go]
    players = rng.choice(np.array([1, -1], np.int8), size=G)
    pb = rng.choice([10, 2, 6], size=G)
    vb = np.where(pb == 10, 11, np.where(pb == 2, 2, 4))
    noise = np.zeros((G, A))
    valid = O.valid_mask(boards, players)
    for g in range(G):
        k = int(valid[g].sum())
        if k:
            noise[g, valid[g] == 1] = rng.dirichlet([0.3] * k)
    m = pkg.engine.BatchedMCTS(G, R, C, sims, aliased=not copied)
    ev = HostHashEvaluator(pb, vb, m.needs_eval)
    counts = m.search(torch.from_numpy(boards).cuda(), torch.from_numpy(players).cuda(), ev, sims,
                      noise=torch.from_numpy(noise).cuda())
    c2, cw, cp = m.root_counts(with_children=True)
    fb = m.boards().cpu().numpy()
    visits, wsum = m.root_stats()
    m.status()
    for g in range(G):
        r = O.search_hash(boards[g], int(players[g]), sims, copied, int(pb[g]), int(vb[g]), noise=noise[g])
        assert np.array_equal(counts[g].cpu().numpy(), r.counts), g
        assert np.array_equal(cw[g].cpu().numpy().astype(np.float64), r.child_w), g
        assert np.array_equal(cp[g].cpu().numpy(), r.child_p), g
        assert np.array_equal(fb[g], r.final_board), g
        assert len(ev.logs[g]) == r.n_evals, g
        assert int(visits[g]) == sims and float(wsum[g]) == r.root_w, g
    m.close()


def test_inactive_games_and_state_errors(pkg):
    import torch
    m = pkg.engine.BatchedMCTS(4, 6, 6, 16)
    with pytest.raises(pkg.YYError):
        m.select()   # nothing begun... select before expand_root is a state error only after begin
        m.expand_backup(torch.zeros((4, 36), device="cuda"), torch.zeros(4, device="cuda"))
        m.expand_backup(torch.zeros((4, 36), device="cuda"), torch.zeros(4, device="cuda"))
    m.close()
    m = pkg.engine.BatchedMCTS(4, 6, 6, 16)
    active = torch.tensor([1, 0, 1, 0], dtype=torch.uint8, device="cuda")
    ev = HostHashEvaluator([10] * 4, [11] * 4, m.needs_eval)
    counts = m.search(torch.zeros((4, 6, 6), dtype=torch.int8, device="cuda"), torch.ones(4, dtype=torch.int8, device="cuda"),
                      ev, 16, active=active)
    c = counts.cpu().numpy()
    assert c[0].sum() == 16 and c[2].sum() == 16 and c[1].sum() == 0 and c[3].sum() == 0
    assert np.array_equal(c[0], c[2])
    m.close()


def test_arena_overflow_is_a_status(pkg):
    import torch
    m = pkg.engine.BatchedMCTS(2, 8, 8, 64, edges_per_game=100, nodes_per_game=8)
    ev = HostHashEvaluator([10, 10], [11, 11], m.needs_eval)
    m.search(torch.zeros((2, 8, 8), dtype=torch.int8, device="cuda"), torch.ones(2, dtype=torch.int8, device="cuda"), ev, 64)
    with pytest.raises(pkg.YYError) as ei:
        m.status()
    assert ei.value.code == -6
    m.close()


@pytest.mark.parametrize("sims", [96, 800])
@pytest.mark.parametrize("copied", [1, 0], ids=["copied", "aliased"])
def test_full_size_batch_4096_games(pkg, copied, sims):
    """BASELINE config[1] at its full size -- G = 4096 concurrent 8x8 games, 800 simulations (and a 96-simulation case) -- searched together (device-side hash
    evaluator, fused step kernel), a random subset checked bit-for-bit against the CPU oracle, plus
    size-independent invariants on all games: root.visits == sims, sum(counts) == sims for non-terminal
    roots, pi sums to 1, counts only on legal moves, boards unchanged in copied mode."""
    import torch
    from hash_eval import hash_eval_torch
    G, R, C = 4096, 8, 8
    rng = np.random.default_rng(77 + copied)
    # staggered positions: random legal play for (g mod 40) plies, generated with the HIP rules kernels
    E = pkg.engine
    boards = torch.zeros((G, R, C), dtype=torch.int8, device="cuda")
    players = torch.ones(G, dtype=torch.int8, device="cuda")
    target = torch.arange(G, device="cuda") % 40
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    for ply in range(40):
        mask = E.valid_mask(boards, players).float()
        adv = (ply < target) & (mask.sum(1) > 0)
        act = torch.multinomial(torch.where(adv[:, None], mask, torch.ones_like(mask)), 1, generator=gen).reshape(-1).to(torch.int32)
        act = torch.where(adv, act, torch.full_like(act, -1)).contiguous()
        old = players.clone()
        E.step_(boards, players, act)
        players = torch.where(adv, players, old).contiguous()
    b0 = boards.clone()
    m = E.BatchedMCTS(G, R, C, sims, aliased=not copied)
    counts = m.search(boards, players, lambda p: hash_eval_torch(p, 6, 4), sims)
    pi = m.root_policy()
    visits, _ = m.root_stats()
    fb = m.boards()
    ctr = m.status()
    c, pi_h = counts.cpu().numpy(), pi.cpu().numpy()
    legal = E.valid_mask(b0, players).cpu().numpy()
    ended = E.game_ended(b0, players).cpu().numpy()
    assert (visits.cpu().numpy() == sims).all()
    assert ((c.sum(1) == sims) | (legal.sum(1) == 0) | (ended != 0)).all()
    assert np.allclose(pi_h.sum(1), 1.0)
    assert ((c > 0) <= (legal > 0)).all()
    if copied:
        assert torch.equal(fb, b0)
    assert ctr["evals"] <= G * sims and ctr["nodes"] <= ctr["evals"]
    bh, ph = b0.cpu().numpy(), players.cpu().numpy()
    for g in rng.choice(G, size=48, replace=False):
        r = O.search_hash(bh[g], int(ph[g]), sims, copied, 6, 4)
        assert np.array_equal(c[g], r.counts), g
        assert np.array_equal(fb[g].cpu().numpy(), r.final_board), g
    m.close()


def _late_positions(E, G, R, C, max_ply, seed, rowcol=False):
    """random legal play for (g mod max_ply) plies with the HIP rules kernels; a side without a move passes"""
    import torch
    boards = torch.zeros((G, R, C), dtype=torch.int8, device="cuda")
    players = torch.ones(G, dtype=torch.int8, device="cuda")
    target = torch.arange(G, device="cuda") % max_ply
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    for ply in range(max_ply):
        mask = E.valid_mask(boards, players, rowcol).float()
        has = mask.sum(1) > 0
        act = torch.multinomial(torch.where(has[:, None], mask, torch.ones_like(mask)), 1, generator=gen).reshape(-1).to(torch.int32)
        act = torch.where(has, act, torch.full_like(act, -1)).contiguous()
        old_b, old_p = boards.clone(), players.clone()
        E.step_(boards, players, act, rowcol)                      # an action of -1 places nothing and flips the side: a pass
        adv = ply < target
        boards = torch.where(adv[:, None, None], boards, old_b).contiguous()
        players = torch.where(adv, players, old_p).contiguous()
    return boards, players


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("shape,G,sims,max_ply", [((8, 8), 768, 200, 60), ((12, 12), 96, 150, 130), ((6, 6), 256, 120, 34)],
                         ids=["8x8", "12x12", "6x6"])
def test_evaluation_reuse_returns_the_same_search(pkg, fused, shape, G, sims, max_ply):
    """YY_FLAG_REUSE_PASS_VALUE / YY_FLAG_REUSE_TRANSPOSITIONS (include/yy_engine.h).  The reference evaluates a childless
    non-terminal node again on every visit (ai/mcts.py:93-95, 371-397) and every leaf whose position another node of the search
    already holds (:385-397); the flags take those evaluations from the tree.  Positions from the opening to the end of the game
    (pass ROOTS and finished games included; one, two and three bitboard words): visit counts, root statistics and pi are
    identical to the plain search and to the oracle (= the reference's evaluation sequence), the evaluator is asked for exactly
    `reused_values + transposition_hits` fewer rows, and needs_eval flags exactly the rows that are evaluated."""
    import torch
    from hash_eval import hash_eval_torch
    E = pkg.engine
    R, C = shape
    boards, players = _late_positions(E, G, R, C, max_ply, 9)
    legal = E.valid_mask(boards, players).cpu().numpy()
    ended = E.game_ended(boards, players).cpu().numpy()
    pass_roots = (legal.sum(1) == 0) & (ended == 0)
    if shape == (8, 8):
        assert pass_roots.sum() >= 3 and (ended != 0).sum() >= 3          # the batch holds pass roots and finished games too
    res = {}
    for tag, kw in (("plain", {}), ("pass", dict(reuse_pass_value=True)),
                    ("both", dict(reuse_pass_value=True, reuse_transpositions=True)), ("tt", dict(reuse_transpositions=True)),
                    ("keep", dict(reuse_pass_value=True, reuse_transpositions=True, keep_evaluations=True)),
                    ("keep2", dict(reuse_pass_value=True, reuse_transpositions=True, keep_evaluations=True))):
        m = E.BatchedMCTS(G, R, C, sims, **kw)
        rows = []

        def ev(planes, m=m, rows=rows):
            rows.append(int(m.needs_eval.sum()))
            return hash_eval_torch(planes, 6, 4)

        if tag == "keep2":            # the same search again on a context that keeps its evaluations: every leaf is cached
            m.search(boards, players, lambda p: hash_eval_torch(p, 6, 4), sims, fused=fused)
            m.reset_counters()
        counts = m.search(boards, players, ev, sims, fused=fused)
        visits, wsum = m.root_stats()
        res[tag] = (counts.cpu().numpy(), visits.cpu().numpy(), wsum.cpu().numpy(), m.root_policy().cpu().numpy(), m.status(), rows)
        m.close()
    c0, n0, w0, p0, k0, r0 = res["plain"]
    assert k0["reused_values"] == 0 and k0["transposition_hits"] == 0 and sum(r0[1:]) == k0["evals"]
    for tag in ("pass", "both", "tt", "keep", "keep2"):
        c1, n1, w1, p1, k1, r1 = res[tag]
        assert np.array_equal(c0, c1) and np.array_equal(n0, n1) and np.array_equal(w0, w1) and np.array_equal(p0, p1), tag
        assert k1["evals"] + k1["reused_values"] + k1["transposition_hits"] == k0["evals"], (tag, k0, k1)
        assert k1["nodes"] == k0["nodes"] and k1["terminal_revisits"] == k0["terminal_revisits"] and k1["children_created"] == k0["children_created"]
        assert sum(r1[1:]) == k1["evals"]                          # (call 0 = the root call, mcts.py:288)
    assert res["pass"][4]["reused_values"] > 0 and res["pass"][4]["transposition_hits"] == 0
    assert res["tt"][4]["reused_values"] == 0 and res["tt"][4]["transposition_hits"] > 0
    both = res["both"][4]
    assert both["transposition_hits"] > 0 and both["reused_values"] > 0
    assert res["keep"][4]["evals"] == both["evals"]                  # a fresh context: nothing to find from earlier searches
    assert res["keep2"][4]["evals"] == 0                             # everything the search needs was evaluated by the first one
    print("%dx%d, %d sims: evaluator rows %d plain, %d with pass values, %d with both (-%.1f %%)" % (
        R, C, sims, k0["evals"], res["pass"][4]["evals"], both["evals"], 100 - 100.0 * both["evals"] / k0["evals"]))
    bh, ph = boards.cpu().numpy(), players.cpu().numpy()
    c1 = res["both"][0]
    for g in list(np.flatnonzero(pass_roots)[:3]) + list(np.random.default_rng(3).choice(G, 24, replace=False)):
        r = O.search_hash(bh[g], int(ph[g]), sims, 1, 6, 4)
        assert np.array_equal(c1[g], r.counts), g


def test_evaluation_cache_replacement_keeps_results(pkg):
    """A tiny cache (sized from nodes_per_game: 128 slots per game) across eight consecutive searches of advancing positions:
    the table fills, entries of positions that cannot recur are replaced, then live ones; every search still returns what the
    plain search returns, and later searches find positions of earlier ones."""
    import torch
    from hash_eval import hash_eval_torch
    E = pkg.engine
    G, R, C, sims = 128, 8, 8, 30
    boards, players = _late_positions(E, G, R, C, 40, 4)
    ev = lambda p: hash_eval_torch(p, 6, 4)
    plain = E.BatchedMCTS(G, R, C, sims)
    cached = E.BatchedMCTS(G, R, C, sims, reuse_pass_value=True, reuse_transpositions=True, keep_evaluations=True, nodes_per_game=32)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    hits = []
    for rnd in range(8):
        c0 = plain.search(boards, players, ev, sims)
        w0 = plain.root_stats()[1]
        cached.reset_counters()
        c1 = cached.search(boards, players, ev, sims)
        assert torch.equal(c0, c1) and torch.equal(w0, cached.root_stats()[1]), rnd
        hits.append(cached.status()["transposition_hits"])
        mask = E.valid_mask(boards, players).float()
        has = mask.sum(1) > 0
        act = torch.multinomial(torch.where(has[:, None], mask, torch.ones_like(mask)), 1, generator=gen).reshape(-1).to(torch.int32)
        E.step_(boards, players, torch.where(has, act, torch.full_like(act, -1)).contiguous())
    plain.status()
    plain.close()
    cached.close()
    assert min(hits[1:]) > 0, hits


def test_opening_book_enumerates_the_reachable_positions(pkg):
    """OpeningBook's breadth-first enumeration against the oracle's rules: the number of distinct positions alternating legal
    play reaches with 1..6 stones on 8x8 (64, 4 032, 6 944, 11 848, 30 056, 75 432: same-colour stones must stay connected, so
    the counts are far below the binomials), every one stored, and the stored evaluation of a sample is the evaluator's."""
    import torch
    from hash_eval import hash_eval_torch
    E = pkg.engine
    ev = lambda p: hash_eval_torch(p, 6, 4)
    book = E.OpeningBook(8, 8, ev, 6)
    assert book.n == 64 + 4032 + 6944 + 11848 + 30056 + 75432 and book.stored >= 0.9995 * book.n
    # level by level with the oracle's legal masks (CPU restatement), the first three stone counts
    b = np.zeros((1, 8, 8), np.int8)
    player, want = 1, []
    for _ in range(3):
        m = O.valid_mask(b, np.full(len(b), player, np.int8))
        bi, a = np.nonzero(m)
        c = b[bi].reshape(len(bi), -1).copy()
        c[np.arange(len(bi)), a] = player
        b = np.unique(c, axis=0).reshape(-1, 8, 8)
        want.append(len(b))
        player = -player
    assert want == [64, 4032, 6944]
    # stored rows = the evaluator on that position
    used = torch.nonzero(book.meta).reshape(-1)[:: max(1, book.n // 500)]
    black, white = book.table_keys[used, 0].contiguous()[None], book.table_keys[used, 1].contiguous()[None]
    boards = E.unpack_boards(black, white, 8, 8)
    p, v = ev(E.encode_planes(boards))
    assert torch.equal(p, book.policy[used]) and torch.equal(v, book.value[used])


@pytest.mark.parametrize("shape,stones,G,rowcol", [((8, 8), 5, 512, False), ((6, 6), 6, 256, False), ((12, 12), 3, 96, False),
                                                   ((5, 5), 6, 128, True)], ids=["8x8", "6x6", "12x12 (3 words)", "5x5 row/column rule"])
def test_opening_book_returns_the_same_search(pkg, shape, stones, G, rowcol):
    """Searches from opening positions (0..9 plies) with the shared book of pre-evaluated positions (yy_mcts_set_book), alone
    and on top of the per-game evaluation cache: the same visit counts, value sums and pi as the plain search and the oracle;
    the rows the evaluator is spared are counted (counters[7])."""
    import torch
    from hash_eval import hash_eval_torch
    E = pkg.engine
    R, C = shape
    sims = 150
    ev = lambda p: hash_eval_torch(p, 6, 4)
    book = E.OpeningBook(R, C, ev, stones, rowcol=rowcol)
    boards, players = _late_positions(E, G, R, C, 10, 21, rowcol=rowcol)
    res = {}
    for tag, kw, use_book in (("plain", {}, False), ("book", {}, True),
                              ("book+cache", dict(reuse_pass_value=True, reuse_transpositions=True, keep_evaluations=True), True)):
        m = E.BatchedMCTS(G, R, C, sims, rowcol=rowcol, **kw)
        if use_book:
            m.set_book(book)
        c = m.search(boards, players, ev, sims)
        res[tag] = (c.cpu().numpy(), m.root_stats()[1].cpu().numpy(), m.root_policy().cpu().numpy(), m.status())
        if use_book and tag == "book":          # unset: the plain search again on the same context
            m.set_book(None)
            m.reset_counters()
            c2 = m.search(boards, players, ev, sims)
            assert torch.equal(c, c2) and m.status()["transposition_hits"] == 0
        m.close()
    k0 = res["plain"][3]
    for tag in ("book", "book+cache"):
        for i in range(3):
            assert np.array_equal(res["plain"][i], res[tag][i]), (tag, i)
        k1 = res[tag][3]
        assert k1["evals"] + k1["reused_values"] + k1["transposition_hits"] == k0["evals"] and k1["nodes"] == k0["nodes"]
    assert res["book"][3]["transposition_hits"] > 0                        # the shallow leaves live in the book
    assert res["book+cache"][3]["evals"] < res["book"][3]["evals"]
    bh, ph = boards.cpu().numpy(), players.cpu().numpy()
    for g in np.random.default_rng(5).choice(G, 16, replace=False):
        assert np.array_equal(res["book+cache"][0][g], O.search_hash(bh[g], int(ph[g]), sims, 1, 6, 4, flags=int(rowcol)).counts), g


def test_evaluation_cache_with_aliased_boards_and_clear(pkg):
    """The cache is keyed by the position, so it also serves the literal aliased-board search (the shared board a leaf is
    evaluated on is looked up as it is at that moment); yy_mcts_cache_clear forgets everything (a new network)."""
    import torch
    from hash_eval import hash_eval_torch
    E = pkg.engine
    G, R, C, sims = 256, 8, 8, 120
    boards, players = _late_positions(E, G, R, C, 56, 2)
    ev = lambda p: hash_eval_torch(p, 6, 4)
    out = {}
    for tag, kw in (("plain", {}), ("cache", dict(reuse_transpositions=True, keep_evaluations=True))):
        m = E.BatchedMCTS(G, R, C, sims, aliased=True, **kw)
        b = boards.clone()
        c = m.search(b, players, ev, sims)
        out[tag] = (c.cpu().numpy(), m.root_stats()[1].cpu().numpy(), m.boards().cpu().numpy(), m.status())
        if tag == "cache":
            m.reset_counters()
            m.search(boards.clone(), players, ev, sims)
            again = m.status()
            m.clear_evaluation_cache()
            m.reset_counters()
            m.search(boards.clone(), players, ev, sims)
            cleared = m.status()
        m.close()
    for i in range(3):
        assert np.array_equal(out["plain"][i], out["cache"][i]), i
    k0, k1 = out["plain"][3], out["cache"][3]
    assert k1["transposition_hits"] > 0 and k1["evals"] + k1["transposition_hits"] == k0["evals"]
    assert again["evals"] == 0 and cleared["evals"] == k1["evals"]


def test_pass_value_reuse_needs_copied_boards(pkg):
    with pytest.raises(pkg._lib.YYError) as ei:      # the pass value lives in the NODE, whose position changes with aliased boards
        pkg.engine.BatchedMCTS(4, 6, 6, 10, aliased=True, reuse_pass_value=True)
    assert ei.value.code == -1


def test_search_with_rowcol_rule_vs_oracle(pkg):
    """YY_FLAG_ROWCOL inside the tree kernels (browser-only rule; pinned by the oracle only)."""
    import torch
    R, C, sims, G = 5, 5, 80, 24
    rng = np.random.default_rng(31)
    boards = np.zeros((G, R, C), np.int8)
    pl = np.ones(G, np.int8)
    for ply in range(14):
        m = O.valid_mask(boards, pl, flags=1)
        go = (rng.random(G) < 0.9) & m.any(1)
        act = np.array([rng.choice(np.flatnonzero(r)) if r.any() else 0 for r in m], np.int32)
        nb, npl, _ = O.next_state(boards, pl, act, flags=1)
        boards[go], pl[go] = nb[go], npl[go]
    for copied in (1, 0):
        m = pkg.engine.BatchedMCTS(G, R, C, sims, aliased=not copied, rowcol=True)
        ev = HostHashEvaluator([6] * G, [4] * G, m.needs_eval)
        counts = m.search(torch.from_numpy(boards).cuda(), torch.from_numpy(pl).cuda(), ev, sims).cpu().numpy()
        fb = m.boards().cpu().numpy()
        m.status()
        for g in range(G):
            r = O.search_hash(boards[g], int(pl[g]), sims, copied, 6, 4, flags=1)
            assert np.array_equal(counts[g], r.counts), (copied, g)
            assert np.array_equal(fb[g], r.final_board), (copied, g)
        m.close()


# ---- HIP search with a LIVE GPU evaluator against the reference's own searches (CPU float32 network), 800 simulations.
# north_star: "within 1e-5 on returned pi/v versus the reference MCTS given a frozen network".  pi is visit counts / 800, so
# "within 1e-5" means IDENTICAL visit counts; one flipped arg-max between two near-tied children moves pi by >= 1.25e-3.
# Bounds per evaluator mode: (max |pi - pi_ref| allowed, max |root value - ref| allowed, min fraction of roots whose visit
# counts are identical to the reference's).  The measured figures of the run are printed and written to
# gpurun_out/live_eval_parity.json; the fp32-grade modes are held to the north-star bound on every root.
LIVE_BOUNDS = {
    "fp32":   (1e-5, 1e-5, 1.0),
    "fp32t":  (1e-5, 1e-5, 1.0),
    "f16x3":  (1e-5, 1e-5, 1.0),
    "f16x3r": (1e-5, 1e-5, 1.0),     # the round-2 32x32x16 tower kernel (A/B partner of the general kernel)
    "bf16":   (None, None, 0.0),     # 8 significant bits: reported
}


# (fixture, modes): the reference's own searches with its CPU float32 network (tests/golden/make_golden.py: NET_SEARCH_SETS) --
# BASELINE config 2 (8x8 / 800), config 4 (12x12 / 1600), config 1 (6x6 / 25, plus 800), and three shapes outside the defaults
# (non-square boards, 32 / 64 / 96 channels) that `--nn auto` must also run at reference precision.
LIVE_SETS = [("net800_8x8", list(LIVE_BOUNDS)),
             ("net1600_12x12", ["f16x3", "f16x3r", "fp32"]),
             ("net800_6x6", ["f16x3", "f16x3r", "fp32"]),
             ("net25_6x6", ["f16x3", "f16x3r", "fp32"]),
             ("net400_10x10_c64b4", ["f16x3", "fp32"]),
             ("net400_5x7_c96b3", ["f16x3", "fp32"]),
             ("net200_9x12_c32b2", ["f16x3", "fp32"])]


@pytest.mark.parametrize("name,mode", [(n, m) for n, ms in LIVE_SETS for m in ms])
def test_live_gpu_evaluator_search_vs_reference_pi(pkg, name, mode):
    import json
    import torch
    z = np.load(os.path.join(GOLDEN, "search_%s.npz" % name))
    R, C = z["root_board"].shape[1:]
    sims = int(z["sims"][0])
    shape = tuple(int(x) for x in z["net_shape"]) if "net_shape" in z.files else (128, 10)
    torch.manual_seed(0)                                             # the generator's seed: same weights as the reference's net
    game = pkg.YinYangGame(R, C)
    net = pkg.YinYangNeuralNetwork(game, *shape).cuda().eval()
    if mode == "f16x3":
        assert pkg.BatchedEvaluator(net).mode == "f16x3"             # what `--nn auto` runs for this shape
    ev = pkg.BatchedEvaluator(net, mode)
    n = z["counts"].shape[0]
    dpi, dv, same = np.zeros(n), np.zeros(n), np.zeros(n, bool)
    rp, rv = ev(pkg.engine.encode_planes(torch.from_numpy(z["root_board"]).cuda()))
    e_root_p = float((rp.cpu().numpy() - z["root_policy"]).__abs__().max())
    e_root_v = float(np.abs(rv.cpu().numpy() - z["root_value"]).max())
    for copied in (0, 1):
        for has_noise in (0, 1):
            idx = np.flatnonzero((z["copied"] == copied) & (z["has_noise"] == has_noise))
            mc = pkg.MCTS(game, ev, num_simulations=sims, board_semantics="copied" if copied else "aliased")
            boards = torch.from_numpy(z["root_board"][idx]).cuda()
            players = torch.ones(len(idx), dtype=torch.int8, device="cuda")
            noise = torch.from_numpy(z["noise"][idx]).cuda() if has_noise else None
            pi, ctx = mc.search_batch(boards, players, noise=noise)
            counts = ctx.root_counts().cpu().numpy()
            visits, wsum = ctx.root_stats()
            ctx.status()
            assert (visits.cpu().numpy() == sims).all()
            dpi[idx] = np.abs(pi.cpu().numpy() - z["pi"][idx]).max(1)
            dv[idx] = np.abs(wsum.cpu().numpy() - z["root_w"][idx]) / float(sims)       # root value = value_sum / visits
            same[idx] = (counts == z["counts"][idx]).all(1)
            mc.close()
    rec = dict(fixture=name, board="%dx%d" % (R, C), net="%dx%d" % shape, mode=mode, roots=int(n), sims=sims,
               identical_visit_counts=float(same.mean()), max_dpi=float(dpi.max()),
               median_dpi=float(np.median(dpi)), max_dvalue=float(dv.max()), root_policy_err=e_root_p, root_value_err=e_root_v)
    print("live evaluator parity:", json.dumps(rec))
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "live_eval_parity.json")
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec["%s/%s" % (name, mode)] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    bpi, bv, frac = LIVE_BOUNDS[mode]
    assert same.mean() >= frac, rec
    if bpi is not None:
        assert dpi.max() <= bpi and dv.max() <= bv, rec


def test_live_gpu_evaluator_with_evaluation_reuse_vs_reference_pi(pkg):
    """The 64 recorded 800-simulation searches of the reference (search_net800_8x8: both board semantics, with and without
    root noise) with the LIVE split-f16 evaluator AND the evaluation reuse on: the reference's visit counts on every root,
    pi / root value within 1e-5, while the evaluator is asked for each position once."""
    import torch
    z = np.load(os.path.join(GOLDEN, "search_net800_8x8.npz"))
    torch.manual_seed(0)
    game = pkg.YinYangGame(8, 8)
    ev = pkg.BatchedEvaluator(pkg.YinYangNeuralNetwork(game).cuda().eval(), "f16x3")
    rows_plain = rows_reuse = 0
    for copied in (0, 1):
        for has_noise in (0, 1):
            idx = np.flatnonzero((z["copied"] == copied) & (z["has_noise"] == has_noise))
            boards = torch.from_numpy(z["root_board"][idx]).cuda()
            players = torch.ones(len(idx), dtype=torch.int8, device="cuda")
            noise = torch.from_numpy(z["noise"][idx]).cuda() if has_noise else None
            for reuse in (False, True):
                mc = pkg.MCTS(game, ev, num_simulations=800, board_semantics="copied" if copied else "aliased", evaluation_reuse=reuse)
                pi, ctx = mc.search_batch(boards.clone(), players, noise=noise)
                counts = ctx.root_counts().cpu().numpy()
                _, wsum = ctx.root_stats()
                k = ctx.status()
                assert np.array_equal(counts, z["counts"][idx]), (copied, has_noise, reuse)
                assert np.abs(pi.cpu().numpy() - z["pi"][idx]).max() <= 1e-5
                assert (np.abs(wsum.cpu().numpy() - z["root_w"][idx]) / 800.0).max() <= 1e-5
                if reuse:
                    rows_reuse += k["evals"]
                    assert k["transposition_hits"] > 0
                else:
                    rows_plain += k["evals"]
                mc.close()
    print("64 reference roots x 800 sims: evaluator rows %d plain, %d with evaluation reuse" % (rows_plain, rows_reuse))
    assert rows_reuse < rows_plain


def test_nan_from_the_evaluator_is_a_sticky_error(pkg):
    """A NaN policy or value at search k of a multi-search run must surface at the NEXT status call even though later
    searches (yy_mcts_begin) ran cleanly in between; status clears it."""
    import torch
    from hash_eval import hash_eval_torch
    G = 6
    for what in ("policy", "value"):
        m = pkg.engine.BatchedMCTS(G, 6, 6, 12)
        boards = torch.zeros((G, 6, 6), dtype=torch.int8, device="cuda")
        players = torch.ones(G, dtype=torch.int8, device="cuda")
        state = {"calls": 0, "poison": False}

        def ev(planes):
            p, v = hash_eval_torch(planes, 6, 4)
            state["calls"] += 1
            if state["poison"] and what == "policy" and state["calls"] == 1:     # the root's priors: read by the first selection
                p[2] = float("nan")
            if state["poison"] and what == "value" and state["calls"] == 4:
                v[2] = float("nan")
            return p, v

        m.search(boards, players, ev, 12)
        m.status()
        state.update(calls=0, poison=True)
        m.search(boards, players, ev, 12)            # game 2 is poisoned in this search
        state.update(calls=0, poison=False)
        counts = m.search(boards, players, ev, 12)   # a clean search afterwards: begin resets the per-search flag
        assert int(counts[2].sum()) == 12
        with pytest.raises(pkg.YYError):
            m.status()
        m.status()                                   # cleared by the status call that reported it
        m.close()

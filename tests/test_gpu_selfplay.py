"""GPU: the reference-API layer (MCTS, SelfPlayWorker, YinYangGame) and the batched SelfPlayEngine.

* SelfPlayWorker.play_game against the reference's own transcripts (G4 fixtures): identical seeds =>
  identical searched boards, pi (float64, tolerance 0), actions and value labels, for the literal
  semantics (aliased boards + Q4/Q5 quirks) and for the copied-board adapter.
* the reference's mcts_tests.py known-answer cases restated against this API (real 3x3 game).
* SelfPlayEngine invariants at scale (size-independent properties) + agreement with the oracle-driven
  episode loop in non-quirk mode.
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from hash_eval import hash_eval_np
from oracle_episode import play_game as oracle_play_game
from test_oracle_episodes import check_transcript

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPIS = sorted(glob.glob(os.path.join(GOLDEN, "episodes_*.npz")))


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available()
    import yinyang_game_alphazero_amd as pkg
    return pkg


class HashNet:
    """predict(board) like the real net: (np.float32[A], np.float32)."""

    def __init__(self, pbits, vbits):
        self.pbits, self.vbits, self.calls = pbits, vbits, 0

    def predict(self, board):
        self.calls += 1
        return hash_eval_np(board.get_board(), self.pbits, self.vbits)


def _traced_worker(pkg, R, C, sims, copied, pbits, vbits, quirks=True):
    game = pkg.YinYangGame(R, C)
    w = pkg.SelfPlayWorker(game, "/nonexistent", num_simulations=sims, neural_net=HashNet(pbits, vbits),
                           board_semantics="copied" if copied else "aliased", reference_quirks=quirks)
    trace = dict(search_boards=[], pis=[], actions=[], players=[])
    orig_search, orig_next = w.mcts.search, game.getNextState

    def search(board, player, add_exploration_noise=False):
        trace["search_boards"].append(board.get_board())
        pi, root = orig_search(board, player, add_exploration_noise)
        trace["pis"].append(pi)
        return pi, root

    def nxt(board, player, action):
        trace["actions"].append(int(action))
        trace["players"].append(int(player))
        return orig_next(board, player, action)

    w.mcts.search, game.getNextState = search, nxt
    return w, trace


@pytest.mark.parametrize("path", EPIS, ids=[os.path.basename(p) for p in EPIS])
def test_play_game_matches_reference_transcripts(pkg, path):
    z = np.load(path)
    R, C = z["search_boards"].shape[2:]
    for i in range(z["n"].shape[0]):
        w, trace = _traced_worker(pkg, R, C, int(z["sims"][i]), int(z["copied"][i]), int(z["pbits"][i]), int(z["vbits"][i]))
        np.random.seed(int(z["seed"][i]))
        examples = w.play_game()
        t = dict(trace, n=len(examples), z=[e[2] for e in examples], example_boards=[e[0].get_board() for e in examples])
        check_transcript(z, i, t)
        for e, pi in zip(examples, trace["pis"]):
            assert e[1] is pi or np.array_equal(e[1], pi)
        w.mcts.close()


def test_play_game_real_player_mode_matches_oracle(pkg):
    """non-quirk mode (real side to move searched, alternating labels): no reference implementation
    exists, pinned by the oracle-driven episode loop."""
    for seed, (R, C), sims in ((11, (6, 6), 30), (12, (5, 7), 20), (13, (8, 8), 24)):
        w, trace = _traced_worker(pkg, R, C, sims, 1, 6, 4, quirks=False)
        np.random.seed(seed)
        examples = w.play_game()
        want = oracle_play_game(R, C, seed, sims, 1, 6, 4, quirks=False)
        assert len(examples) == want["n"] > 4
        assert trace["actions"] == want["actions"] and trace["players"] == want["players"]
        assert np.array_equal(np.stack(trace["pis"]), np.stack(want["pis"]))
        assert [e[2] for e in examples] == want["z"]
        assert set(abs(v) for v in want["z"]) <= {1, 1e-4}
        w.mcts.close()


# ---- src/yin_yang/ai/mcts_tests.py restated on the real game (the kernels implement the rules, so the
# SimpleGame fakes are replaced by YinYangGame(3,3) and a mock evaluator with np.float32 outputs)
class MockNet:
    def __init__(self, A, value=0.0):
        self.policy, self.value, self.predict_calls = np.full(A, 1.0 / A, np.float32), np.float32(value), 0

    def predict(self, board):
        self.predict_calls += 1
        return self.policy.copy(), self.value


@pytest.mark.parametrize("semantics", ["copied", "aliased"])
def test_search_invariants_like_reference_tests(pkg, semantics):
    game = pkg.YinYangGame(3, 3)
    net = MockNet(9)
    m = pkg.MCTS(game, net, num_simulations=10, verbose=0, board_semantics=semantics)
    board = game.getInitBoard()
    pi, root = m.search(board, 1)
    assert abs(pi.sum() - 1.0) < 1e-12                     # mcts_tests.py:219-220
    assert root.is_expanded() and root.visits == 10        # :223-226
    assert len(root.children) == 9                         # :97 (all 9 moves legal on the empty board)
    for a, ch in root.children.items():                    # :100-103, :332-335
        assert ch.parent is root and ch.action == a and isinstance(ch.visits, int)
        assert abs(float(ch.prior) - 1.0 / 9.0) < 1e-7
    assert net.predict_calls >= 2
    np.random.seed(3)
    pi2, root2 = m.search(game.getInitBoard(), 1, add_exploration_noise=True)   # :228-235
    assert abs(pi2.sum() - 1.0) < 1e-12 and root2.visits == 10
    a = m.select_action(game.getInitBoard(), 1, temperature=0)                  # :292-297
    assert 0 <= int(a) < 9
    old = m.reuse_tree(root, board, -1, 0)                 # :299-321
    assert old.parent is None
    assert m.reuse_tree(root, board, -1, 9999).visits == 0
    m.close()


def test_node_built_by_hand_like_reference_tests(pkg):
    """mcts_tests.py:84-157 (TestNode: init, expand, expand_terminal, select_child, update, get_value) restated with the
    reference's constructor `Node(game)` on the real 3x3 game (rules from the HIP kernels through the game shim)."""
    game = pkg.YinYangGame(3, 3)
    node = pkg.Node(game)
    assert node.game is game and node.parent is None and node.action is None and node.children == {}
    assert node.visits == 0 and node.value_sum == 0.0 and node.prior == 0.0 and not node.is_expanded()
    board = game.getInitBoard()
    policy = np.ones(9, np.float32) / 9
    node.expand(board, 1, policy)
    assert node.board is board and node.player == 1 and not node.is_terminal and node.is_expanded()
    assert sorted(node.children) == list(range(9)) and all(ch.prior == policy[a] and ch.parent is node and ch.action == a
                                                           for a, ch in node.children.items())
    assert np.array_equal(node.valid_moves, np.ones(9))
    other = pkg.Node(game)
    other.expand(board, 1, None)                                   # no policy: uniform over the legal moves (:81)
    assert all(abs(ch.prior - 1.0 / 9) < 1e-15 for ch in other.children.values())
    # terminal expansion (:63-71): a full board without legal moves for either colour
    full = pkg.YinYangLogic(3, 3)
    full.board = np.array([[1, -1, 1], [-1, 1, -1], [1, -1, -1]], np.int8)
    term = pkg.Node(game)
    term.expand(full, 1, policy)
    assert term.is_terminal and term.is_expanded() and term.children == {} and term.terminal_value == game.getGameEnded(full, 1) != 0
    # select_child / update / get_value on the expanded node (:117-157)
    node.children[4].visits, node.children[4].value_sum = 3, np.float32(2.4)
    a, ch = node.select_child(c_puct=1.0)
    assert a == 4 and ch is node.children[4]                       # q = 0.8 dominates the equal priors
    ch.update(0.5)
    assert ch.visits == 4 and abs(float(ch.value_sum) - 2.9) < 1e-6 and abs(float(ch.get_value()) - 0.725) < 1e-6
    assert pkg.Node(game).get_value() == 0.0


def test_terminal_root_and_only_move(pkg):
    game = pkg.YinYangGame(3, 3)
    # full board: terminal root, value from the root player's view, no children, uniform pi (:252-269, mcts.py:209-213)
    b = game.getInitBoard()
    b.board[...] = np.array([[1, -1, 1], [-1, 1, -1], [1, -1, 1]], np.int8)
    m = pkg.MCTS(game, MockNet(9), num_simulations=7, verbose=0, board_semantics="copied")
    pi, root = m.search(b, 1)
    assert root.is_terminal and root.terminal_value == 1 and len(root.children) == 0
    assert root.visits == 7 and root.value_sum == 7.0
    assert np.allclose(pi, 1.0 / 9.0)
    # a position with exactly one legal move for black -> pi is one-hot (:477-496)
    found = None
    rng = np.random.default_rng(5)
    for _ in range(4000):
        arr = rng.integers(-1, 2, size=(1, 3, 3)).astype(np.int8)
        if O.valid_mask(arr, 1)[0].sum() == 1 and O.game_ended(arr, 1)[0] == 0:
            found = arr[0]
            break
    assert found is not None
    b2 = game.getInitBoard()
    b2.board[...] = found
    pi, root = m.search(b2, 1)
    a = int(np.flatnonzero(O.valid_mask(found[None], 1)[0])[0])
    assert np.argmax(pi) == a and abs(pi[a] - 1.0) < 1e-12
    m.close()


def test_game_api_matches_oracle(pkg):
    """YinYangGame / YinYangLogic single-board API (batch-of-one kernel launches) on random boards."""
    rng = np.random.default_rng(9)
    game = pkg.YinYangGame(5, 7)
    for _ in range(40):
        arr = rng.integers(-1, 2, size=(5, 7)).astype(np.int8) * (rng.random((5, 7)) < 0.5)
        arr = arr.astype(np.int8)
        lb = game.getInitBoard()
        lb.board = arr.copy()
        for p in (1, -1):
            assert np.array_equal(game.getValidMoves(lb, p), O.valid_mask(arr[None], p)[0].astype(np.float64))
            assert game.getGameEnded(lb, p) == O.game_ended(arr[None], p)[0]
            assert lb.has_valid_move(p) == bool(O.valid_mask(arr[None], p)[0].any())
        a = int(rng.integers(35))
        nb, npl = game.getNextState(lb, 1, a)
        ob, _, _ = O.next_state(arr[None], 1, np.array([a], np.int32))
        assert nb is lb and npl == -1 and np.array_equal(lb.board, ob[0])
        assert lb.count_pieces() == (int((ob[0] == 1).sum()), int((ob[0] == -1).sum()))
    assert game.getBoardSize() == (5, 7) and game.getActionSize() == 35 and game._action_to_coords(17) == (2, 3)


def test_logic_predicates_compose_to_the_legal_mask(pkg):
    """yin_yang_logic.py:31-56: is_valid_move == in-bounds, empty, and with the stone placed: _check_connectivity(piece) and
    _check_2x2_constraint().  The two host predicates of the shim must compose to exactly the device kernel's mask (which the
    goldens pin to the reference) on random boards, both colours."""
    rng = np.random.default_rng(9)
    for (R, C) in ((5, 5), (4, 7), (8, 8)):
        lb = pkg.YinYangLogic(R, C)
        for _ in range(12):
            lb.board = (rng.integers(-1, 2, size=(R, C)) * (rng.random((R, C)) < rng.uniform(0.2, 0.9))).astype(np.int8)
            for piece in (1, -1):
                mask = lb._mask(piece).reshape(R, C)
                for x in range(R):
                    for y in range(C):
                        ok = False
                        if lb.board[x, y] == 0:
                            lb.board[x, y] = piece
                            ok = lb._check_connectivity(piece) and lb._check_2x2_constraint()
                            lb.board[x, y] = 0
                        assert ok == bool(mask[x, y]), (R, C, x, y, piece)


@pytest.mark.parametrize("use_graph,row_tiers", [(False, ()), (False, (8, 16, 32, 64)), (True, (8, 16, 32, 64))],
                         ids=["full_rows", "packed_tail", "packed_tail_graph"])
def test_engine_full_games_properties(pkg, use_graph, row_tiers):
    """SelfPlayEngine, 96 concurrent 6x6 games to completion, device RNG.  Size-independent properties:
    every example state is reachable (stone counts differ by <= 1 + passes... here: legal position with
    no monochrome 2x2 and both colours connected), pi sums to 1 and is supported on legal moves, labels are
    +-1 / +-1e-4 and alternate with the side to move, per-game plies are 0..n-1."""
    import torch
    game = pkg.YinYangGame(6, 6)

    def ev(planes):   # uniform priors, zero value: cheap deterministic evaluator on the device
        G = planes.shape[0]
        return torch.full((G, 36), 1.0 / 36, device=planes.device), torch.zeros(G, device=planes.device)

    eng = pkg.SelfPlayEngine(game, ev, num_simulations=24, concurrent_games=96, seed=5, use_graph=use_graph,
                             row_tiers=row_tiers)
    ex = eng.run(150)
    assert eng.games_finished == 150
    if row_tiers:       # the draining batch was packed to the front and evaluated on fewer rows
        assert eng.rows < 96 and (not use_graph or len(eng.search.graphs) >= 2)
    st, pi, z, gid, ply = (ex[k].cpu().numpy() for k in ("states", "policies", "values", "game_id", "ply"))
    assert st.shape[0] == pi.shape[0] == z.shape[0] > 150 * 10
    assert np.allclose(pi.sum(1), 1.0, atol=1e-6)
    assert all(min(abs(v - 1.0), abs(v - 1e-4)) < 1e-7 for v in np.unique(np.abs(z)).tolist())
    assert len(np.unique(gid)) == 150
    # pi is supported on the legal moves of the side to move; side to move = black iff equal stone counts... not
    # guaranteed with passes, so check against both colours' masks
    m1, m2 = O.valid_mask(st, 1), O.valid_mask(st, -1)
    sup = pi > 0
    assert (sup <= ((m1 | m2) > 0)).all()
    for g in np.unique(gid)[:40]:
        sel = gid == g
        assert sorted(ply[sel].tolist()) == list(range(int(sel.sum())))
    # every game is a legal trajectory: consecutive recorded states differ by exactly one stone, placed on a cell the
    # oracle calls legal for its colour (passes record nothing, so the mover is read off the new stone)
    for g in np.unique(gid)[:60]:
        sel = np.flatnonzero(gid == g)
        sel = sel[np.argsort(ply[sel])]
        assert not st[sel[0]].any()
        for a, b in zip(sel[:-1], sel[1:]):
            d = st[b].astype(np.int32) - st[a].astype(np.int32)
            cells = np.flatnonzero(d.reshape(-1))
            assert len(cells) == 1 and abs(int(d.reshape(-1)[cells[0]])) == 1 and st[a].reshape(-1)[cells[0]] == 0
            colour = int(d.reshape(-1)[cells[0]])
            assert O.valid_mask(st[a][None], colour)[0][cells[0]] == 1
            assert pi[a][cells[0]] > 0
    # no finished position contains a monochrome 2x2 block
    s = st.astype(np.int32)
    blk = (s[:, :-1, :-1] == s[:, 1:, :-1]) & (s[:, :-1, :-1] == s[:, :-1, 1:]) & (s[:, :-1, :-1] == s[:, 1:, 1:]) & (s[:, :-1, :-1] != 0)
    assert not blk.any()
    eng.close()


@pytest.mark.parametrize("shape,G,games,sims,semantics", [((6, 6), 64, 80, 40, "copied"), ((8, 8), 96, 96, 32, "copied"),
                                                         ((5, 7), 40, 40, 24, "aliased")],
                         ids=["6x6_copied_refill", "8x8_copied_packed_tail", "5x7_aliased"])
def test_engine_every_policy_equals_oracle_search(pkg, shape, G, games, sims, semantics):
    """End-to-end parity of the BATCHED lockstep engine (hipGraph replay, slot refill, packed tail): with the exact hash
    evaluator and no root noise, EVERY recorded pi of every game must be the oracle's search result from that recorded state
    for the side that then moved (tolerance 0 after the float32 cast the engine applies when it stores pi)."""
    import torch
    from hash_eval import hash_eval_torch
    R, C = shape
    game = pkg.YinYangGame(R, C)
    ev = lambda planes: hash_eval_torch(planes, 10, 11)
    eng = pkg.SelfPlayEngine(game, ev, num_simulations=sims, concurrent_games=G, seed=21, dirichlet_epsilon=0.0,
                             board_semantics=semantics, row_tiers=(8, 16, 32, 64))
    ex = eng.run(games)
    assert eng.games_finished == games
    st, pi, gid, ply = (ex[k].cpu().numpy() for k in ("states", "policies", "game_id", "ply"))
    checked = 0
    for g in np.unique(gid):
        sel = np.flatnonzero(gid == g)
        sel = sel[np.argsort(ply[sel])]
        for a, b in zip(sel[:-1], sel[1:]):
            d = (st[b].astype(np.int32) - st[a].astype(np.int32)).reshape(-1)
            cells = np.flatnonzero(d)
            if semantics == "aliased" or len(cells) != 1:
                continue       # aliased: the recorded state is already mutated by its own search (Q2); checked below instead
            mover = int(d[cells[0]])
            r = O.search_hash(st[a], mover, sims, 1, 10, 11, noise=None, eps=0.0)
            assert np.array_equal(r.pi.astype(np.float32), pi[a]), (int(g), int(ply[a]))
            checked += 1
    if semantics == "copied":
        assert checked > games * 5
    else:
        # literal aliased semantics: every game starts from the empty board as black; its first recorded pi must be the
        # oracle's aliased search from the empty board, and the oracle's mutated final board is where play continues from
        r = O.search_hash(np.zeros((R, C), np.int8), 1, sims, 0, 10, 11, noise=None, eps=0.0)
        for g in np.unique(gid):
            first = np.flatnonzero((gid == g) & (ply == 0))
            assert len(first) == 1 and np.array_equal(r.pi.astype(np.float32), pi[first[0]])
    eng.close()


def test_engine_graph_equals_eager(pkg):
    """hipGraph replay of the simulation loop gives the same visit counts as eager launches."""
    import torch
    game = pkg.YinYangGame(8, 8)
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(game, 32, 2).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "fp32")
    from yinyang_game_alphazero_amd.self_play import LockstepSearch
    boards = torch.zeros((32, 8, 8), dtype=torch.int8, device="cuda")
    players = torch.ones(32, dtype=torch.int8, device="cuda")
    outs = []
    for use_graph in (False, True):
        ctx = pkg.engine.BatchedMCTS(32, 8, 8, 64)
        s = LockstepSearch(ctx, ev, use_graph=use_graph)
        s.run(boards, players, 64)
        s.run(boards, players, 64)      # second search replays the captured graph from the start
        outs.append(ctx.root_counts().cpu().numpy())
        ctx.status()
        ctx.close()
    assert np.array_equal(outs[0], outs[1]) and (outs[0].sum(1) == 64).all()


def test_engine_is_reproducible_for_a_seed(pkg):
    """Same seed, same network -> the same examples (device RNG is seeded per engine; the lockstep order of
    operations is fixed, so two runs are bit-identical); a different seed gives different games."""
    import torch
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(game, 128, 1).cuda().eval()
    runs = []
    for seed in (11, 11, 12):
        eng = pkg.SelfPlayEngine(game, pkg.BatchedEvaluator(net, "bf16"), num_simulations=16, concurrent_games=32, seed=seed)
        ex = eng.run(48)
        order = torch.argsort(ex["game_id"] * 1000 + ex["ply"])
        runs.append({k: v[order].cpu() for k, v in ex.items()})
        eng.close()
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    assert runs[0]["states"].shape != runs[2]["states"].shape or not torch.equal(runs[0]["states"], runs[2]["states"])


@pytest.mark.parametrize("shape,G,games,sims", [((8, 8), 320, 320, 48), ((6, 6), 96, 128, 64)], ids=["8x8", "6x6"])
def test_evaluation_reuse_plays_the_same_games(pkg, shape, G, games, sims):
    """The engine's default with the split-f16 evaluator: a node without legal moves is evaluated once (YY_FLAG_REUSE_PASS_VALUE)
    and a leaf whose position the search already holds takes that node's evaluation (YY_FLAG_REUSE_TRANSPOSITIONS), where the
    reference evaluates each of them (ai/mcts.py:93-95, 385-397).  With the LIVE GPU evaluator, whole games (slot refill
    included; the 8x8 batch is above the evaluator's register-ring threshold, so both tower kernels run): every state, pi, z of
    every game is bit-identical with the options on and off, and the evaluator is asked for fewer rows."""
    import torch
    game = pkg.YinYangGame(*shape)
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(game, 128, 2).cuda().eval()
    ev = pkg.BatchedEvaluator(net)
    assert ev.mode == "f16x3" and ev.row_independent
    runs, ctrs = [], []
    for reuse in (None, False):
        eng = pkg.SelfPlayEngine(game, ev, num_simulations=sims, concurrent_games=G, seed=5, reuse_pass_value=reuse,
                                 reuse_transpositions=reuse, keep_evaluations=reuse)
        assert eng.reuse_pass_value == eng.reuse_transpositions == eng.keep_evaluations == (reuse is None)
        ex = eng.run(games)
        order = torch.argsort(ex["game_id"] * 1000 + ex["ply"])
        runs.append({k: v[order].cpu() for k, v in ex.items()})
        ctrs.append(eng.ctx.status())
        eng.close()
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    on, off = ctrs
    assert off["reused_values"] == 0 and off["transposition_hits"] == 0 and on["reused_values"] > 0 and on["transposition_hits"] > 0
    assert on["evals"] + on["reused_values"] + on["transposition_hits"] == off["evals"] and on["nodes"] == off["nodes"]
    print("evaluation reuse, %dx%d, %d games x %d sims: evaluator rows %d -> %d (-%.1f %%: %d pass values, %d transpositions)" % (
        *shape, games, sims, off["evals"], on["evals"], 100 - 100.0 * on["evals"] / off["evals"], on["reused_values"],
        on["transposition_hits"]))


def test_opening_book_plays_the_same_games(pkg):
    """Whole 6x6 games with the live split-f16 evaluator and an opening book of every position with <= 6 stones (built with the
    same evaluator, in batches): the examples are bit-identical to the engine without a book, and the evaluator is spared the
    shallow rows of every game."""
    import torch
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(0)
    ev = pkg.BatchedEvaluator(pkg.YinYangNeuralNetwork(game, 128, 2).cuda().eval())
    runs, ctrs = [], []
    for book in (6, None):
        eng = pkg.SelfPlayEngine(game, ev, num_simulations=48, concurrent_games=96, seed=9, opening_book=book)
        assert (eng.book is not None) == (book is not None)
        ex = eng.run(160)
        order = torch.argsort(ex["game_id"] * 1000 + ex["ply"])
        runs.append({k: v[order].cpu() for k, v in ex.items()})
        ctrs.append(eng.ctx.status())
        eng.close()
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    assert ctrs[0]["evals"] < ctrs[1]["evals"]
    print("opening book <= 6 stones on 6x6: evaluator rows %d -> %d" % (ctrs[1]["evals"], ctrs[0]["evals"]))


# ---- per-game counter-based random streams (csrc/yy_selfplay.hip)
def test_sample_actions_kernel_equals_host_restatement(pkg):
    """yy_selfplay_sample_actions against tests/philox_ref.py (Philox4x32-10 pinned by Random123's known answers): same
    uniform, same float64 running sums in the same order -> the same action, for temperature 1 (incl. zero-mass rows ->
    uniform over the legal moves), temperature 0 (ties), idle games (-1)."""
    import torch
    from philox_ref import sample_action
    rng = np.random.default_rng(3)
    for A in (9, 36, 64, 144):
        G = 96
        pi = rng.random((G, A)) * (rng.random((G, A)) < 0.3)
        pi[::7] = np.round(pi[::7] * 4) / 4            # ties
        mask = (rng.random((G, A)) < 0.5).astype(np.uint8)
        pi[5] = 0.0                                     # no mass on the legal moves
        mask[6] = 0
        pi = pi / np.maximum(pi.sum(1, keepdims=True), 1e-300)
        ply = rng.integers(0, 20, size=G).astype(np.int32)
        gid = rng.integers(0, 2 ** 40, size=G).astype(np.int64)
        searching = (rng.random(G) < 0.8).astype(np.uint8)
        seed = 1000 + A
        got = pkg.engine.sample_actions(seed, torch.from_numpy(gid).cuda(), torch.from_numpy(ply).cuda(),
                                        torch.from_numpy(searching).cuda(), torch.from_numpy(pi).cuda(),
                                        torch.from_numpy(mask).cuda(), 10).cpu().numpy()
        for g in range(G):
            want = sample_action(seed, int(gid[g]), int(ply[g]), pi[g], mask[g], 10) if searching[g] else -1
            assert got[g] == want, (A, g)
            if searching[g] and ply[g] < 10 and mask[g].any():
                assert mask[g][got[g]] == 1


def test_root_noise_is_a_dirichlet_draw_per_game(pkg):
    """noise rows: zero for games that do not draw and on illegal cells, sum to 1 over the legal cells, depend only on
    (seed, game id, ply) -- not on the row or the batch -- and have the Dirichlet(0.3) component mean 1/k and variance
    (1/k)(1-1/k)/(0.3k+1) within sampling error."""
    import torch
    rng = np.random.default_rng(4)
    G, A, k = 4096, 64, 16
    mask = np.zeros((G, A), np.uint8)
    mask[:, :k] = 1
    gid = torch.arange(G, dtype=torch.int64, device="cuda")
    ply = torch.zeros(G, dtype=torch.int32, device="cuda")
    draw = torch.ones(G, dtype=torch.uint8, device="cuda")
    draw[::5] = 0
    n = pkg.engine.root_noise(7, gid, ply, draw, torch.from_numpy(mask).cuda(), 0.3).cpu().numpy()
    on = draw.cpu().numpy().astype(bool)
    assert (n[~on] == 0).all() and (n[:, k:] == 0).all() and (n >= 0).all()
    assert np.allclose(n[on].sum(1), 1.0, atol=1e-12)
    x = n[on][:, :k]
    assert abs(x.mean() - 1 / k) < 1e-3
    var = (1 / k) * (1 - 1 / k) / (0.3 * k + 1)
    assert abs(x.var() - var) < 0.05 * var
    # the same games in another order / batch: identical rows
    perm = torch.from_numpy(rng.permutation(G)[:500]).cuda()
    n2 = pkg.engine.root_noise(7, gid[perm].contiguous(), ply[perm].contiguous(), draw[perm].contiguous(),
                               torch.from_numpy(mask).cuda()[perm].contiguous(), 0.3).cpu().numpy()
    assert np.array_equal(n2, n[perm.cpu().numpy()])
    n3 = pkg.engine.root_noise(8, gid, ply, draw, torch.from_numpy(mask).cuda(), 0.3).cpu().numpy()
    assert not np.array_equal(n3, n)


def test_game_transcripts_do_not_depend_on_slot_batch_or_world_size(pkg):
    """Game g's (states, pi, z) must be identical whether it is played in a batch of 64 slots, through 16 refilled slots
    (different slot, different neighbours, packed tail), or by one of two 'ranks' that each play every second game
    (first_game_index / stride = (r, 2)): noise and moves come from (seed, game id, ply) alone."""
    import torch
    from hash_eval import hash_eval_torch
    game = pkg.YinYangGame(6, 6)
    ev = lambda planes: hash_eval_torch(planes, 10, 11)

    def play(G, n, first, stride):
        eng = pkg.SelfPlayEngine(game, ev, num_simulations=20, concurrent_games=G, seed=1000, first_game_index=first,
                                 game_index_stride=stride, row_tiers=(8, 16, 32))
        ex = eng.run(n)
        eng.close()
        out = {}
        gid, ply = ex["game_id"].cpu().numpy(), ex["ply"].cpu().numpy()
        st, pi, z = ex["states"].cpu().numpy(), ex["policies"].cpu().numpy(), ex["values"].cpu().numpy()
        for g in np.unique(gid):
            sel = np.flatnonzero(gid == g)
            sel = sel[np.argsort(ply[sel])]
            out[int(g)] = (st[sel], pi[sel], z[sel])
        return out

    ref = play(64, 64, 0, 1)
    assert sorted(ref) == list(range(64))
    assert len({v[0].tobytes() for v in ref.values()}) > 32          # the games really differ from one another
    small = play(16, 64, 0, 1)
    two = {**play(32, 32, 0, 2), **play(32, 32, 1, 2)}
    for other in (small, two):
        assert sorted(other) == sorted(ref)
        for g in ref:
            for a, b in zip(ref[g], other[g]):
                assert np.array_equal(a, b), g


def test_lanes_play_the_same_games(pkg):
    """SelfPlayLanes (the batch cut into lanes on separate HIP streams, moves enqueued back to back and pipelined) plays the
    same games as one SelfPlayEngine: every game's (states, pi, z) identical for 1, 2 and 3 lanes, with the exact hash
    evaluator and with the live split-f16 evaluator + evaluation reuse (static per-stream buffers, per-lane caches); the lanes
    split the global game indices as lane k of K = first + (k + j*K) * stride, also inside a 2-rank sharding."""
    import torch
    from hash_eval import hash_eval_torch
    game = pkg.YinYangGame(6, 6)

    def by_game(ex):
        out = {}
        gid, ply = ex["game_id"].cpu().numpy(), ex["ply"].cpu().numpy()
        st, pi, z = ex["states"].cpu().numpy(), ex["policies"].cpu().numpy(), ex["values"].cpu().numpy()
        for g in np.unique(gid):
            sel = np.flatnonzero(gid == g)
            sel = sel[np.argsort(ply[sel])]
            out[int(g)] = (st[sel], pi[sel], z[sel])
        return out

    def same(a, b):
        assert sorted(a) == sorted(b)
        for g in a:
            for x, y in zip(a[g], b[g]):
                assert np.array_equal(x, y), g

    ev = lambda planes: hash_eval_torch(planes, 10, 11)
    one = pkg.SelfPlayEngine(game, ev, num_simulations=20, concurrent_games=48, seed=1000)
    ref = by_game(one.run(80))
    one.close()
    assert sorted(ref) == list(range(80))
    for K in (1, 2, 3):
        lanes = pkg.SelfPlayLanes(game, ev, num_simulations=20, concurrent_games=30, lanes=K, seed=1000, row_tiers=(4, 8))
        assert len(lanes.lanes) == K and lanes.G == 30
        got = by_game(lanes.run(80))
        assert lanes.games_finished == 80 and lanes.n_alive == 0
        lanes.close()
        same(ref, got)
    two = {}
    for r in (0, 1):
        lanes = pkg.SelfPlayLanes(game, ev, num_simulations=20, concurrent_games=16, lanes=2, seed=1000, first_game_index=r, game_index_stride=2)
        two.update(by_game(lanes.run(40)))
        lanes.close()
    same(ref, two)
    # live evaluator (static buffers per stream) + evaluation reuse + a shared opening book
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(game, 32, 1).cuda().eval()
    evn = pkg.BatchedEvaluator(net)
    assert evn.mode == "f16x3"
    one = pkg.SelfPlayEngine(game, evn, num_simulations=24, concurrent_games=40, seed=5)
    ref = by_game(one.run(56))
    one.close()
    lanes = pkg.SelfPlayLanes(game, evn, num_simulations=24, concurrent_games=40, lanes=2, seed=5, opening_book=2)
    assert lanes.reuse_transpositions and lanes.book is not None
    got = by_game(lanes.run(56))
    moves = [lanes.play_move() for _ in range(2)] if False else None
    lanes.close()
    same(ref, got)


@pytest.mark.parametrize("path", EPIS, ids=[os.path.basename(p) for p in EPIS])
def test_batched_engine_replays_reference_transcripts_with_numpy_rng(pkg, path):
    """SelfPlayEngine(rng="numpy"): every game of the batch draws from its own numpy RandomState(seed) exactly where the
    reference's play_game draws from the global stream after np.random.seed(seed) (Dirichlet noise at the first search,
    np.random.choice per move: ai/mcts.py:305, ai/self_play.py:146, 160).  All recorded reference games of one board
    semantics are played TOGETHER in one lockstep batch (different lengths, per-game hash evaluators as recorded) and must
    equal the G4 transcripts: the boards of the examples, pi and z, game by game."""
    import torch
    from hash_eval import hash_eval_batch, planes_to_boards
    z = np.load(path)
    R, C = z["search_boards"].shape[2:]
    game = pkg.YinYangGame(R, C)
    for copied in (0, 1):
        games = np.flatnonzero(z["copied"] == copied)
        pb, vb = z["pbits"][games], z["vbits"][games]

        def ev(planes):                                   # slot g plays game games[g] for the whole run (no refill, no packing)
            b = planes_to_boards(planes.cpu().numpy())
            pol, val = np.zeros((len(games), R * C), np.float32), np.zeros(len(games), np.float32)
            for key in set(zip(pb.tolist(), vb.tolist())):
                idx = np.flatnonzero((pb == key[0]) & (vb == key[1]))
                pol[idx], val[idx] = hash_eval_batch(b[idx], key[0], key[1])
            return torch.from_numpy(pol).cuda(), torch.from_numpy(val).cuda()

        sims = int(z["sims"][games[0]])
        assert (z["sims"][games] == sims).all()
        eng = pkg.SelfPlayEngine(game, ev, num_simulations=sims, concurrent_games=len(games), use_graph=False, compact_tail=False,
                                 board_semantics="copied" if copied else "aliased", reference_quirks=True, rng="numpy",
                                 numpy_seeds={g: int(z["seed"][games[g]]) for g in range(len(games))})
        ex = eng.run(len(games))
        eng.close()
        gid, ply = ex["game_id"].cpu().numpy(), ex["ply"].cpu().numpy()
        st, pi, zz = ex["states"].cpu().numpy(), ex["policies"].cpu().numpy(), ex["values"].cpu().numpy()
        for g, i in enumerate(games):
            n = int(z["n"][i])
            sel = np.flatnonzero(gid == g)
            sel = sel[np.argsort(ply[sel])]
            assert len(sel) == n, (copied, i, len(sel), n)
            assert np.array_equal(st[sel], z["example_boards"][i, :n]), (copied, i)
            assert np.array_equal(pi[sel], z["pis"][i, :n].astype(np.float32)), (copied, i)
            assert np.array_equal(zz[sel], z["z"][i, :n].astype(np.float32)), (copied, i)


def test_evaluation_reuse_across_the_drain_tiers_plays_the_same_games(pkg):
    """1 100 games through 1 024 slots with the live split-f16 evaluator (128 channels): the batch refills, then drains through
    the row tiers (1 024 -> 512 -> 256: packed live games, one captured step per tier, tower forms chosen on the device), and the
    evaluation cache + a shared opening book feed results computed in one batch size into searches running at another.  Every
    state, pi, z of every game is bit-identical with the reuse on and off, on one lane and on two -- the property that the
    evaluator's rows do not depend on the batch (tests/test_gpu_network.py) seen from the engine."""
    import torch
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(0)
    ev = pkg.BatchedEvaluator(pkg.YinYangNeuralNetwork(game, 128, 1).cuda().eval())
    assert ev.mode == "f16x3" and ev.row_independent

    def play(reuse, lanes, book):
        eng = pkg.SelfPlayLanes(game, ev, num_simulations=16, concurrent_games=1024, lanes=lanes, seed=21, opening_book=book,
                                reuse_pass_value=reuse, reuse_transpositions=reuse, keep_evaluations=reuse)
        assert any(t < 1024 // lanes for t in eng.lanes[0].tiers)
        ex = eng.run(1100)
        c = eng.ctx.status()
        eng.close()
        order = torch.argsort(ex["game_id"] * 1000 + ex["ply"])
        return {k: v[order].cpu() for k, v in ex.items()}, c

    ref, c_off = play(False, 1, None)
    assert sorted(set(ref["game_id"].tolist())) == list(range(1100))
    for lanes, book in ((1, 3), (2, 3)):
        got, c_on = play(None, lanes, book)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (lanes, k)
        assert c_on["evals"] < c_off["evals"] and c_on["transposition_hits"] > 0


def test_changing_the_evaluator_clears_kept_evaluations_and_drops_the_book(pkg):
    """Evaluations kept across searches and the opening book are results of ONE network.  An engine that is handed another
    evaluator (search.evaluator = ...) must not serve the old network's numbers: the context is bound to the evaluator object,
    a search with a different one clears the cache and drops the book -- the games it then plays equal a fresh engine's."""
    import torch
    game = pkg.YinYangGame(6, 6)
    evs = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        evs.append(pkg.BatchedEvaluator(pkg.YinYangNeuralNetwork(game, 32, 1).cuda().eval()))
    eng = pkg.SelfPlayEngine(game, evs[0], num_simulations=16, concurrent_games=32, seed=3, opening_book=2)
    assert eng.keep_evaluations and eng.ctx.book is not None
    eng.run(32)                                                  # fills the cache with network 0's evaluations
    eng.evaluator = eng.search.evaluator = evs[1]
    ex = eng.run(32)                                             # games 32..63 with network 1
    assert eng.ctx.book is None
    eng.close()
    fresh = pkg.SelfPlayEngine(game, evs[1], num_simulations=16, concurrent_games=32, seed=3, first_game_index=32)
    want = fresh.run(32)
    fresh.close()
    for e in (ex, want):
        order = torch.argsort(e["game_id"] * 1000 + e["ply"])
        for k in e:
            e[k] = e[k][order].cpu()
    assert sorted(set(ex["game_id"].tolist())) == list(range(32, 64))
    for k in want:
        assert torch.equal(ex[k], want[k]), k

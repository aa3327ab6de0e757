"""`MCTS` / `Node` of the reference (src/yin_yang/ai/mcts.py:28-567) on top of the HIP engine.

Same constructor and methods as the reference (`search`, `select_action`, `reuse_tree`), same
return values: `search` gives (pi float64[A], root) with `root.visits == num_simulations`, and --
in the literal `aliased` board semantics -- the caller's board object is mutated by the search
exactly as the reference's is.  The tree itself lives in HBM inside
engine.BatchedMCTS; `Node` here is a read-only view of the root and its children.

The rules are fused into the kernels, so `game` must be a Yin-Yang game (anything exposing
getBoardSize()/getActionSize(); fakes with other rules are not supported).  `neural_net` is either
an object with `predict(board) -> (policy[A], value)` (called once per evaluation on the host, like
the reference does) or a callable batched evaluator `planes[G,5,R,C] -> (policy[G,A], value[G])`
on the device (network.BatchedEvaluator), which is the fast path.
"""
import math

import numpy as np
import torch

from . import engine
from .game import YinYangLogic


class Node:
    """Host view of a tree node with the reference's constructor, fields and methods (mcts.py:28-215).  The searched tree
    itself lives in HBM; `MCTS.search` returns a view of the root and its children.  A Node built by hand (as the
    reference's tests do) works on the host: `expand` asks the game for the legal moves / outcome, `select_child`,
    `update` and the distribution helpers follow the reference's arithmetic."""

    def __init__(self, game=None, parent=None, action=None, prior=0.0, visits=0, value_sum=0.0):
        self.game = game
        self.parent, self.action = parent, action
        self.children = {}
        self.visits, self.value_sum, self.prior = visits, value_sum, prior
        self.board = self.player = self.valid_moves = None
        self.is_terminal, self.terminal_value = False, None

    def expand(self, board, player, policy_probs):
        """mcts.py:50-91: terminal -> store the value, no children; else one child per legal move, ascending action, with
        the RAW policy entry as prior (uniform over the legal moves when no policy is given)."""
        self.board, self.player = board, player
        result = self.game.getGameEnded(board, player)
        if result != 0:
            self.is_terminal, self.terminal_value = True, result
            return
        self.valid_moves = self.game.getValidMoves(board, player)
        legal = np.where(self.valid_moves == 1)[0]
        for a in legal:
            prior = policy_probs[a] if policy_probs is not None else 1.0 / len(legal)
            self.children[a] = Node(game=self.game, parent=self, action=a, prior=prior)

    def is_expanded(self):
        return len(self.children) > 0 or self.is_terminal

    def get_value(self):
        return 0.0 if self.visits == 0 else self.value_sum / self.visits

    def get_visit_count(self):
        return self.visits

    def select_child(self, c_puct=1.0):
        """PUCT choice over this node's children (mcts.py:97-145), evaluated on the host view with the same scalar types
        the reference holds (np.float32 priors / value sums, python ints), hence the same float32 rounding: S = sum of the
        CHILDREN's visits, q = W/N (0 when unvisited), u = c_puct * P * sqrt(S) / (1 + N), strict > so the lowest
        action wins ties.  The device kernel does the same per level (csrc/yy_engine.hip, do_select)."""
        total = sum(ch.visits for ch in self.children.values())
        root_s = math.sqrt(total)
        best, best_a, best_child = -float("inf"), -1, None
        for a in sorted(self.children):
            ch = self.children[a]
            q = ch.value_sum / ch.visits if ch.visits > 0 else 0.0
            score = q + c_puct * ch.prior * root_s / (1 + ch.visits)
            if score > best:
                best, best_a, best_child = score, a, ch
        return best_a, best_child

    def update(self, value):
        """mcts.py:147-156: one visit, value added from this node's point of view (the caller flips the sign per ply)."""
        self.visits += 1
        self.value_sum += value

    def get_children_distribution(self, temperature=1.0, action_size=None):
        """mcts.py:183-215 on this node's children."""
        return children_distribution(self.get_children_visit_counts(action_size), temperature)

    def get_children_visit_counts(self, action_size=None):
        n = action_size if action_size is not None else (max(self.children) + 1 if self.children else 0)
        counts = np.zeros(n)
        for a, ch in self.children.items():
            counts[a] = ch.visits
        return counts


def children_distribution(counts, temperature=1.0):
    """Node.get_children_distribution (mcts.py:183-215) on a float64 count vector."""
    counts = np.asarray(counts, dtype=np.float64)
    n = counts.size
    if temperature == 0:
        best = np.where(counts == np.max(counts))[0]
        probs = np.zeros(n)
        probs[best] = 1.0 / len(best)
        return probs
    if temperature != 1.0:
        counts = np.power(counts, 1.0 / temperature)
    s = np.sum(counts)
    return counts / s if s > 0 else np.ones(n) / n


class _HostPredictEvaluator:
    """Adapter: object with predict(board) -> batched evaluator, evaluating only the rows the
    engine asked for, in game order (the reference's call order for G == 1)."""

    def __init__(self, net, mcts_ctx, R, C):
        self.net, self.ctx, self.R, self.C = net, mcts_ctx, R, C
        self.first = True

    def __call__(self, planes):
        G, A = planes.shape[0], self.R * self.C
        need = np.ones(G, bool) if self.first else self.ctx.needs_eval.cpu().numpy().astype(bool)
        self.first = False
        pol = np.zeros((G, A), np.float32)
        val = np.zeros(G, np.float32)
        if need.any():
            p = planes.cpu().numpy()
            for g in np.flatnonzero(need):
                lb = YinYangLogic(self.R, self.C)
                lb.board = (p[g, 1] - p[g, 2]).astype(np.int8)
                pg, vg = self.net.predict(lb)
                pol[g], val[g] = np.asarray(pg, np.float32), np.float32(vg)
        return torch.from_numpy(pol).to(planes.device), torch.from_numpy(val).to(planes.device)


class MCTS:
    def __init__(self, game, neural_net, num_simulations=800, cpuct=1.0, temperature=1.0, num_threads=1,
                 dirichlet_noise=True, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, verbose=1,
                 board_semantics="aliased", device=None, evaluation_reuse=False):
        """board_semantics: "aliased" = literal reference (the search mutates the caller's board);
        "copied" = every node owns its board (what the reference's own tests assume).
        evaluation_reuse (not in the reference; default off = its evaluator call sequence): a position is evaluated once per
        search (pass values with copied boards + the evaluation cache, include/yy_engine.h YY_FLAG_REUSE_*); same results, fewer
        evaluator rows; needs a deterministic network whose row results do not depend on the rest of the batch.
        num_threads is accepted for signature compatibility; simulations of one search are always
        sequential (the reference's thread pool is an unsynchronised race)."""
        assert board_semantics in ("aliased", "copied")
        self.game, self.neural_net = game, neural_net
        self.num_simulations, self.cpuct, self.temperature = num_simulations, cpuct, temperature
        self.num_threads = max(1, num_threads)
        self.use_dirichlet, self.dirichlet_alpha, self.dirichlet_epsilon = dirichlet_noise, dirichlet_alpha, dirichlet_epsilon
        self.board_semantics = board_semantics
        self.evaluation_reuse = bool(evaluation_reuse)
        self.R, self.C = game.getBoardSize()
        self.A = game.getActionSize()
        self.rowcol = bool(getattr(game, "rowcol_rule", False))
        self.device = device
        self._ctx = {}
        self.use_graph = True          # hipGraph replay of the simulation step for device-resident evaluators

    def _context(self, G):
        ctx = self._ctx.get(G)
        if ctx is None or ctx.max_sims < self.num_simulations:
            if ctx is not None:
                ctx.close()
            ctx = engine.BatchedMCTS(G, self.R, self.C, self.num_simulations, cpuct=self.cpuct,
                                     aliased=(self.board_semantics == "aliased"), rowcol=self.rowcol,
                                     device=self.device,
                                     reuse_pass_value=self.evaluation_reuse and self.board_semantics == "copied",
                                     reuse_transpositions=self.evaluation_reuse)
            self._ctx[G] = ctx
        return ctx

    def _capturable(self, ev):
        """True for evaluators known to do device work only: the CUDA nn.Module's predict_batch and BatchedEvaluator."""
        from .network import BatchedEvaluator
        net = self.neural_net
        return isinstance(net, BatchedEvaluator) or (isinstance(net, torch.nn.Module) and ev == getattr(net, "predict_batch", None))

    def _evaluator(self, ctx):
        net = self.neural_net
        if isinstance(net, torch.nn.Module) and hasattr(net, "predict_batch") and next(net.parameters()).is_cuda:
            return net.predict_batch                       # one forward for all G leaves on the device
        if callable(net) and not hasattr(net, "predict"):
            return net                                     # network.BatchedEvaluator or any batched callable
        return _HostPredictEvaluator(net, ctx, self.R, self.C)

    # ---- batched entry point
    def search_batch(self, boards, players, noise=None, active=None):
        """boards int8 [G,R,C], players int8 [G] (device tensors) -> (pi float64 [G,A], ctx).
        noise: None or float64 [G,A] Dirichlet draws scattered to the legal actions."""
        ctx = self._context(boards.shape[0])
        ev = self._evaluator(ctx)
        if self.use_graph and self._capturable(ev) and self.num_simulations > 8:
            # device-resident evaluator: replay [forward + yy_mcts_step] from a hipGraph (one host call per simulation
            # instead of one per kernel); same launches, same results as the eager loop
            from .self_play import LockstepSearch
            cached = getattr(ctx, "_lockstep", None)       # lives and dies with the context whose buffers it captured;
            if cached is None or cached[0] is not self.neural_net:   # holds the network it captured alive
                cached = ctx._lockstep = (self.neural_net, LockstepSearch(ctx, ev, use_graph=True))
            cached[1].run(boards, players, self.num_simulations, noise=noise, eps=self.dirichlet_epsilon, active=active)
        else:
            ctx.search(boards, players, ev, self.num_simulations, noise=noise, eps=self.dirichlet_epsilon, active=active)
        if self.temperature in (0, 1.0):
            pi = ctx.root_policy(temperature_zero=(self.temperature == 0))
        else:
            c = ctx.root_counts().cpu().numpy()
            pi = torch.from_numpy(np.stack([children_distribution(r, self.temperature) for r in c])).to(boards.device)
        return pi, ctx

    # ---- reference API (mcts.py:275-343)
    def search(self, board, player, add_exploration_noise=False):
        dev = torch.device("cuda", torch.cuda.current_device()) if self.device is None else torch.device(self.device)
        b = torch.from_numpy(np.ascontiguousarray(board.board, dtype=np.int8)[None]).to(dev)
        pl = torch.tensor([1 if player == 1 else -1], dtype=torch.int8, device=dev)
        noise = None
        valid = engine.valid_mask(b, pl, self.rowcol)[0].cpu().numpy()
        if add_exploration_noise and self.use_dirichlet:
            idx = np.flatnonzero(valid)
            if len(idx) > 0:
                draw = np.random.dirichlet([self.dirichlet_alpha] * len(idx))   # same global stream, mcts.py:305
                nz = np.zeros((1, self.A))
                nz[0, idx] = draw
                noise = torch.from_numpy(nz).to(dev)
        ended = float(engine.game_ended(b, pl, self.rowcol)[0])
        terminal = 0 if ended == 0.0 else (ended if ended == 0.0001 else int(ended))
        pi, ctx = self.search_batch(b, pl, noise=noise)
        counts, cw, cp = ctx.root_counts(with_children=True)
        visits, wsum = ctx.root_stats()
        if self.board_semantics == "aliased":
            board.board[...] = ctx.boards()[0].cpu().numpy()     # the reference mutates the caller's board
        ctx.status()
        counts, cw, cp = counts[0].cpu().numpy(), cw[0].cpu().numpy(), cp[0].cpu().numpy()
        root = Node(game=self.game, visits=int(visits[0]), value_sum=float(wsum[0]))
        root.board, root.player = board, player
        root.is_terminal, root.terminal_value = (terminal != 0), (terminal if terminal != 0 else None)
        root.valid_moves = valid.astype(np.float64)
        for a in (np.flatnonzero(valid) if terminal == 0 else []):
            root.children[int(a)] = Node(game=self.game, action=int(a), prior=np.float32(cp[a]), visits=int(counts[a]),
                                         value_sum=np.float32(cw[a]), parent=root)
        return pi[0].cpu().numpy(), root

    # ---- mcts.py:427-479
    def select_action(self, board, player, temperature=None, valid_moves=None, add_exploration_noise=False):
        temp = temperature if temperature is not None else self.temperature
        action_probs, _ = self.search(board, player, add_exploration_noise)
        if valid_moves is not None:
            if len(valid_moves) != len(action_probs):
                valid_moves = np.ones_like(action_probs)
            masked = action_probs * valid_moves
            s = np.sum(masked)
            if s <= 0:
                return np.argmax(action_probs)
            action_probs = masked / s
        if temp == 0:
            return np.argmax(action_probs)
        return np.random.choice(np.arange(len(action_probs)), p=action_probs)

    # ---- mcts.py:481-505; the reference never feeds the result back into search (dead code there)
    def reuse_tree(self, old_root, board, player, action_taken):
        if old_root is not None and action_taken in old_root.children:
            new_root = old_root.children[action_taken]
            new_root.parent = None
            return new_root
        node = Node()
        node.board, node.player = board, player
        return node

    def close(self):
        for ctx in self._ctx.values():
            ctx.close()
        self._ctx = {}

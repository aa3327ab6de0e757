"""Arena / evaluation and the iteration orchestrator.

Mirrors src/yin_yang/ai/alphazero.py (AlphaZero :20-270, AlphaZeroPlayer :272-365),
src/yin_yang/yin_yang_players.py (RandomPlayer :5-42) and the `evaluate` mode of
train_alphazero.py:124-243.  Matches run batched on the same HIP engine: every game of a match is a
slot of one lockstep batch, the two networks share the batch through `DualEvaluator` (each row is
routed to the network whose turn it is), moves are arg-max of the visit counts (temperature 0) with
the REAL side to move as root player and copied boards.

Scoring note.  The reference scores a finished game with `getGameEnded(board, player)` for the player
who would move NEXT and reads +1 as "first player won" (alphazero.py:206-218), which mis-attributes
the winner whenever the loser is to move; here the winner is the colour with more stones
(yin_yang_game.py:98-107), attributed to whichever network played that colour.
"""
import os
import shutil

import numpy as np
import torch

from . import engine
from .mcts import MCTS
from .network import BatchedEvaluator, YinYangNeuralNetwork
from .self_play import LockstepSearch, generate_self_play_data
from .training import run_training_pipeline


class DualEvaluator:
    """Gives each row of the leaf batch the output of evaluator A or B (fixed during one search).

    Two modes.  `dense` (small matches): both networks run on ALL rows and a device-side row mask picks the output, so
    the step has no host-dependent shapes and the simulation loop can be replayed from a hipGraph (a 40-game arena is
    launch-bound, not compute-bound).  Routed (large matches): each network sees only its own rows (index_select /
    index_copy), eager launches."""

    def __init__(self, ev_a, ev_b, G, A, device, dense=False):
        self.ev_a, self.ev_b, self.dense = ev_a, ev_b, dense
        self.idx_a = self.idx_b = None
        self.side = None
        self.sel = torch.ones(G, dtype=torch.bool, device=device)
        self.policy = torch.zeros((G, A), dtype=torch.float32, device=device)
        self.value = torch.zeros(G, dtype=torch.float32, device=device)
        # dense mode with two compacting evaluators: the step's needs_eval flags go to both, so a step in which no game needs an
        # evaluation (terminal revisits, reused evaluations) launches two towers that exit at once
        self.supports_compaction = dense and all(getattr(e, "supports_compaction", False) for e in (ev_a, ev_b))
        self.row_independent = all(getattr(e, "row_independent", False) for e in (ev_a, ev_b))

    def assign(self, a_rows):
        self.sel.copy_(a_rows)
        if not self.dense:
            self.idx_a = a_rows.nonzero(as_tuple=True)[0]
            self.idx_b = (~a_rows).nonzero(as_tuple=True)[0]

    def __call__(self, planes, needs_eval=None):
        if self.dense:
            kw = {} if needs_eval is None else dict(needs_eval=needs_eval)
            # the two forwards are independent and each fills only a few CUs: fork B onto a side stream (capturable),
            # join before the row select
            cur = torch.cuda.current_stream(planes.device)
            if self.side is None:
                self.side = torch.cuda.Stream(device=planes.device)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                pb, vb = self.ev_b(planes, **kw)
            pa, va = self.ev_a(planes, **kw)
            cur.wait_stream(self.side)
            torch.where(self.sel[:, None], pa, pb, out=self.policy)
            torch.where(self.sel, va, vb, out=self.value)
            return self.policy, self.value
        for idx, ev in ((self.idx_a, self.ev_a), (self.idx_b, self.ev_b)):
            if idx.numel():
                p, v = ev(planes.index_select(0, idx).contiguous())
                self.policy.index_copy_(0, idx, p)
                self.value.index_copy_(0, idx, v)
        return self.policy, self.value


def _uniform_evaluator(A):
    def ev(planes):
        g = planes.shape[0]
        return torch.full((g, A), 1.0 / A, device=planes.device), torch.zeros(g, device=planes.device)
    return ev


def _lowest_argmax(pi):
    """np.argmax(pi) per row: the LOWEST index among the maxima (select_action with temperature 0, mcts.py:471-472)."""
    A = pi.shape[1]
    best = pi == pi.max(1, keepdim=True).values
    idx = torch.arange(A, device=pi.device)[None, :].expand_as(pi)
    return torch.where(best, idx, torch.full_like(idx, A)).min(1).values.to(torch.int32)


class Arena:
    """num_games games between two players on one GPU.  A player is a batched evaluator
    (planes -> policy, value) searched with `num_simulations`, or the string "random" (uniformly random
    legal moves, RandomPlayer).  Moves of a searching player are np.argmax of the search's pi (lowest index among the most
    visited moves), exactly the reference's select_action(temperature=0) (alphazero.py:192-195, mcts.py:471-472): a match
    between two fixed networks is deterministic, as the reference's is."""

    def __init__(self, game, player_a, player_b, num_simulations=800, cpuct=1.0, device=None, seed=0,
                 reference_scoring=False, literal=False, evaluation_reuse=True):
        """evaluation_reuse: inside one search a position is evaluated once (pass values, other move orders) when both players
        are row-independent compacting evaluators (the float32-accurate BatchedEvaluator); the moves are the same.
        reference_scoring=True reproduces the reference's attribution literally (alphazero.py:206-218): the value of
        getGameEnded for the player who would move NEXT is read as "+1 = first player won, -1 = second player won".
        literal=True reproduces the whole reference match loop (alphazero.py:171-220): the search runs on the game's own
        board object and mutates it (aliased boards), there is no pass handling -- a side without a move still
        searches, np.argmax of the uniform pi gives action 0, the illegal placement is ignored and the player flips (Q3) --
        and the scoring is the reference's."""
        self.game = game
        self.literal = bool(literal)
        self.evaluation_reuse = bool(evaluation_reuse)
        self.reference_scoring = bool(reference_scoring) or self.literal
        self.R, self.C = game.getBoardSize()
        self.A = self.R * self.C
        self.pa, self.pb, self.sims, self.cpuct = player_a, player_b, num_simulations, cpuct
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.gen = torch.Generator(device=self.device)      # RandomPlayer moves only
        self.gen.manual_seed(seed)
        self.rowcol = bool(getattr(game, "rowcol_rule", False))
        self.transcript = None

    def play(self, num_games, record=False):
        """Returns dict(a_wins, b_wins, draws).  Game i: A plays black iff i is even (alphazero.py:177-186).
        record=True keeps (actions int32 [G,T], players int8 [G,T], n_moves [G], results float64 [G]) in self.transcript:
        every call of the match loop's getNextState and the value of getGameEnded that ended each game."""
        G, dev = int(num_games), self.device
        boards = torch.zeros((G, self.R, self.C), dtype=torch.int8, device=dev)
        players = torch.ones(G, dtype=torch.int8, device=dev)
        a_is_black = (torch.arange(G, device=dev) % 2 == 0)
        alive = torch.ones(G, dtype=torch.bool, device=dev)
        ev_a = _uniform_evaluator(self.A) if self.pa == "random" else self.pa
        ev_b = _uniform_evaluator(self.A) if self.pb == "random" else self.pb
        dense = G <= 1024
        dual = DualEvaluator(ev_a, ev_b, G, self.A, dev, dense=dense)
        # a search uses ONE network per game (assigned per move), so inside a search a position's evaluation can be reused
        # (pass values, other move orders); never across searches: the two networks alternate
        reuse = self.evaluation_reuse and (not self.literal) and dense and dual.supports_compaction and dual.row_independent
        ctx = engine.BatchedMCTS(G, self.R, self.C, max(1, self.sims), cpuct=self.cpuct, rowcol=self.rowcol, device=dev,
                                 aliased=self.literal, reuse_pass_value=reuse, reuse_transpositions=reuse)
        try:
            search = LockstepSearch(ctx, dual, use_graph=dense)     # routed mode: the row sets change every move
            result = torch.zeros(G, dtype=torch.int8, device=dev)   # +1 black won, -1 white won, 2 draw
            ended_at = torch.zeros(G, dtype=torch.float64, device=dev)
            T = 4 * self.A + 8
            rec_a = torch.full((G, T), -1, dtype=torch.int32, device=dev)
            rec_p = torch.zeros((G, T), dtype=torch.int8, device=dev)
            n_moves = torch.zeros(G, dtype=torch.int64, device=dev)
            ar = torch.arange(G, device=dev)
            for _ in range(T):
                if not bool(alive.any()):
                    break
                if self.literal:
                    movers = alive.clone()                            # no pass handling: the side to move always "moves"
                    mask = engine.valid_mask(boards, players, self.rowcol)
                else:
                    mask = engine.valid_mask(boards, players, self.rowcol)
                    has = mask.bool().any(1)
                    # a side without a move passes; when neither side can move the game is over (scored below)
                    players = torch.where(alive & ~has, -players, players).contiguous()
                    if record:     # the reference's loop makes that pass as a move: action 0, ignored, player flips (Q3)
                        passed = alive & ~has
                        slot = n_moves.clamp_max(T - 1)
                        rec_a[ar, slot] = torch.where(passed, torch.zeros_like(rec_a[ar, slot]), rec_a[ar, slot])
                        rec_p[ar, slot] = torch.where(passed, -players, rec_p[ar, slot])
                        n_moves += passed.to(torch.int64)
                    mask = engine.valid_mask(boards, players, self.rowcol)
                    has = mask.bool().any(1)
                    movers = alive & has
                a_to_move = (players == 1) == a_is_black
                rand_rows = torch.zeros(G, dtype=torch.bool, device=dev)
                if self.pa == "random":
                    rand_rows |= a_to_move
                if self.pb == "random":
                    rand_rows |= ~a_to_move
                action = torch.full((G,), -1, dtype=torch.int32, device=dev)
                srch = movers & ~rand_rows
                if bool(srch.any()):
                    dual.assign(a_to_move)
                    search.run(boards, players, self.sims, noise=None, active=srch.to(torch.uint8))
                    pi = ctx.root_policy()                               # search() returns the T == 1 distribution (:329)
                    action = torch.where(srch, _lowest_argmax(pi), action)   # select_action(temperature=0), :471-472
                    if self.literal:                                     # the search mutated the game's board (Q2)
                        boards = torch.where(srch[:, None, None], ctx.boards(), boards).contiguous()
                rnd = movers & rand_rows
                if bool(rnd.any()):
                    m = torch.where(rnd[:, None], mask.float(), torch.full((G, self.A), 1.0, device=dev))
                    pick = torch.multinomial(m, 1, generator=self.gen).reshape(-1).to(torch.int32)
                    action = torch.where(rnd, pick, action)
                if record:
                    slot = n_moves.clamp_max(T - 1)
                    rec_a[ar, slot] = torch.where(movers, action, rec_a[ar, slot])
                    rec_p[ar, slot] = torch.where(movers, players, rec_p[ar, slot])
                    n_moves += movers.to(torch.int64)
                old = players.clone()
                engine.step_(boards, players, action.contiguous(), self.rowcol)
                players = torch.where(movers, players, old).contiguous()
                ended, counts = engine.game_ended(boards, players, self.rowcol, with_counts=True)
                over = alive & (ended != 0)
                diff = (counts[:, 0] - counts[:, 1])
                res = torch.where(diff > 0, 1, torch.where(diff < 0, -1, 2)).to(torch.int8)
                if self.reference_scoring:
                    res = torch.where(ended == 1, 1, torch.where(ended == -1, -1, 2)).to(torch.int8)
                result = torch.where(over, res, result)
                ended_at = torch.where(over, ended, ended_at)
                alive &= ~over
            ctx.status()
        finally:
            ctx.close()
        if record:
            self.transcript = dict(actions=rec_a.cpu().numpy(), players=rec_p.cpu().numpy(), n_moves=n_moves.cpu().numpy(),
                                   results=ended_at.cpu().numpy())
        black_won, white_won = (result == 1), (result == -1)
        a_wins = int(((black_won & a_is_black) | (white_won & ~a_is_black)).sum())
        b_wins = int(((black_won & ~a_is_black) | (white_won & a_is_black)).sum())
        return dict(a_wins=a_wins, b_wins=b_wins, draws=G - a_wins - b_wins, games=G)


class RandomPlayer:
    """yin_yang_players.py:5-42 (without the print)."""

    def __init__(self, game):
        self.game = game

    def play(self, board, player):
        idx = np.where(self.game.getValidMoves(board, player) == 1)[0]
        return -1 if len(idx) == 0 else int(np.random.choice(idx))


class AlphaZeroPlayer:
    """alphazero.py:272-365: network + MCTS behind `play(board, player) -> action` (-1 = no move)."""

    def __init__(self, game, model_path, num_simulations=800, num_threads=1, device=None):
        self.game = game
        net = YinYangNeuralNetwork(game)
        if os.path.exists(model_path):
            net.load_model(model_path)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.neural_net = net.to(dev).eval()
        self.mcts = MCTS(game, self.neural_net, num_simulations=num_simulations, num_threads=num_threads,
                         board_semantics="copied", device=dev)
        self.root = None

    def reset(self):
        self.root = None

    def play(self, board, player):
        valid = self.game.getValidMoves(board, player)
        if np.sum(valid) == 0:
            return -1
        return int(self.mcts.select_action(board, player, temperature=0, valid_moves=valid))

    def notify(self, board, action):
        """alphazero.py:354-364: keep the subtree under the opponent's move (the reference never searches from it either)."""
        if self.root is not None:
            self.root = self.mcts.reuse_tree(self.root, board, -1, action)


def _load_evaluator(game, path, device, nn_mode, num_channels, num_res_blocks):
    net = YinYangNeuralNetwork(game, num_channels, num_res_blocks)
    net.load_model(path)
    return BatchedEvaluator(net.to(device).eval(), nn_mode)


class AlphaZero:
    """Iteration loop (alphazero.py:20-270): self_play(best) -> train -> evaluate(current vs best) ->
    promote at win ratio >= update_threshold; same file names (current_model / best_model /
    checkpoint_<n>.pth.tar)."""

    def __init__(self, game, model_dir="models", data_dir="data", num_iterations=100, num_episodes=100,
                 num_simulations=800, num_epochs=10, temperature_threshold=10, update_threshold=0.6, num_workers=1,
                 mcts_threads=1, arena_games=40, nn_mode="auto", num_channels=128, num_res_blocks=10,
                 concurrent_games=4096, device=None, lr=0.001, batch_size=64):
        self.game, self.model_dir, self.data_dir = game, model_dir, data_dir
        self.num_iterations, self.num_episodes, self.num_simulations = num_iterations, num_episodes, num_simulations
        self.num_epochs, self.temperature_threshold, self.update_threshold = num_epochs, temperature_threshold, update_threshold
        self.num_workers, self.mcts_threads, self.arena_games = num_workers, mcts_threads, arena_games
        self.nn_mode, self.num_channels, self.num_res_blocks = nn_mode, num_channels, num_res_blocks
        self.concurrent_games = concurrent_games
        self.lr, self.batch_size = lr, batch_size
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        for d in (model_dir, data_dir):
            os.makedirs(d, exist_ok=True)
        self.current_model_path = os.path.join(model_dir, "current_model.pth.tar")
        self.best_model_path = os.path.join(model_dir, "best_model.pth.tar")
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            if not os.path.exists(self.current_model_path):
                YinYangNeuralNetwork(game, num_channels, num_res_blocks).save_model(self.current_model_path)
            if not os.path.exists(self.best_model_path):
                shutil.copy(self.current_model_path, self.best_model_path)
        if multi:
            dist.barrier()
        self.history = []

    def self_play(self, model_path):
        return generate_self_play_data(self.game, model_path, self.data_dir, num_games=self.num_episodes,
                                       num_workers=self.num_workers, num_simulations=self.num_simulations,
                                       temperature_threshold=self.temperature_threshold, nn_mode=self.nn_mode,
                                       num_channels=self.num_channels, num_res_blocks=self.num_res_blocks,
                                       concurrent_games=self.concurrent_games, seed=len(self.history))

    def train(self):
        # the reference passes num_iterations=1 and ignores --epochs/--batch-size/--lr (alphazero.py:120-127,
        # train_alphazero.py:42-44); they are honoured here
        new_path = run_training_pipeline(self.game, self.model_dir, self.data_dir, num_iterations=1, sample_size=10000,
                                         checkpoint_interval=1, epochs_per_iteration=self.num_epochs, device=self.device,
                                         num_channels=self.num_channels, num_res_blocks=self.num_res_blocks,
                                         lr=self.lr, batch_size=self.batch_size)
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            shutil.copy(new_path, self.current_model_path)
        if multi:
            dist.barrier()
        return self.current_model_path

    def evaluate(self, current_model_path, best_model_path, num_games=None):
        """alphazero.py:136-226.  Multi-rank: the match is sharded (2 games per colour pair at a time) and the win
        counts are summed with one all-reduce, so every rank takes the same promote decision."""
        import torch.distributed as dist
        n = self.arena_games if num_games is None else num_games
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
        pairs = n // 2                                   # keep colours balanced inside every shard
        mine = 2 * (pairs // world + (1 if rank < pairs % world else 0)) + (n % 2 if rank == 0 else 0)
        cur = _load_evaluator(self.game, current_model_path, self.device, self.nn_mode, self.num_channels, self.num_res_blocks)
        best = _load_evaluator(self.game, best_model_path, self.device, self.nn_mode, self.num_channels, self.num_res_blocks)
        res = (Arena(self.game, cur, best, self.num_simulations, device=self.device, seed=len(self.history) * 131 + rank).play(mine)
               if mine > 0 else dict(a_wins=0, b_wins=0, draws=0, games=0))
        if multi:
            t = torch.tensor([res["a_wins"], res["b_wins"], res["draws"], res["games"]], dtype=torch.int64, device=self.device)
            dist.all_reduce(t)
            res = dict(a_wins=int(t[0]), b_wins=int(t[1]), draws=int(t[2]), games=int(t[3]))
        self.last_arena = res
        return res["a_wins"] / max(1, res["games"])

    def update_best_model(self, win_ratio):
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        promote = win_ratio >= self.update_threshold
        if promote and (not multi or dist.get_rank() == 0):
            shutil.copy(self.current_model_path, self.best_model_path)
        if multi:
            dist.barrier()
        return promote

    def run(self):
        import time
        for it in range(self.num_iterations):
            t0 = time.perf_counter()
            data_file = self.self_play(self.best_model_path)
            t1 = time.perf_counter()
            self.train()
            t2 = time.perf_counter()
            ratio = self.evaluate(self.current_model_path, self.best_model_path)
            t3 = time.perf_counter()
            promoted = self.update_best_model(ratio)
            import torch.distributed as dist
            sums = [run_training_pipeline.last["param_checksum"]]
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                every = [None] * dist.get_world_size()
                dist.all_gather_object(every, sums[0])      # every rank's fingerprint of its weights after the DDP training
                sums = every
            self.history.append(dict(iteration=it + 1, data_file=data_file, self_play_s=t1 - t0, train_s=t2 - t1,
                                     arena_s=t3 - t2, win_ratio=ratio, promoted=promoted, arena=self.last_arena,
                                     losses=run_training_pipeline.last["metrics"]["total_loss"],
                                     examples=run_training_pipeline.last["examples"], param_checksums=sums))
        return self.history


def evaluate_vs_random(game, model_path, num_games=10, num_simulations=800, nn_mode="auto", device=None,
                       num_channels=128, num_res_blocks=10):
    """`--mode evaluate` (train_alphazero.py:124-243): the model against RandomPlayer, alternating colours."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ev = _load_evaluator(game, model_path, dev, nn_mode, num_channels, num_res_blocks)
    res = Arena(game, ev, "random", num_simulations, device=dev).play(num_games)
    return dict(alphazero_wins=res["a_wins"], random_wins=res["b_wins"], draws=res["draws"],
                win_rate=res["a_wins"] / num_games)

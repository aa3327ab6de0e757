"""Self-play: the batched lockstep engine (throughput path) and the reference's self-play API.

Reference: src/yin_yang/ai/self_play.py -- SelfPlayWorker.play_game :72-192, generate_games
:194-216, SelfPlayManager :218-335, generate_self_play_data :337-387.

`SelfPlayEngine` is the MI355X design: G games advance in lockstep on one GPU; per move every
game runs `num_simulations` simulations, each simulation being ONE fused HIP kernel
(expand+backup of the previous leaves, PUCT select + rules + plane encode of the next) and ONE
batched CNN forward over all G leaves, replayed from a hipGraph.  Finished games are replaced in
their slot so the batch stays full.  Episodes shard across ranks (one process per GPU) and the
(state, pi, z) examples are collected with one all-gather at the end (`gather_examples`).

`SelfPlayWorker` / `SelfPlayManager` / `generate_self_play_data` keep the reference's names and
signatures.  `reference_quirks=True` reproduces Q4/Q5 of SURVEY.md (always search and mask as
player +1, never alternate the value labels); `board_semantics="aliased"` reproduces Q2.
"""
import os
import time

import numpy as np
import torch

from . import engine
from .game import YinYangLogic
from .mcts import MCTS
from .network import BatchedEvaluator, YinYangNeuralNetwork

DRAW = 1e-4


# =============================================================================== graph-replayed search
class LockstepSearch:
    """MCTS.search for G games with the simulation loop replayed from a hipGraph.

    One graph = [evaluator forward on ctx.planes] + [yy_mcts_step]; everything in it is enqueued on
    the capture stream (the C ABI takes the stream as an argument), so a replay costs one host call
    instead of ~60 kernel launches."""

    def __init__(self, ctx, evaluator, use_graph=True, eager_sims=3, unroll=None):
        self.ctx, self.evaluator, self.use_graph = ctx, evaluator, use_graph
        self.eager_sims = eager_sims
        # simulations per replayed graph: besides the one-step graph a second one holds `unroll` consecutive steps
        # (fewer host calls, and no graph-to-graph launch gap between the steps inside it); 1 = one-step graphs only
        self.unroll = max(1, int(os.environ.get("YY_GRAPH_UNROLL", "8") if unroll is None else unroll))
        self.graphs = {}       # evaluated rows -> captured step
        self.timer = None      # optional object with start()/stop() bracketing every tree-kernel launch (bench.py)
        self._policy = self._value = None
        self._book_version = getattr(ctx, "book_version", 0)
        self._graph_evaluator = evaluator

    @property
    def graph(self):
        return self.graphs.get(self.ctx.G)

    @graph.setter
    def graph(self, g):
        if g is None:
            self.graphs.clear()
        else:
            self.graphs[self.ctx.G] = g

    def _evaluate(self, rows, compact=False):
        """Evaluator on the first `rows` leaf rows (all searching games must sit there); returns full-height buffers.
        compact: pass the step's needs_eval flags to an evaluator that can skip the rows whose leaf needs no evaluation
        (terminal revisits, mcts.py:365-366; finished or idle slots)."""
        ctx = self.ctx
        kw = {}
        if compact and getattr(self.evaluator, "supports_compaction", False):
            kw["needs_eval"] = ctx.needs_eval if rows >= ctx.G else ctx.needs_eval[:rows]
        if getattr(self.evaluator, "supports_static", False):
            kw["static"] = id(self)                # private buffers: the results are consumed by the tree kernel before this search's next call
        if rows >= ctx.G:
            return self.evaluator(ctx.planes, **kw)
        policy, value = self.evaluator(ctx.planes[:rows], **kw)
        if self._policy is None:
            self._policy = torch.zeros((ctx.G, policy.shape[1]), dtype=torch.float32, device=ctx.device)
            self._value = torch.zeros(ctx.G, dtype=torch.float32, device=ctx.device)
        self._policy[:rows].copy_(policy)
        self._value[:rows].copy_(value)
        return self._policy, self._value

    def _sim_step(self, rows):
        policy, value = self._evaluate(rows, compact=True)
        if self.timer is not None:
            self.timer.start()
        self.ctx.step(policy, value)
        if self.timer is not None:
            self.timer.stop()

    def run(self, boards, root_players, num_sims, noise=None, eps=0.25, active=None, rows=None):
        """rows: evaluate only leaf rows [0, rows) -- the caller guarantees every active game has an index below it
        (SelfPlayEngine packs the live games to the front when a batch drains).  One graph per distinct `rows`."""
        ctx = self.ctx
        ctx.bind_evaluator(self.evaluator)                     # kept evaluations / the book belong to ONE network (may drop the book)
        if getattr(ctx, "book_version", 0) != self._book_version or self._graph_evaluator is not self.evaluator:
            # a captured step has its kernel arguments frozen (the book's tables) and calls the evaluator it was captured with
            self.graphs.clear()
            self._book_version = getattr(ctx, "book_version", 0)
            self._graph_evaluator = self.evaluator
        rows = ctx.G if rows is None else min(int(rows), ctx.G)
        ctx.begin(boards, root_players, active)
        policy, _ = self._evaluate(rows)                       # mcts.py:295, value discarded
        ctx.expand_root(policy, noise, eps)
        ctx.select()
        done = 0
        n_fused = num_sims - 1
        gkey = rows if not hasattr(self.evaluator, "form_key") else (rows, self.evaluator.form_key(id(self)))
        graph = self.graphs.get(gkey)
        if self.use_graph and graph is None and n_fused > self.eager_sims:
            for _ in range(self.eager_sims):                   # warm-up (MIOpen algo search etc.) = real sims
                self._sim_step(rows)
                done += 1
            torch.cuda.synchronize(ctx.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._sim_step(rows)
            self.graphs[gkey] = graph
            if self.unroll > 1 and n_fused - done >= 2 * self.unroll:
                many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(many):
                    for _ in range(self.unroll):
                        self._sim_step(rows)
                self.graphs[(gkey, self.unroll)] = many
        many = self.graphs.get((gkey, self.unroll)) if self.use_graph else None
        if not self.use_graph:
            graph = None
        while done < n_fused:
            if many is not None and n_fused - done >= self.unroll:
                many.replay()
                done += self.unroll
                continue
            if graph is not None:
                graph.replay()
            else:
                self._sim_step(rows)
            done += 1
        policy, value = self._evaluate(rows, compact=True)     # last simulation: no further select
        ctx.expand_backup(policy, value)


# =============================================================================== batched engine
class SelfPlayEngine:
    def __init__(self, game, evaluator, num_simulations=800, concurrent_games=4096, cpuct=1.0,
                 dirichlet_alpha=0.3, dirichlet_epsilon=0.25, temperature_threshold=10,
                 board_semantics="copied", reference_quirks=False, use_graph=True, seed=0,
                 device=None, first_game_index=0, game_index_stride=1, compact_tail=True, row_tiers=None,
                 reuse_pass_value=None, reuse_transpositions=None, keep_evaluations=None,
                 opening_book=None, stream=None, rng="philox", numpy_seeds=None):
        """reuse_pass_value / reuse_transpositions / keep_evaluations: None = on when the boards are copied and the evaluator
        declares `row_independent` (the split-f16 evaluator does).  The reference asks the network for every leaf: a node
        without legal moves again on every visit (ai/mcts.py:93-95, 371-397), a position another move order of the same search
        already reached (:385-397), a position the previous move's search evaluated -- and gets the same numbers each time.
        With these options the search takes them from the tree / from a per-game evaluation cache in HBM instead
        (YY_FLAG_REUSE_PASS_VALUE, YY_FLAG_REUSE_TRANSPOSITIONS, YY_FLAG_KEEP_EVALUATIONS in include/yy_engine.h).  The games
        played are the same, move for move; the evaluator sees a fraction of the rows.  The engine owns ONE evaluator for its
        lifetime, which is what keep_evaluations needs.
        opening_book: an engine.OpeningBook built with THIS evaluator, or a stone count N to build one here (every position
        reachable with <= N stones is evaluated once, in large batches, before play; all games start from the empty board, so
        the first plies of every game search the same positions), or None.
        rng: "philox" (default) = the per-game counter streams of csrc/yy_selfplay.hip, drawn on the device; "numpy" = the
        REFERENCE's generator and call sequence, per game: a host numpy RandomState(numpy_seeds[game index], default seed +
        index) draws np.random.dirichlet at the game's first search (ai/mcts.py:305) and np.random.choice for every move
        (ai/self_play.py:146, 160) exactly where SelfPlayWorker.play_game draws from the global stream after np.random.seed(s) --
        a batch of games then replays the reference's transcripts move for move (tests: episodes_*.npz).  One host round
        trip of (pi, legal mask) per move: a parity mode, not the throughput path."""
        assert board_semantics in ("aliased", "copied")
        self.game = game
        self.R, self.C = game.getBoardSize()
        self.A = self.R * self.C
        self.G = int(concurrent_games)
        self.sims = int(num_simulations)
        self.alpha, self.eps = float(dirichlet_alpha), float(dirichlet_epsilon)
        self.thr = int(temperature_threshold)
        self.aliased = board_semantics == "aliased"
        self.quirks = bool(reference_quirks)
        self.rowcol = bool(getattr(game, "rowcol_rule", False))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.evaluator = evaluator
        self.stream = stream                       # HIP stream every launch of this engine goes to (None: the caller's current one)
        self._pending = None                       # a move enqueued by enqueue_move() and not yet finished
        self._evals_seen = 0                       # tree-context counter at the end of the last move (rows_hint)
        if reuse_pass_value is None:
            reuse_pass_value = (not self.aliased) and bool(getattr(evaluator, "row_independent", False))
        if reuse_transpositions is None:
            reuse_transpositions = (not self.aliased) and bool(getattr(evaluator, "row_independent", False))
        if keep_evaluations is None:
            keep_evaluations = (not self.aliased) and bool(getattr(evaluator, "row_independent", False))
        self.reuse_pass_value, self.reuse_transpositions = bool(reuse_pass_value), bool(reuse_transpositions)
        self.keep_evaluations = bool(keep_evaluations)
        self.ctx = engine.BatchedMCTS(self.G, self.R, self.C, self.sims, cpuct=cpuct, aliased=self.aliased,
                                      rowcol=self.rowcol, device=self.device, reuse_pass_value=self.reuse_pass_value,
                                      reuse_transpositions=self.reuse_transpositions, keep_evaluations=self.keep_evaluations)
        if isinstance(opening_book, int):
            opening_book = (engine.OpeningBook(self.R, self.C, evaluator, opening_book, rowcol=self.rowcol, device=self.device)
                            if opening_book > 0 else None)
        self.book = opening_book
        if opening_book is not None:
            self.ctx.set_book(opening_book)
        self.search = LockstepSearch(self.ctx, evaluator, use_graph=use_graph)
        self.seed = int(seed)                      # key of the per-game counter streams (csrc/yy_selfplay.hip)
        assert rng in ("philox", "numpy")
        self.rng, self.numpy_seeds, self._rs = rng, numpy_seeds, {}
        self.n_alive = 0                           # live games, tracked on the host (no device read needed)
        self.first_game_index, self.stride = int(first_game_index), int(game_index_stride)
        # a game has at most A placements; passes never add examples.  Literal quirk mode can make
        # no-op moves (illegal placements), so leave generous room there.
        self.T = self.A + 2 if not self.quirks else 4 * self.A + 8
        dev, G, T = self.device, self.G, self.T
        self.boards = torch.zeros((G, self.R, self.C), dtype=torch.int8, device=dev)
        self.players = torch.ones(G, dtype=torch.int8, device=dev)
        self.ply = torch.zeros(G, dtype=torch.int32, device=dev)
        self.n_ex = torch.zeros(G, dtype=torch.int64, device=dev)
        self.alive = torch.zeros(G, dtype=torch.bool, device=dev)
        self.game_id = torch.full((G,), -1, dtype=torch.int64, device=dev)
        self.hist_state = torch.zeros((G, T, self.R, self.C), dtype=torch.int8, device=dev)
        self.hist_pi = torch.zeros((G, T, self.A), dtype=torch.float32, device=dev)
        self.hist_player = torch.zeros((G, T), dtype=torch.int8, device=dev)
        self.out = []          # finished examples: tuples of device tensors
        self.games_started = 0
        self.games_finished = 0
        self.games_target = 0
        self.positions = 0
        self._ar = torch.arange(G, device=dev)
        self._ones = torch.ones(G, dtype=torch.int8, device=dev)
        # draining batch: once no new game will start, live games are packed to the front and only the first `rows`
        # leaf rows are evaluated (rows = the smallest tier that holds them; one captured step per tier)
        self.compact_tail = bool(compact_tail)
        self.rows = G
        self.tiers = sorted({G} | ({m for m in range(1024, G, 1024)} | {m for m in (512, 256) if m < G}
                                   if row_tiers is None else {int(t) for t in row_tiers if 0 < int(t) < G}))

    def _pack_live_games(self):
        """Once per tier change: stable permutation of every per-slot tensor, live games first."""
        perm = torch.argsort((~self.alive).to(torch.int8), stable=True)
        for name in ("boards", "players", "ply", "n_ex", "alive", "game_id", "hist_state", "hist_pi", "hist_player"):
            setattr(self, name, getattr(self, name).index_select(0, perm).contiguous())

    # ---- slots
    def _start_games(self, slots):
        n = int(slots.numel())
        if n == 0:
            return
        ids = self.first_game_index + (self.games_started + torch.arange(n, device=self.device)) * self.stride
        self.games_started += n
        self.n_alive += n
        self.rows = self.G                            # new games may sit anywhere
        self.boards[slots] = 0
        self.players[slots] = 1                       # black starts (self_play.py:81)
        self.ply[slots] = 0
        self.n_ex[slots] = 0
        self.alive[slots] = True
        self.game_id[slots] = ids

    def _finalize(self, over, result, final_player):
        """Label and emit the examples of the games in `over` (self_play.py:114-121, 170-188), then
        refill their slots."""
        idx = over.nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            return
        n = self.n_ex[idx]
        sel = torch.arange(self.T, device=self.device)[None, :] < n[:, None]          # [k,T]
        res = result[idx].to(torch.float64)
        if self.quirks:
            z = res[:, None].expand(-1, self.T)                                          # Q5: never alternates
        else:
            same = self.hist_player[idx] == final_player[idx][:, None]
            z = torch.where(same, res[:, None], -res[:, None])
        states = self.hist_state[idx]
        if self.quirks and self.aliased:
            states = self.boards[idx][:, None].expand(-1, self.T, -1, -1)               # Q5: all alias the final board
        gid = self.game_id[idx][:, None].expand(-1, self.T)
        plyi = torch.arange(self.T, device=self.device)[None, :].expand(idx.numel(), -1)
        self.out.append((states[sel], self.hist_pi[idx][sel], z[sel].to(torch.float32), gid[sel], plyi[sel]))
        self.games_finished += int(idx.numel())
        self.n_alive -= int(idx.numel())
        self.alive[idx] = False
        self.game_id[idx] = -1
        room = max(0, self.games_target - self.games_started)
        self._start_games(idx[:room])

    # ---- one lockstep move for every live game (self_play.py:91-192)
    # ---- rng="numpy": the reference's generator, one RandomState per game
    def _numpy_stream(self, gid):
        rs = self._rs.get(gid)
        if rs is None:
            seed = self.seed + gid if self.numpy_seeds is None else int(self.numpy_seeds[gid])
            rs = self._rs[gid] = np.random.RandomState(seed)
        return rs

    def _numpy_actions(self, searching, pi, mask_u8):
        """self_play.py:143-160 on the host, game by game, from each game's own RandomState."""
        idx = searching.nonzero(as_tuple=True)[0]
        action = torch.full((self.G,), -1, dtype=torch.int32, device=self.device)
        if idx.numel() == 0:
            return action
        pi_h, m_h = pi[idx].cpu().numpy(), mask_u8[idx].cpu().numpy().astype(np.float64)
        gid_h, ply_h = self.game_id[idx].cpu().numpy(), self.ply[idx].cpu().numpy()
        out = np.zeros(len(gid_h), np.int32)
        for j, gid in enumerate(gid_h):
            rs, p, valid = self._numpy_stream(int(gid)), pi_h[j], m_h[j]
            if ply_h[j] >= self.thr:                                           # temperature 0: random among the most visited
                out[j] = rs.choice(np.where(p == np.max(p))[0])
            else:
                probs = p * valid
                if np.sum(probs) > 0:
                    probs = probs / np.sum(probs)
                else:
                    vi = np.flatnonzero(valid)
                    probs = np.zeros_like(valid)
                    probs[vi] = 1.0 / len(vi)
                out[j] = rs.choice(len(probs), p=probs)
        action[idx] = torch.from_numpy(out).to(self.device)
        return action

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def play_move(self):
        """One lockstep move of every live game: enqueue_move() + finish_move()."""
        self.enqueue_move()
        return self.finish_move()

    def finish_move(self):
        """Waits for the move enqueue_move() put on the engine's stream, reads its one small statistics record (positions
        searched, games finished) and, when games finished, labels their examples and refills their slots."""
        assert self._pending is not None, "finish_move() without enqueue_move()"
        host, event, fin, fin_res, fin_player = self._pending
        self._pending = None
        event.synchronize()
        if hasattr(self.evaluator, "rows_hint"):
            # the stream is idle here: rows the evaluator was asked for per step of this move (also surfaces a failed search now)
            evals = self.ctx.status()["evals"]
            if evals >= self._evals_seen:
                self.evaluator.rows_hint(id(self.search), (evals - self._evals_seen) / float(self.sims + 1))
            self._evals_seen = evals
        n_pos, n_fin = int(host[0]), int(host[1])
        self.positions += n_pos
        if n_fin:
            with self._on_stream():
                self._finalize(fin, fin_res, fin_player)
        return n_pos

    def enqueue_move(self):
        """Every per-move decision is taken on the device with fixed-shape masked operations and ENQUEUED on the engine's
        stream without waiting: the host reads ONE small statistics record per move, in finish_move().  Random draws come from the
        per-game counter streams of csrc/yy_selfplay.hip, so a game's noise and moves depend only on (seed, global game index,
        ply).  Between the two calls the host is free to enqueue the moves of other engines on other streams
        (SelfPlayLanes): their kernels fill the compute units this engine's partly empty last round of workgroups leaves idle."""
        assert self._pending is None, "enqueue_move() twice without finish_move()"
        with self._on_stream():
            self._enqueue_move()

    def _enqueue_move(self):
        dev, G = self.device, self.G
        if self.compact_tail and self.games_started >= self.games_target:      # the batch is draining
            need = next(t for t in self.tiers if t >= self.n_alive)
            if need < self.rows:
                self._pack_live_games()
                self.rows = need
        ones = self._ones
        pending = self.alive.clone()
        searching = torch.zeros(G, dtype=torch.bool, device=dev)
        passes = torch.zeros(G, dtype=torch.int32, device=dev)
        fin = torch.zeros(G, dtype=torch.bool, device=dev)                     # games that end in this move
        fin_res = torch.zeros(G, dtype=torch.float64, device=dev)
        fin_player = self.players.clone()
        for _ in range(2):                                                     # pass handling :103-125
            rp = ones if self.quirks else self.players                         # Q4
            has = engine.valid_mask(self.boards, rp.contiguous(), self.rowcol).bool().any(1)
            go = pending & has
            searching |= go
            pending &= ~go
            nomove = pending & ~has
            passes += nomove.to(torch.int32)
            over = nomove & (passes >= 2)
            r = engine.game_ended(self.boards, self.players, self.rowcol)
            r = torch.where(r == 0, torch.full_like(r, DRAW), r)               # :110-112
            fin |= over
            fin_res = torch.where(over, r, fin_res)
            fin_player = torch.where(over, self.players, fin_player)
            pending &= ~over
            flip = nomove & ~over
            self.players = torch.where(flip, -self.players, self.players)
        searching &= self.alive
        rp = (ones if self.quirks else self.players).contiguous()
        mask_u8 = engine.valid_mask(self.boards, rp, self.rowcol)
        s_u8 = searching.to(torch.uint8)
        noise = None
        if self.eps > 0 and self.rng == "numpy":                               # the reference's own draw, game by game (mcts.py:298-312)
            first = (searching & (self.ply == 0)).nonzero(as_tuple=True)[0]
            noise = torch.zeros((G, self.A), dtype=torch.float64, device=dev)
            if first.numel():
                m_host, gid_host = mask_u8[first].cpu().numpy(), self.game_id[first].cpu().numpy()
                rows = np.zeros((len(gid_host), self.A), np.float64)
                for j, gid in enumerate(gid_host):
                    idx = np.flatnonzero(m_host[j])
                    rows[j, idx] = self._numpy_stream(int(gid)).dirichlet([self.alpha] * len(idx))
                noise[first] = torch.from_numpy(rows).to(dev)
        elif self.eps > 0:                                                     # add_noise = (step == 0), :131
            first = (searching & (self.ply == 0)).to(torch.uint8)
            noise = engine.root_noise(self.seed, self.game_id, self.ply, first, mask_u8, self.alpha)
        self.search.run(self.boards, rp, self.sims, noise=noise, eps=self.eps, active=s_u8, rows=self.rows)
        pi = self.ctx.root_policy()                                            # T == 1 distribution, :329
        # ---- record the example before the move (:140): fixed-shape scatter, rows of idle games rewrite themselves
        ar, slot = self._ar, self.n_ex.clamp_max(self.T - 1)
        self.hist_state[ar, slot] = torch.where(searching[:, None, None], self.boards, self.hist_state[ar, slot])
        self.hist_pi[ar, slot] = torch.where(searching[:, None], pi.to(torch.float32), self.hist_pi[ar, slot])
        self.hist_player[ar, slot] = torch.where(searching, self.players, self.hist_player[ar, slot])
        self.n_ex += searching.to(torch.int64)
        # ---- choose the action (:143-160) from the game's own stream
        if self.rng == "numpy":
            action = self._numpy_actions(searching, pi, mask_u8)
        else:
            action = engine.sample_actions(self.seed, self.game_id, self.ply, s_u8, pi, mask_u8, self.thr)
        # ---- make the move (:163); aliased: the search has mutated the game's board (Q2)
        if self.aliased:
            self.boards = torch.where(searching[:, None, None], self.ctx.boards(), self.boards)
        old_players = self.players
        self.players = self.players.clone()
        engine.step_(self.boards, self.players, action, self.rowcol)
        self.players = torch.where(searching, self.players, old_players)
        self.ply += searching.to(torch.int32)
        ended = engine.game_ended(self.boards, self.players, self.rowcol)      # :167
        done = searching & (ended != 0)
        fin |= done
        fin_res = torch.where(done, ended, fin_res)
        fin_player = torch.where(done, self.players, fin_player)
        stats = torch.stack([searching.sum(), fin.sum()])                      # the move's one host read, taken in finish_move()
        if getattr(self, "_stats_host", None) is None:
            self._stats_host = torch.zeros(2, dtype=stats.dtype).pin_memory()
        self._stats_host.copy_(stats, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._pending = (self._stats_host, event, fin, fin_res, fin_player)

    def begin_run(self, num_games):
        """Arm the engine for `num_games` more games: the first ones start in the free slots, the others as slots free up."""
        with self._on_stream():
            self.games_target = self.games_started + int(num_games)
            free = (~self.alive).nonzero(as_tuple=True)[0]
            self._start_games(free[: int(num_games)])

    def run(self, num_games, progress=None):
        """Play `num_games` games to completion; returns the examples (device tensors)."""
        self.begin_run(num_games)
        moves = 0
        while self.n_alive > 0:
            self.play_move()
            moves += 1
            if moves % 8 == 0:
                self.ctx.status()    # a failed search (NaN from the evaluator, arena overflow) stops the run now, not at its end
            if progress and moves % 10 == 0:
                progress(self)
        self.ctx.status()            # per-game search errors are sticky on the device: any failure of any move raises here
        return self.collect()

    def collect(self):
        out, self.out = self.out, []
        if not out:
            e = torch.empty
            return dict(states=e((0, self.R, self.C), dtype=torch.int8, device=self.device),
                        policies=e((0, self.A), dtype=torch.float32, device=self.device),
                        values=e((0,), dtype=torch.float32, device=self.device),
                        game_id=e((0,), dtype=torch.int64, device=self.device),
                        ply=e((0,), dtype=torch.int64, device=self.device))
        cat = [torch.cat([o[i] for o in out]) for i in range(5)]
        return dict(states=cat[0], policies=cat[1], values=cat[2], game_id=cat[3], ply=cat[4])

    def close(self):
        self.ctx.close()


class _LaneCounters:
    """ctx-like view over the lanes' tree contexts: reset_counters() / status() summed over the lanes."""

    def __init__(self, lanes):
        self.lanes = lanes

    def reset_counters(self):
        for ln in self.lanes:
            with ln._on_stream():
                ln.ctx.reset_counters()

    def status(self):
        tot = {}
        for ln in self.lanes:
            for k, v in ln.ctx.status().items():
                tot[k] = tot.get(k, 0) + v
        return tot


class SelfPlayLanes:
    """K independent SelfPlayEngine lanes of G / K games each on K HIP streams of ONE GPU, their moves enqueued back to back.

    Why: a lockstep step's evaluator launch holds only the leaves that need an evaluation -- ~3 800 of 4 096 without evaluation
    reuse, ~800 with it -- and a tower workgroup occupies a whole compute unit for the ~0.2-0.4 ms its boards take, so the last
    round of workgroups of every launch leaves compute units idle until the launch ends (402 two-board workgroups on 256 CUs =
    1.57 rounds = 2).  Games are independent, so the batch is cut into lanes whose steps are enqueued on separate streams: while
    one lane's launch drains, the other lane's workgroups take the free compute units -- the chip sees one continuous flow of
    workgroups instead of rounds.  A game's transcript depends only on (seed, global game index), never on its lane or slot,
    so the games played are the same as a single engine's (tests/test_gpu_selfplay.py::test_lanes_play_the_same_games).
    Lane k of K plays the games first_game_index + (k + j*K) * game_index_stride, j = 0, 1, ..."""

    def __init__(self, game, evaluator, num_simulations=800, concurrent_games=4096, lanes=2, seed=0, device=None,
                 first_game_index=0, game_index_stride=1, opening_book=None, **engine_kwargs):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        K = max(1, min(int(lanes), int(concurrent_games)))
        self.game, self.evaluator, self.sims = game, evaluator, int(num_simulations)
        if isinstance(opening_book, int):           # one book, shared read-only by every lane
            R, C = game.getBoardSize()
            opening_book = (engine.OpeningBook(R, C, evaluator, opening_book, rowcol=bool(getattr(game, "rowcol_rule", False)),
                                               device=self.device) if opening_book > 0 else None)
        self.book = opening_book
        self.lanes = []
        for k in range(K):
            g_k = concurrent_games // K + (1 if k < concurrent_games % K else 0)
            import contextlib
            with torch.cuda.device(self.device):
                st = torch.cuda.Stream(device=self.device) if K > 1 else None
            # the lane's tensors are allocated under its own stream, so the caching allocator never recycles one of them
            # for another stream while this stream still uses it
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                self.lanes.append(SelfPlayEngine(game, evaluator, num_simulations=num_simulations, concurrent_games=g_k, seed=seed,
                                                 device=self.device, first_game_index=first_game_index + k * game_index_stride,
                                                 game_index_stride=K * game_index_stride, opening_book=opening_book, stream=st,
                                                 **engine_kwargs))
        torch.cuda.synchronize(self.device)      # evaluator weights / the book were written on the caller's stream
        ln = self.lanes[0]
        self.G = sum(l.G for l in self.lanes)
        self.R, self.C, self.A, self.T = ln.R, ln.C, ln.A, ln.T
        self.reuse_pass_value, self.reuse_transpositions, self.keep_evaluations = ln.reuse_pass_value, ln.reuse_transpositions, ln.keep_evaluations
        self.ctx = _LaneCounters(self.lanes)

    positions = property(lambda self: sum(l.positions for l in self.lanes))
    games_finished = property(lambda self: sum(l.games_finished for l in self.lanes))
    n_alive = property(lambda self: sum(l.n_alive for l in self.lanes))

    def play_move(self):
        """One lockstep move of every lane: all lanes' moves are enqueued on their streams, then finished in turn."""
        live = [l for l in self.lanes if l.n_alive > 0]
        for l in live:
            l.enqueue_move()
        return sum(l.finish_move() for l in live)

    def run(self, num_games, progress=None):
        """Play `num_games` games to completion; returns the examples.  The lanes start every move TOGETHER (play_move): with
        evaluation reuse the rows per evaluator launch rise from a dozen to hundreds inside each search, and two lanes in the
        same phase fill the compute units together in the heavy half, while a lane left to run free (its next move enqueued as
        soon as its last one was read back) drifts out of phase and spends its heavy half alone on the chip -- config 2 played
        to completion, same box: 32.8 s in step against 44.0 s free-running with the reuse on, 128.7 s against 132.2 s without."""
        K = len(self.lanes)
        for k, l in enumerate(self.lanes):
            l.begin_run(shard_games(int(num_games), k, K)[0])
        moves = 0
        while self.n_alive > 0:
            self.play_move()
            moves += 1
            if moves % 8 == 0:
                self.ctx.status()    # a failed search (NaN from the evaluator, arena overflow) stops the run now, not at its end
            if progress and moves % 10 == 0:
                progress(self)
        self.ctx.status()            # per-game search errors are sticky on the device: any failure of any move raises here
        return self.collect()

    def collect(self):
        torch.cuda.synchronize(self.device)
        parts = [l.collect() for l in self.lanes]
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}

    def close(self):
        torch.cuda.synchronize(self.device)
        for l in self.lanes:
            l.close()


# =============================================================================== multi-GPU sharding + gather
def shard_games(total, rank, world):
    """Episode sharding (replaces `games_per_worker` processes, self_play.py:299-304): rank r of `world` plays the
    games with global index r, r + world, r + 2*world, ... < total.  Returns (count, first_index, stride); the
    union over ranks is exactly range(total) for any world size, so per-game ids do not depend on it."""
    count = total // world + (1 if rank < total % world else 0)
    return count, rank, world



def example_capacity(total_games, world, rows_per_game):
    """Upper bound of the examples one rank can produce: its share of the games (shard_games) times the most examples a
    game can hold.  Every rank computes the same number, so the exchange needs no size negotiation."""
    return (total_games // world + (1 if total_games % world else 0)) * rows_per_game


def gather_examples(ex, group=None, capacity=None):
    """The ONE exchange of the path (north_star: a single all-gather of the (state, pi, z) examples at iteration end; the
    reference's return_queue.get loop, self_play.py:311-315).  Every rank packs its examples row-wise into one byte buffer
    [header: row count | rows: state int8 R*C, pi f32 A, z f32, game id i64, ply i64] and ONE all_gather_into_tensor
    (RCCL over xGMI on GPUs; gloo on CPU tensors in the tests) delivers all buffers to every rank, which returns the
    concatenation in rank order.  capacity = rows each rank's buffer holds (`example_capacity`; the same on every rank);
    without it a second, 8-byte all-gather of the counts sizes the buffers first."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return ex
    world = dist.get_world_size(group)
    keys = sorted(ex)
    if dist.get_backend(group) == "gloo" and ex[keys[0]].is_cuda:      # gloo gathers host buffers: stage the examples there
        ex = {k: v.cpu() for k, v in ex.items()}
    dev = ex[keys[0]].device
    n = int(ex[keys[0]].shape[0])
    widths = [int(np.prod(ex[k].shape[1:], dtype=np.int64)) * ex[k].element_size() for k in keys]      # bytes per row and key
    cols = [ex[k].contiguous().reshape(n, -1).view(torch.uint8) for k in keys] if n else []
    rb = sum(widths)
    if capacity is None:
        mine = torch.tensor([n], dtype=torch.int64, device=dev)
        every = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(every, mine, group=group)
        capacity = max(int(every.max()), 1)
    if n > capacity:
        raise ValueError(f"gather_examples: {n} examples exceed the agreed capacity {capacity}")
    HDR = 16
    buf = torch.zeros(HDR + capacity * rb, dtype=torch.uint8, device=dev)
    buf[:HDR].view(torch.int64).copy_(torch.tensor([n, rb], dtype=torch.int64))
    if n:
        buf[HDR:HDR + n * rb].view(n, rb).copy_(torch.cat(cols, dim=1))
    out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)                      # <- the single data-path collective
    out = out.view(world, -1)
    hdr = out[:, :HDR].contiguous().view(torch.int64).reshape(world, 2).cpu()
    assert bool((hdr[:, 1] == rb).all()), "ranks disagree on the example row layout"
    rows = torch.cat([out[r, HDR:HDR + int(hdr[r, 0]) * rb].view(-1, rb) for r in range(world)])
    res, off = {}, 0
    for k, w in zip(keys, widths):
        col = rows[:, off:off + w].contiguous()
        res[k] = col.view(ex[k].dtype).reshape((rows.shape[0],) + tuple(ex[k].shape[1:]))
        off += w
    return res


def publish_examples_file(ex, output_dir, reference_format=False):
    """Rank 0 writes self_play_data_<unix_ts>.npz (self_play.py:374-384) to a temporary name and renames it into place, every
    rank waits at a barrier until the file is complete, and all ranks return the SAME path (rank 0's, broadcast): a rank that
    went on to load_data() could otherwise miss the file or open a half-written zip, and a per-rank time stamp could name
    different files."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0
    os.makedirs(output_dir, exist_ok=True)
    filename = os.path.join(output_dir, f"self_play_data_{int(time.time())}.npz")
    n = 1
    while os.path.exists(filename):                    # two publications within one second: never overwrite
        filename = os.path.join(output_dir, f"self_play_data_{int(time.time())}_{n}.npz")
        n += 1
    if multi:
        names = [filename]
        dist.broadcast_object_list(names, src=0)
        filename = names[0]
    if rank == 0:
        # temporary name that the loaders' glob (self_play_data_*.npz) cannot match: a crash before the rename leaves no
        # half-written file that would be picked up as training data
        tmp = f"{filename}.{os.getpid()}.partial"
        states = ex["states"].cpu().numpy()
        with open(tmp, "wb") as fh:
            if reference_format:
                # `boards` = pickled board objects of the reference's own class, readable by ITS TrainingDataQueue.push_file
                from .training import save_examples_reference_format
                save_examples_reference_format(fh, states, ex["policies"].cpu().numpy(), ex["values"].cpu().numpy())
            else:
                np.savez(fh, boards=states, states=states, policies=ex["policies"].cpu().numpy().astype(np.float64),
                         values=ex["values"].cpu().numpy().astype(np.float64), game_id=ex["game_id"].cpu().numpy(),
                         ply=ex["ply"].cpu().numpy())
        os.replace(tmp, filename)
    if multi:
        dist.barrier()
    return filename


# =============================================================================== reference API
class SelfPlayWorker:
    """self_play.py:22-216 with the same constructor; plays games one at a time through MCTS (G == 1)
    drawing from the global numpy stream exactly where the reference does, so identical seeds give
    identical transcripts.  Defaults reproduce the literal reference (aliased boards, quirks)."""

    def __init__(self, game, model_path, num_simulations=800, num_games=1, temperature_threshold=10,
                 dirichlet_alpha=0.3, dirichlet_epsilon=0.25, cpuct=1.0, num_parallel=1,
                 board_semantics="aliased", reference_quirks=True, neural_net=None, device=None):
        self.game, self.model_path = game, model_path
        self.num_simulations, self.num_games = num_simulations, num_games
        self.temperature_threshold = temperature_threshold
        self.dirichlet_alpha, self.dirichlet_epsilon, self.cpuct = dirichlet_alpha, dirichlet_epsilon, cpuct
        self.num_parallel = num_parallel
        self.board_semantics, self.reference_quirks = board_semantics, reference_quirks
        if neural_net is None:
            neural_net = YinYangNeuralNetwork(game)
            if os.path.exists(model_path):
                neural_net.load_model(model_path)
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            neural_net = neural_net.to(dev).eval()    # the reference keeps self-play inference on the CPU (:54-59)
        self.neural_net = neural_net
        self.mcts = MCTS(game, neural_net, num_simulations=num_simulations, cpuct=cpuct,
                         dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon,
                         num_threads=num_parallel, board_semantics=board_semantics, device=device)

    def play_game(self):
        game, examples = self.game, []
        board = game.getInitBoard()
        player, step, passes = 1, 0, 0
        copied = self.board_semantics == "copied"

        def label(result):
            out, value = [], result
            for i, (b, pi, pl) in enumerate(examples):
                if self.reference_quirks:
                    out.append((b, pi, value if i % 2 == 0 else -value))      # :116-118 (nets out to +result)
                    value = -value
                else:
                    out.append((b, pi, result if pl == player else -result))
            return out

        while True:
            temperature = 1.0 if step < self.temperature_threshold else 0
            root_player = 1 if self.reference_quirks else player               # Q4
            valid_moves = game.getValidMoves(board, root_player)
            valid_idx = np.where(valid_moves == 1)[0]
            if len(valid_idx) == 0:
                passes += 1
                if passes >= 2:
                    result = game.getGameEnded(board, player)
                    if result == 0:
                        result = DRAW
                    return label(result)
                player = -player
                continue
            passes = 0
            pi, _root = self.mcts.search(board, root_player, add_exploration_noise=(step == 0))
            examples.append((board, pi, player))
            if temperature == 0:
                action = np.random.choice(np.where(pi == np.max(pi))[0])
            else:
                probs = pi * valid_moves
                if np.sum(probs) > 0:
                    probs = probs / np.sum(probs)
                else:
                    probs = np.zeros_like(valid_moves)
                    probs[valid_idx] = 1.0 / len(valid_idx)
                action = np.random.choice(len(probs), p=probs)
            if copied:                                                         # examples keep the pre-move board
                nb = YinYangLogic(board.n, board.m, getattr(board, "rowcol_rule", False))
                nb.board = board.board.copy()
                board = nb
            board, player = game.getNextState(board, player, action)
            step += 1
            result = game.getGameEnded(board, player)
            if result != 0:
                return label(result)

    def generate_games(self):
        all_examples = []
        for _ in range(self.num_games):
            all_examples.extend(self.play_game())
        return all_examples


class SelfPlayManager:
    """self_play.py:218-335.  The reference forks `num_workers` CPU processes and pickles lists back
    through a queue; here `num_workers * games_per_worker` episodes are sharded over the ranks of the
    running job (one process per GPU, `torchrun`), each rank plays its shard on the batched engine,
    and one all-gather collects the examples."""

    def __init__(self, game, model_path, num_workers=1, num_simulations=800, games_per_worker=1,
                 temperature_threshold=10, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, cpuct=1.0,
                 mcts_parallel=1, concurrent_games=4096, board_semantics="copied", reference_quirks=False,
                 nn_mode="auto", seed=0, num_channels=128, num_res_blocks=10, evaluation_reuse=None,
                 opening_book_stones=None, lanes=None):
        """evaluation_reuse: None = the engine's default (on for copied boards with the float32-accurate evaluator: pass values +
        per-game evaluation cache, SelfPlayEngine); False = the network is asked for every leaf like the reference."""
        self.evaluation_reuse = evaluation_reuse
        self.lanes = lanes                 # HIP streams the rank's games are cut over (SelfPlayLanes); None = 2 from 512 slots on
        # None = 8 stones when it pays: evaluation reuse on, a board of at most 64 cells (770 k positions at 8x8: ~1.5 s to build)
        # and at least 1024 games for this rank; 0 = no book
        self.opening_book_stones = opening_book_stones
        self.game, self.model_path = game, model_path
        self.num_workers, self.games_per_worker = num_workers, games_per_worker
        self.num_simulations, self.temperature_threshold = num_simulations, temperature_threshold
        self.dirichlet_alpha, self.dirichlet_epsilon, self.cpuct = dirichlet_alpha, dirichlet_epsilon, cpuct
        self.mcts_parallel = mcts_parallel
        self.concurrent_games, self.board_semantics, self.reference_quirks = concurrent_games, board_semantics, reference_quirks
        self.nn_mode, self.seed = nn_mode, seed
        self.num_channels, self.num_res_blocks = num_channels, num_res_blocks
        self.stats = {}

    def generate_games_parallel(self):
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        total = self.num_workers * self.games_per_worker
        mine, first, stride = shard_games(total, rank, world)
        dev = torch.device("cuda", torch.cuda.current_device())
        net = YinYangNeuralNetwork(self.game, self.num_channels, self.num_res_blocks)
        if os.path.exists(self.model_path):
            net.load_model(self.model_path)       # every rank reads the same file: no broadcast needed
        net = net.to(dev).eval()
        evaluator = BatchedEvaluator(net, self.nn_mode)
        book = self.opening_book_stones
        if book is None:
            auto = (self.evaluation_reuse is not False and self.board_semantics == "copied" and mine >= 1024
                    and self.game.getActionSize() <= 64 and getattr(evaluator, "row_independent", False))
            book = 8 if auto else 0
        slots = max(1, min(self.concurrent_games, mine))
        eng = SelfPlayLanes(self.game, evaluator, num_simulations=self.num_simulations, opening_book=int(book),
                            lanes=self.lanes if self.lanes else (2 if slots >= 512 else 1),
                             concurrent_games=slots, cpuct=self.cpuct,
                             dirichlet_alpha=self.dirichlet_alpha, dirichlet_epsilon=self.dirichlet_epsilon,
                             temperature_threshold=self.temperature_threshold, board_semantics=self.board_semantics,
                             reference_quirks=self.reference_quirks, seed=1000 + self.seed,   # key of the per-game streams: the same on every rank
                             first_game_index=first, game_index_stride=stride, device=dev,
                             reuse_pass_value=self.evaluation_reuse, reuse_transpositions=self.evaluation_reuse,
                             keep_evaluations=self.evaluation_reuse)
        t0 = time.perf_counter()
        ex = eng.run(mine) if mine > 0 else eng.collect()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        counters = eng.ctx.status()
        ex = gather_examples(ex, capacity=example_capacity(total, world, eng.T))
        self.stats = dict(seconds=t1 - t0, positions=eng.positions, games=eng.games_finished, **counters)
        eng.close()
        return ex


def generate_self_play_data(game, model_path, output_dir, num_games=100, num_workers=1, num_simulations=800,
                            reference_format=False, **engine_kwargs):
    """self_play.py:337-387: same arguments and the same file name pattern.  The .npz holds plain
    tensors (`states` int8 [N,R,C], `policies` float64 [N,A], `values` float64 [N]; `boards` is an
    alias of `states`) instead of pickled board objects; `reference_format=True` writes the reference's pickled-object
    layout as well (training.save_examples_reference_format) so that the reference's training pipeline can read the file."""
    os.makedirs(output_dir, exist_ok=True)
    games_per_worker = max(1, num_games // num_workers)               # :355 (remainder dropped)
    manager = SelfPlayManager(game, model_path, num_workers=num_workers, games_per_worker=games_per_worker,
                              num_simulations=num_simulations, **engine_kwargs)
    ex = manager.generate_games_parallel()
    filename = publish_examples_file(ex, output_dir, reference_format)
    generate_self_play_data.last_stats = manager.stats
    return filename

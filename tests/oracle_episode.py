"""Test-side restatement of SelfPlayWorker.play_game (src/yin_yang/ai/self_play.py:72-192) driven by
the CPU oracle's search and rules, with numpy's legacy RandomState drawing exactly where the
reference draws from the global stream.  TEST INFRASTRUCTURE (small cases only)."""
import numpy as np

import oracle_lib as O


def play_game(R, C, seed, sims, copied, pbits, vbits, quirks=True, temperature_threshold=10, alpha=0.3, eps=0.25):
    rs = np.random.RandomState(seed)
    A = R * C
    board = np.zeros((R, C), np.int8)
    player, step, passes = 1, 0, 0
    trace = dict(search_boards=[], pis=[], actions=[], players=[])
    ex_boards, ex_players = [], []

    def finish(result):
        n = len(trace["pis"])
        if quirks:
            z = [result] * n                                     # Q5 (self_play.py:116-118 nets out to +result)
        else:
            z = [result if p == player else -result for p in ex_players]
        # aliased: every example aliases the one (final) board object; copied: the pre-move boards
        eb = [board.copy() for _ in range(n)] if not copied else ex_boards
        return dict(trace, z=z, example_boards=eb, n=n)

    while True:
        temperature = 1.0 if step < temperature_threshold else 0
        rp = 1 if quirks else player
        valid = O.valid_mask(board[None], rp)[0].astype(np.float64)
        idx = np.flatnonzero(valid)
        if len(idx) == 0:
            passes += 1
            if passes >= 2:
                r = float(O.game_ended(board[None], player)[0])
                return finish(r if r != 0 else 1e-4)
            player = -player
            continue
        passes = 0
        noise = None
        if step == 0:
            noise = np.zeros(A)
            noise[idx] = rs.dirichlet([alpha] * len(idx))
        trace["search_boards"].append(board.copy())
        res = O.search_hash(board, rp, sims, copied, pbits, vbits, noise=noise, eps=eps)
        if not copied:
            board = res.final_board                             # the search mutated the caller's board (Q2)
        pi = res.pi
        trace["pis"].append(pi)
        ex_boards.append(board.copy())
        ex_players.append(player)
        if temperature == 0:
            action = rs.choice(np.where(pi == np.max(pi))[0])
        else:
            p = pi * valid
            if p.sum() > 0:
                p = p / p.sum()
            else:
                p = np.zeros_like(valid)
                p[idx] = 1.0 / len(idx)
            action = rs.choice(len(p), p=p)
        trace["actions"].append(int(action))
        trace["players"].append(int(player))
        nb, npl, _ = O.next_state(board[None], np.array([player], np.int8), np.array([action], np.int32))
        board, player = nb[0], int(npl[0])
        step += 1
        r = float(O.game_ended(board[None], player)[0])
        if r != 0:
            return finish(r)

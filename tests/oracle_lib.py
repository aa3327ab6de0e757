"""ctypes loader for the CPU oracle (oracle/libyy_oracle.so).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  Nothing under the product package imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

c_i8p = C.POINTER(C.c_int8)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, c_i8p, C.c_int, C.c_int, c_f32p, c_f32p)


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libyy_oracle.so")
    src = os.path.join(ORACLE_DIR, "yy_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libyy_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.yyo_game_ended.restype = C.c_double
        L.yyo_search_hash.restype = C.c_long
        L.yyo_search_replay.restype = C.c_long
        L.yyo_search_cb.restype = C.c_long
        L.yyo_search_cb.argtypes = [c_i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                    C.c_int, C.c_double, c_f64p, EVAL_FN, c_i32p, c_f64p, c_f32p,
                                    c_f64p, c_i8p, C.c_long]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def valid_mask(boards, players, flags=0):
    boards = np.ascontiguousarray(boards, np.int8)
    G, R, Cc = boards.shape
    players = np.ascontiguousarray(np.broadcast_to(np.asarray(players, np.int8), (G,)))
    out = np.zeros((G, R * Cc), np.uint8)
    lib().yyo_valid_mask_batch(_p(boards, c_i8p), _p(players, c_i8p), G, R, Cc, flags, _p(out, c_u8p))
    return out


def next_state(boards, players, actions, flags=0):
    """Returns (new_boards, new_players, placed); inputs are not modified."""
    boards = np.array(boards, np.int8, copy=True, order="C")
    G, R, Cc = boards.shape
    players = np.array(np.broadcast_to(np.asarray(players, np.int8), (G,)), np.int8, copy=True)
    actions = np.ascontiguousarray(actions, np.int32)
    placed = np.zeros(G, np.uint8)
    lib().yyo_next_state_batch(_p(boards, c_i8p), _p(players, c_i8p), _p(actions, c_i32p), G, R,
                               Cc, flags, _p(placed, c_u8p))
    return boards, players, placed


def game_ended(boards, players, flags=0):
    boards = np.ascontiguousarray(boards, np.int8)
    G, R, Cc = boards.shape
    players = np.ascontiguousarray(np.broadcast_to(np.asarray(players, np.int8), (G,)))
    out = np.zeros(G, np.float64)
    lib().yyo_game_ended_batch(_p(boards, c_i8p), _p(players, c_i8p), G, R, Cc, flags, _p(out, c_f64p))
    return out


def encode_planes(boards):
    boards = np.ascontiguousarray(boards, np.int8)
    G, R, Cc = boards.shape
    out = np.zeros((G, 5, R, Cc), np.float32)
    lib().yyo_encode_planes_batch(_p(boards, c_i8p), G, R, Cc, _p(out, c_f32p))
    return out


def hash_eval(board, pbits, vbits):
    board = np.ascontiguousarray(board, np.int8)
    R, Cc = board.shape
    pol = np.zeros(R * Cc, np.float32)
    val = C.c_float(0)
    lib().yyo_hash_eval(_p(board, c_i8p), R, Cc, pbits, vbits, _p(pol, c_f32p), C.byref(val))
    return pol, np.float32(val.value)


class SearchResult:
    pass


def _alloc(R, Cc, leaf_cap):
    A = R * Cc
    r = SearchResult()
    r.counts = np.zeros(A, np.int32)
    r.child_w = np.zeros(A, np.float64)
    r.child_p = np.zeros(A, np.float32)
    r.root_stats = np.zeros(2, np.float64)
    r.leaves = np.zeros((max(leaf_cap, 1), R, Cc), np.int8)
    return r


def _finish(r, board, n, leaf_cap):
    r.n_evals = int(n)
    r.final_board = board
    r.leaves = r.leaves[: min(r.n_evals, leaf_cap)]
    s = r.counts.sum()
    A = r.counts.size
    r.pi = (r.counts / s) if s > 0 else np.ones(A) / A          # mcts.py:209-213 (T == 1)
    r.root_visits = int(r.root_stats[0])
    r.root_w = float(r.root_stats[1])
    return r


def search_hash(board, root_player, sims, copied, pbits, vbits, noise=None, eps=0.25, cpuct=1.0,
                flags=0, leaf_cap=0):
    board = np.array(board, np.int8, copy=True, order="C")
    R, Cc = board.shape
    r = _alloc(R, Cc, leaf_cap)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
    n = lib().yyo_search_hash(_p(board, c_i8p), R, Cc, int(root_player), int(sims),
                              C.c_float(cpuct), int(copied), flags, C.c_double(eps),
                              _p(nz, c_f64p), int(pbits), int(vbits), _p(r.counts, c_i32p),
                              _p(r.child_w, c_f64p), _p(r.child_p, c_f32p),
                              _p(r.root_stats, c_f64p), _p(r.leaves, c_i8p), C.c_long(leaf_cap))
    return _finish(r, board, n, leaf_cap)


def search_replay(board, root_player, sims, copied, rec_policy, rec_value, noise=None, eps=0.25,
                  cpuct=1.0, flags=0, leaf_cap=0):
    board = np.array(board, np.int8, copy=True, order="C")
    R, Cc = board.shape
    r = _alloc(R, Cc, leaf_cap)
    rp = np.ascontiguousarray(rec_policy, np.float32)
    rv = np.ascontiguousarray(rec_value, np.float32)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
    n = lib().yyo_search_replay(_p(board, c_i8p), R, Cc, int(root_player), int(sims),
                                C.c_float(cpuct), int(copied), flags, C.c_double(eps),
                                _p(nz, c_f64p), _p(rp, c_f32p), _p(rv, c_f32p),
                                C.c_long(rp.shape[0]), _p(r.counts, c_i32p), _p(r.child_w, c_f64p),
                                _p(r.child_p, c_f32p), _p(r.root_stats, c_f64p),
                                _p(r.leaves, c_i8p), C.c_long(leaf_cap))
    return _finish(r, board, n, leaf_cap)


def search_callback(board, root_player, sims, copied, predict, noise=None, eps=0.25, cpuct=1.0,
                    flags=0, leaf_cap=0):
    """predict(board_i8[R,C]) -> (policy f32[A], value f32); called in reference order."""
    board = np.array(board, np.int8, copy=True, order="C")
    R, Cc = board.shape
    A = R * Cc
    r = _alloc(R, Cc, leaf_cap)

    def _cb(user, bptr, rr, cc, pptr, vptr):
        arr = np.ctypeslib.as_array(bptr, shape=(rr, cc))
        pol, val = predict(arr)
        np.ctypeslib.as_array(pptr, shape=(A,))[:] = pol
        vptr[0] = float(val)

    cb = EVAL_FN(_cb)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
    n = lib().yyo_search_cb(_p(board, c_i8p), R, Cc, int(root_player), int(sims), C.c_float(cpuct),
                            int(copied), flags, C.c_double(eps), _p(nz, c_f64p), cb,
                            _p(r.counts, c_i32p), _p(r.child_w, c_f64p), _p(r.child_p, c_f32p),
                            _p(r.root_stats, c_f64p), _p(r.leaves, c_i8p), C.c_long(leaf_cap))
    return _finish(r, board, n, leaf_cap)

"""Torch-tensor level binding of the HIP hot path (include/yy_engine.h).

PyTorch is plumbing here: it owns the HBM tensors and the stream; every operation below is a
hand-written gfx950 kernel in csrc/yy_engine.hip.  No CPU fallback exists: tensors must be on a
ROCm device and the extension must be built, otherwise these functions raise.
"""
import ctypes as ct

import torch

from . import _lib
from ._lib import FLAG_ALIASED, FLAG_REUSE_PASS_VALUE, FLAG_KEEP_EVALUATIONS, FLAG_REUSE_TRANSPOSITIONS, FLAG_ROWCOL, MctsConfig, check, lib


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ct.c_void_p(t.data_ptr())


def _need(t, dtype, shape=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.YYError(-1, f"{name}: expected a ROCm device tensor (the hot path has no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.YYError(-1, f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.YYError(-1, f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _flags(rowcol=False, aliased=False):
    return (FLAG_ROWCOL if rowcol else 0) | (FLAG_ALIASED if aliased else 0)


# ------------------------------------------------------------------ stateless rules kernels
def valid_mask(boards, players, rowcol=False):
    """getValidMoves for a batch (yin_yang_game.py:60-78): int8 [G,R,C], int8 [G] -> uint8 [G,A]."""
    G, R, Cc = boards.shape
    _need(boards, torch.int8, name="boards")
    _need(players, torch.int8, (G,), "players")
    out = torch.empty((G, R * Cc), dtype=torch.uint8, device=boards.device)
    with torch.cuda.device(boards.device):
        check(lib().yy_rules_valid_mask(_p(boards), _p(players), G, R, Cc, _flags(rowcol), _p(out), _stream()))
    return out


def step_(boards, players, actions, rowcol=False):
    """getNextState IN PLACE (yin_yang_game.py:39-58): boards and players are updated; returns the
    uint8 [G] `placed` bits (illegal placements are silently skipped, the player still flips)."""
    G, R, Cc = boards.shape
    _need(boards, torch.int8, name="boards")
    _need(players, torch.int8, (G,), "players")
    _need(actions, torch.int32, (G,), "actions")
    placed = torch.empty(G, dtype=torch.uint8, device=boards.device)
    with torch.cuda.device(boards.device):
        check(lib().yy_rules_step(_p(boards), _p(players), _p(actions), G, R, Cc, _flags(rowcol), _p(placed), _stream()))
    return placed


def game_ended(boards, players, rowcol=False, with_counts=False):
    """getGameEnded (yin_yang_game.py:80-110): float64 [G] in {0, 1, -1, 0.0001}."""
    G, R, Cc = boards.shape
    _need(boards, torch.int8, name="boards")
    _need(players, torch.int8, (G,), "players")
    out = torch.empty(G, dtype=torch.float64, device=boards.device)
    counts = torch.empty((G, 2), dtype=torch.int32, device=boards.device) if with_counts else None
    with torch.cuda.device(boards.device):
        check(lib().yy_rules_game_ended(_p(boards), _p(players), G, R, Cc, _flags(rowcol), _p(out), _p(counts), _stream()))
    return (out, counts) if with_counts else out


def encode_planes(boards, out=None):
    """board_to_input for a batch (neural_network.py:156-196): float32 [G,5,R,C]."""
    G, R, Cc = boards.shape
    _need(boards, torch.int8, name="boards")
    if out is None:
        out = torch.empty((G, 5, R, Cc), dtype=torch.float32, device=boards.device)
    _need(out, torch.float32, (G, 5, R, Cc), "out")
    with torch.cuda.device(boards.device):
        check(lib().yy_encode_planes(_p(boards), G, R, Cc, _p(out), _stream()))
    return out


def pack_boards(boards):
    """int8 [G,R,C] -> (black, white) uint64-as-int64 [NW,G] word-major bitboards."""
    G, R, Cc = boards.shape
    _need(boards, torch.int8, name="boards")
    nw = (R * Cc + 63) // 64
    black = torch.empty((nw, G), dtype=torch.int64, device=boards.device)
    white = torch.empty((nw, G), dtype=torch.int64, device=boards.device)
    with torch.cuda.device(boards.device):
        check(lib().yy_pack_boards(_p(boards), G, R, Cc, _p(black), _p(white), _stream()))
    return black, white


def unpack_boards(black, white, R, Cc):
    nw, G = black.shape
    _need(black, torch.int64, name="black")
    _need(white, torch.int64, (nw, G), "white")
    boards = torch.empty((G, R, Cc), dtype=torch.int8, device=black.device)
    with torch.cuda.device(black.device):
        check(lib().yy_unpack_boards(_p(black), _p(white), G, R, Cc, _p(boards), _stream()))
    return boards


def mask_terminal_bb(black, white, R, Cc, rowcol=False, out=None):
    """Packed rules kernel: both colours' legal masks + game-ended code from BLACK's view
    (0 ongoing, 1, -1, 2 draw) for bitboards [NW,G]."""
    nw, G = black.shape
    _need(black, torch.int64, name="black")
    _need(white, torch.int64, (nw, G), "white")
    if out is None:
        out = (torch.empty_like(black), torch.empty_like(black),
               torch.empty(G, dtype=torch.int8, device=black.device))
    m1, m2, res = out
    with torch.cuda.device(black.device):
        check(lib().yy_rules_mask_terminal_bb(_p(black), _p(white), G, R, Cc, _flags(rowcol), _p(m1), _p(m2), _p(res), _stream()))
    return m1, m2, res


def bias_act_(x, bias, residual=None, relu=True):
    """In-place fused epilogue on a channels-last bf16 activation tensor [N,C,H,W] (NHWC memory):
    x = relu?(x + bias[c] (+ residual)).  One HIP pass (csrc k_bias_act)."""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype != torch.bfloat16 or not x.is_cuda:
        raise _lib.YYError(-1, "bias_act_: expected a channels-last bf16 device tensor [N,C,H,W]")
    N, Cc, H, W = x.shape
    if bias.dtype != torch.float32 or bias.numel() != Cc or not bias.is_cuda:
        raise _lib.YYError(-1, "bias_act_: bias must be float32 [C] on the device")
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.YYError(-1, "bias_act_: residual must match x (channels-last bf16)")
    with torch.cuda.device(x.device):
        check(lib().yy_nn_bias_act_bf16(_p(x), _p(bias), _p(residual), N * H * W, Cc, int(bool(relu)), _stream()))
    return x


def tower_forward(planes, weights, bias, n_layers):
    """Stem + residual tower in one LDS-resident MFMA kernel (csrc/yy_tower.hip, yy_towerq.hip).
    planes f32 [G,5,R,R] (R = 8 or 12) -> bf16 activations as a channels-last tensor [G,128,R,R]."""
    G, _, R, Cc = planes.shape
    _need(planes, torch.float32, (G, 5, R, Cc), "planes")
    n_chunks = 9 + 18 * (n_layers - 1)
    _need(weights, torch.int16, (n_chunks, 8192), "tower weights")
    _need(bias, torch.float32, (n_layers, 128), "tower bias")
    out = torch.empty((G, R, Cc, 128), dtype=torch.bfloat16, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_bf16(_p(planes), _p(weights), _p(bias), _p(out), G, R, Cc, 128, n_layers, _stream()))
    return out.permute(0, 3, 1, 2)      # NCHW view of NHWC memory == channels_last


def tower_forward_f32(planes, weights, bias, n_layers):
    """Stem + residual tower in exact float32 on the f32 MFMA (csrc/yy_tower_f32.hip).
    planes f32 [G,5,8,8] -> f32 activations as a channels-last tensor [G,128,8,8]."""
    G = planes.shape[0]
    _need(planes, torch.float32, (G, 5, 8, 8), "planes")
    _need(weights, torch.float32, (9 + 36 * (n_layers - 1), 4096), "f32 tower weights")
    _need(bias, torch.float32, (n_layers, 128), "tower bias")
    out = torch.empty((G, 8, 8, 128), dtype=torch.float32, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_f32(_p(planes), _p(weights), _p(bias), _p(out), G, 8, 8, 128, n_layers, _stream()))
    return out.permute(0, 3, 1, 2)


def tower_heads_forward_h3r(planes, weights, head_w, bias, n_layers, exps, rows=None, n_rows=None, out=None):
    """The 32x32x16 split-f16 tower of round 2 (csrc/yy_tower_h3r.hip; boards 6x6 / 8x8 / 12x12, 128 channels): the A/B partner
    of tower_g.  planes f32 [G,5,R,R] -> f32 [G,2,32*R*R] head features; weights in the wave-major order of
    network.pack_tower_h3r / pack_heads_h3r; rows / n_rows as for tower_g."""
    G, _, R, Cc = planes.shape
    _need(planes, torch.float32, (G, 5, R, Cc), "planes")
    _need(weights, torch.int16, (9 + 36 * (n_layers - 1), 8192), "wave-major split-f16 tower weights")
    _need(head_w, torch.int16, (2, 8192), "wave-major split-f16 head weights")
    _need(bias, torch.float32, (n_layers + 1, 128), "tower+heads bias")
    if rows is not None:
        _need(rows, torch.int32, (G,), "rows")
        _need(n_rows, torch.int32, (1,), "n_rows")
    if out is None:
        out = torch.empty((G, 2, 32 * R * Cc), dtype=torch.float32, device=planes.device)
    _need(out, torch.float32, (G, 2, 32 * R * Cc), "out")
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_f16x3_regs(_p(planes), _p(weights), _p(head_w), _p(bias), None, _p(out), _p(rows), _p(n_rows), G, R, Cc,
                                           128, n_layers, int(exps[0]), int(exps[1]), int(exps[2]), _stream()))
    return out


def tower_forward_h3r(planes, weights, bias, n_layers, exps):
    """Tower activations f32 [G,128,R,R] (channels-last memory) from the register-ring kernel (tests)."""
    G, _, R, Cc = planes.shape
    _need(planes, torch.float32, (G, 5, R, Cc), "planes")
    _need(weights, torch.int16, (9 + 36 * (n_layers - 1), 8192), "wave-major split-f16 tower weights")
    _need(bias, torch.float32, (n_layers, 128), "tower bias")
    out = torch.empty((G, R, Cc, 128), dtype=torch.float32, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_f16x3_regs(_p(planes), _p(weights), None, _p(bias), _p(out), None, None, None, G, R, Cc, 128, n_layers,
                                           int(exps[0]), 0, int(exps[2]), _stream()))
    return out.permute(0, 3, 1, 2)


def tower_g_available(channels):
    """Column-block counts the general split-f16 tower kernel is instantiated for at this channel count."""
    buf = (ct.c_int * 8)()
    n = lib().yy_nn_tower_g_forms(int(channels), buf)
    return [int(buf[i]) for i in range(n)]


def tower_g(planes, weights, bias, n_layers, exps, nb, boards, head_w=None, head_bias=None, rows=None, n_rows=None, out=None,
            gate=(-1, 0x7FFFFFFF)):
    """General split-f16 tower (csrc/yy_tower_g.hip): planes f32 [G,5,R,C] (R*C <= 144) with the weights of network.pack_tower_g.
    With head_w / head_bias (pack_heads_g): -> f32 [G,2,32*R*C] = (policy features, value features) in the reference's flatten
    order; without: -> the tower activations f32 [G,CH,R,C] (channels-last memory).  nb column blocks of 16 and `boards` boards
    per workgroup (network.tower_g_forms).  rows int32 [G] / n_rows int32 [1] (device): evaluate planes[rows[i]] for i < n_rows
    into dense row i; gate = (lo, hi): the launch only runs when lo < live rows <= hi."""
    G, _, R, Cc = planes.shape
    _need(planes, torch.float32, (G, 5, R, Cc), "planes")
    ch = bias.shape[1]
    nw = ch // 32
    _need(weights, torch.int16, (9 + 9 * nw * (n_layers - 1), ch * 64), "split-f16 tower weights (pack_tower_g)")
    _need(bias, torch.float32, (n_layers, ch), "tower bias")
    if rows is not None:
        _need(rows, torch.int32, (G,), "rows")
        _need(n_rows, torch.int32, (1,), "n_rows")
    heads = head_w is not None
    if heads:
        _need(head_w, torch.int16, (4 * nw * 2 * 512,), "split-f16 head weights (pack_heads_g)")
        _need(head_bias, torch.float32, (64,), "head bias")
        if out is None:
            out = torch.empty((G, 2, 32 * R * Cc), dtype=torch.float32, device=planes.device)
        _need(out, torch.float32, (G, 2, 32 * R * Cc), "out")
    else:
        out = torch.empty((G, R, Cc, ch), dtype=torch.float32, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_g(_p(planes), _p(weights), _p(head_w), _p(bias), _p(head_bias), None if heads else _p(out),
                                  _p(out) if heads else None, _p(rows), _p(n_rows), G, R, Cc, ch, n_layers, int(exps[0]), int(exps[1]),
                                  int(exps[2]), int(nb), int(boards), int(gate[0]), int(gate[1]), _stream()))
    return out if heads else out.permute(0, 3, 1, 2)


def fc_heads(feats, wpk, bias, jobs, A, H, exps, n_rows=None, logits=None, hidden=None):
    """policy_fc and value_fc1 on the dense feature rows (csrc/yy_fc_heads.hip): feats f32 [G,2,K] -> (logits f32 [G,A], hidden
    f32 [G,H], bias added); weights / bias / jobs from network.pack_fc_heads, exps = (kw, ka).  n_rows int32 [1] (device): rows
    past it are neither read nor written.  A row's results do not depend on G, n_rows or its position in the batch."""
    G, two, K = feats.shape
    _need(feats, torch.float32, (G, 2, K), "feats")
    _need(jobs, torch.int32, (jobs.shape[0], 4), "jobs")
    ksteps = ((K + 127) // 128) * 4
    _need(wpk, torch.int16, (jobs.shape[0] * ksteps * 4096,), "fc head weights (pack_fc_heads)")
    _need(bias, torch.float32, (A + H,), "fc head bias")
    if n_rows is not None:
        _need(n_rows, torch.int32, (1,), "n_rows")
    if logits is None:
        logits = torch.empty((G, A), dtype=torch.float32, device=feats.device)
        hidden = torch.empty((G, H), dtype=torch.float32, device=feats.device)
    _need(logits, torch.float32, (G, A), "logits")
    _need(hidden, torch.float32, (G, H), "hidden")
    with torch.cuda.device(feats.device):
        check(lib().yy_nn_fc_heads_f16x3(_p(feats), _p(wpk), _p(bias), _p(jobs), jobs.shape[0], _p(logits), _p(hidden), _p(n_rows), G, K,
                                         A, H, int(exps[0]), int(exps[1]), _stream()))
    return logits, hidden


def head_finish_f32(logits, hidden, w2, b2, rows=None, n_rows=None, policy=None, value=None):
    """softmax(logits f32 [G,A]) and tanh(relu(hidden f32 [G,H]) @ w2 + b2) in one HIP pass (csrc k_head_finish_f32);
    with rows / n_rows the result of dense row i lands in row rows[i] of (policy, value) for i < n_rows."""
    G, A = logits.shape
    _need(logits, torch.float32, name="logits")
    _need(hidden, torch.float32, (G, hidden.shape[1]), "hidden")
    _need(w2, torch.float32, (hidden.shape[1],), "w2")
    _need(b2, torch.float32, (1,), "b2")
    if rows is not None:
        _need(rows, torch.int32, (G,), "rows")
        _need(n_rows, torch.int32, (1,), "n_rows")
    if policy is None:
        policy = torch.zeros((G, A), dtype=torch.float32, device=logits.device)
        value = torch.zeros(G, dtype=torch.float32, device=logits.device)
    _need(policy, torch.float32, (G, A), "policy")
    _need(value, torch.float32, (G,), "value")
    with torch.cuda.device(logits.device):
        check(lib().yy_nn_head_finish_f32(_p(logits), _p(hidden), G, A, hidden.shape[1], _p(w2), _p(b2), _p(rows), _p(n_rows),
                                          _p(policy), _p(value), _stream()))
    return policy, value


def compact_rows(flags, rows=None, n=None):
    """uint8 [G] flags -> (rows int32 [G], n int32 [1]): rows[:n] = ascending indices of the non-zero flags."""
    G = flags.shape[0]
    _need(flags, torch.uint8, (G,), "flags")
    if rows is None:
        rows = torch.zeros(G, dtype=torch.int32, device=flags.device)
        n = torch.zeros(1, dtype=torch.int32, device=flags.device)
    _need(rows, torch.int32, (G,), "rows")
    _need(n, torch.int32, (1,), "n")
    with torch.cuda.device(flags.device):
        check(lib().yy_compact_rows(_p(flags), G, _p(rows), _p(n), _stream()))
    return rows, n


def tower_heads_forward(planes, weights, bias, n_layers):
    """Tower + fused 1x1 head convolutions: planes f32 [G,5,R,R] (R = 8 or 12) ->
    bf16 [G,2,32*R*R] = (policy features, value features) in the reference's flatten order."""
    G, _, R, Cc = planes.shape
    _need(planes, torch.float32, (G, 5, R, Cc), "planes")
    n_chunks = 9 + 18 * (n_layers - 1) + 1
    _need(weights, torch.int16, (n_chunks, 8192), "tower+heads weights")
    _need(bias, torch.float32, (n_layers + 1, 128), "tower+heads bias")
    out = torch.empty((G, 2, 32 * R * Cc), dtype=torch.bfloat16, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib().yy_nn_tower_heads_bf16(_p(planes), _p(weights), _p(bias), _p(out), G, R, Cc, 128, n_layers, _stream()))
    return out


def head_finish(h, A, w2, b2):
    """softmax over the first A columns of h (bf16 [G, A+H]) and tanh(relu(h[:, A:]) @ w2 + b2) in one
    HIP pass (csrc k_head_finish) -> (policy f32 [G,A], value f32 [G])."""
    G, W = h.shape
    _need(h, torch.bfloat16, name="h")
    _need(w2, torch.float32, (W - A,), "w2")
    _need(b2, torch.float32, (1,), "b2")
    policy = torch.empty((G, A), dtype=torch.float32, device=h.device)
    value = torch.empty(G, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        check(lib().yy_nn_head_finish_bf16(_p(h), G, A, W - A, _p(w2), _p(b2), _p(policy), _p(value), _stream()))
    return policy, value


# ------------------------------------------------------------------ episode-loop random draws (csrc/yy_selfplay.hip)
def root_noise(seed, game_id, ply, draw, mask, alpha):
    """Dirichlet(alpha) noise over the legal cells of the games with draw != 0, keyed by (seed, game_id, ply): float64 [G,A]."""
    G, A = mask.shape
    _need(game_id, torch.int64, (G,), "game_id")
    _need(ply, torch.int32, (G,), "ply")
    _need(draw, torch.uint8, (G,), "draw")
    _need(mask, torch.uint8, (G, A), "mask")
    out = torch.empty((G, A), dtype=torch.float64, device=mask.device)
    with torch.cuda.device(mask.device):
        check(lib().yy_selfplay_root_noise(ct.c_uint64(int(seed) & (2 ** 64 - 1)), _p(game_id), _p(ply), _p(draw), _p(mask), G, A,
                                           float(alpha), _p(out), _stream()))
    return out


def sample_actions(seed, game_id, ply, searching, pi, mask, temperature_threshold):
    """The move choice of play_game (self_play.py:143-160), keyed by (seed, game_id, ply): int32 [G], -1 where not searching."""
    G, A = mask.shape
    _need(game_id, torch.int64, (G,), "game_id")
    _need(ply, torch.int32, (G,), "ply")
    _need(searching, torch.uint8, (G,), "searching")
    _need(pi, torch.float64, (G, A), "pi")
    _need(mask, torch.uint8, (G, A), "mask")
    out = torch.empty(G, dtype=torch.int32, device=mask.device)
    with torch.cuda.device(mask.device):
        check(lib().yy_selfplay_sample_actions(ct.c_uint64(int(seed) & (2 ** 64 - 1)), _p(game_id), _p(ply), _p(searching), _p(pi),
                                               _p(mask), G, A, int(temperature_threshold), _p(out), _stream()))
    return out


# ------------------------------------------------------------------ shared book of pre-evaluated opening positions
class OpeningBook:
    """Every position with at most `max_stones` stones that alternating legal play reaches from the empty board (8x8: 770 232
    for max_stones = 8), evaluated ONCE with `evaluator` in batches of `batch` rows and stored as an open-addressing table in
    HBM [position -> policy row f32[A], value].  BatchedMCTS.set_book(book) makes every search look its shallow leaves up
    there before asking the evaluator (include/yy_engine.h: yy_mcts_set_book): all games start from the empty board, so the
    first plies of thousands of games walk the same positions.  The evaluator must be the one the searches use, and a
    deterministic function of the row (BatchedEvaluator.row_independent); the results of a search are then unchanged.
    max_positions bounds the table: enumeration stops before the stone count that would exceed it (max_stones is lowered)."""

    def __init__(self, R, C, evaluator, max_stones, rowcol=False, batch=4096, device=None, max_positions=8_000_000):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.R, self.C, self.A, self.max_stones = int(R), int(C), int(R) * int(C), int(max_stones)
        nw = (self.A + 63) // 64
        boards = torch.zeros((1, R, C), dtype=torch.int8, device=dev)
        player, keys, states = 1, [], []
        for _ in range(self.max_stones):                       # breadth-first over stone counts; a side without a move ends a line
            players = torch.full((boards.shape[0],), player, dtype=torch.int8, device=dev)
            vm = valid_mask(boards, players, rowcol)
            n_children = int(vm.sum())                          # before de-duplication: what the expansion below allocates
            if n_children == 0:
                break
            if n_children > 8 * max_positions:                  # a frontier this wide cannot fit the budget: stop BEFORE expanding it
                self.max_stones = len(keys)
                break
            bi, a = vm.nonzero(as_tuple=True)
            child = boards.reshape(-1, self.A)[bi]
            child[torch.arange(bi.numel(), device=dev), a] = player
            black, white = pack_boards(child.view(-1, R, C))
            key = torch.cat([black.t(), white.t()], dim=1).contiguous()                 # [m, 2*NW]
            uniq, inv = torch.unique(key, dim=0, return_inverse=True)
            first = torch.zeros(uniq.shape[0], dtype=torch.int64, device=dev)
            first[inv] = torch.arange(key.shape[0], device=dev)                         # any representative of each position
            if sum(k.shape[0] for k in keys) + uniq.shape[0] > max_positions:           # keep whole stone counts only
                self.max_stones = len(keys)
                break
            boards = child[first].view(-1, R, C).contiguous()
            keys.append(uniq)
            states.append(boards)
            player = -player
        if not keys:
            raise _lib.YYError(-1, "OpeningBook: nothing to store (max_stones < 1 or max_positions too small)")
        self.keys = torch.cat(keys).contiguous()
        boards = torch.cat(states)
        self.n = int(self.keys.shape[0])
        pol = torch.empty((self.n, self.A), dtype=torch.float32, device=dev)
        val = torch.empty(self.n, dtype=torch.float32, device=dev)
        for lo in range(0, self.n, batch):                                              # the evaluations: n / batch launches
            p, v = evaluator(encode_planes(boards[lo:lo + batch].contiguous()))
            pol[lo:lo + batch], val[lo:lo + batch] = p, v
        self.cap = 64
        while self.cap < 6 * self.n:                   # load <= 1/6: next to no probe run reaches the 8-step window
            self.cap *= 2
        self.meta = torch.zeros(self.cap, dtype=torch.int32, device=dev)
        self.table_keys = torch.zeros((self.cap, 2 * nw), dtype=torch.int64, device=dev)
        slot = torch.empty(self.n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(lib().yy_book_insert(_p(self.keys), self.n, R, C, _p(self.meta), _p(self.table_keys), self.cap, _p(slot), _stream()))
        ok = slot >= 0
        self.stored = int(ok.sum())
        idx = slot[ok].long()
        self.value = torch.zeros(self.cap, dtype=torch.float32, device=dev)
        self.policy = torch.zeros((self.cap, self.A), dtype=torch.float32, device=dev)
        self.value[idx] = val[ok]
        self.policy[idx] = pol[ok]
        self.bytes = sum(t.numel() * t.element_size() for t in (self.meta, self.table_keys, self.value, self.policy))


# ------------------------------------------------------------------ batched MCTS context
class BatchedMCTS:
    """G games searched in lockstep on one GPU; replaces Node + MCTS.search/_simulate
    (ai/mcts.py:28-225, 275-414).  One game per wavefront; see csrc/yy_engine.hip."""

    COUNTERS = ("evals", "levels", "children_scanned", "children_created", "terminal_revisits", "nodes", "reused_values",
                "transposition_hits")

    def __init__(self, G, R, C, max_sims, cpuct=1.0, aliased=False, rowcol=False, device=None,
                 edges_per_game=0, nodes_per_game=0, reuse_pass_value=False, reuse_transpositions=False,
                 keep_evaluations=False):
        """reuse_pass_value (copied boards only): a childless non-terminal node keeps the value of its first
        evaluation instead of being evaluated again on every visit (YY_FLAG_REUSE_PASS_VALUE, include/yy_engine.h);
        needs a deterministic evaluator whose row results do not depend on the rest of the batch.
        reuse_transpositions: a leaf whose position was already evaluated in this search takes the cached policy row and
        value instead of an evaluator row (YY_FLAG_REUSE_TRANSPOSITIONS); same requirement.
        keep_evaluations: the cache also serves the later searches of the context (YY_FLAG_KEEP_EVALUATIONS): every search
        must then use the same evaluator -- call clear_evaluation_cache() when the network changes."""
        if not torch.cuda.is_available():
            raise _lib.YYError(-100, "BatchedMCTS needs a ROCm device: the hot path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.G, self.R, self.C, self.A = int(G), int(R), int(C), int(R) * int(C)
        self.max_sims, self.cpuct, self.aliased, self.rowcol = int(max_sims), float(cpuct), bool(aliased), bool(rowcol)
        self.reuse_pass_value, self.reuse_transpositions = bool(reuse_pass_value), bool(reuse_transpositions)
        self.keep_evaluations = bool(keep_evaluations)
        cfg = MctsConfig(self.G, self.R, self.C, self.max_sims, self.cpuct,
                         _flags(rowcol, aliased) | (FLAG_REUSE_PASS_VALUE if reuse_pass_value else 0)
                         | (FLAG_REUSE_TRANSPOSITIONS if reuse_transpositions else 0)
                         | (FLAG_KEEP_EVALUATIONS if keep_evaluations else 0),
                         int(edges_per_game), int(nodes_per_game))
        h = ct.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_create(ct.byref(cfg), ct.byref(h)))
        self._h = h
        self.planes = torch.zeros((self.G, 5, self.R, self.C), dtype=torch.float32, device=self.device)
        self.needs_eval = torch.zeros(self.G, dtype=torch.uint8, device=self.device)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            lib().yy_mcts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def memory_bytes(self):
        n = ct.c_uint64(0)
        check(lib().yy_mcts_memory_bytes(self._h, ct.byref(n)))
        return n.value

    # -- the C ABI, one method per entry point
    def begin(self, boards, root_players, active=None):
        _need(boards, torch.int8, (self.G, self.R, self.C), "boards")
        _need(root_players, torch.int8, (self.G,), "root_players")
        if active is not None:
            _need(active, torch.uint8, (self.G,), "active")
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_begin(self._h, _p(boards), _p(root_players), _p(active), _p(self.planes), _stream()))

    def expand_root(self, policy, noise=None, eps=0.25):
        _need(policy, torch.float32, (self.G, self.A), "policy")
        if noise is not None:
            _need(noise, torch.float64, (self.G, self.A), "noise")
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_expand_root(self._h, _p(policy), _p(noise), float(eps), _stream()))

    def select(self):
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_select(self._h, _p(self.planes), _p(self.needs_eval), _stream()))

    def expand_backup(self, policy, value):
        _need(policy, torch.float32, (self.G, self.A), "policy")
        _need(value, torch.float32, (self.G,), "value")
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_expand_backup(self._h, _p(policy), _p(value), _stream()))

    def step(self, policy, value):
        _need(policy, torch.float32, (self.G, self.A), "policy")
        _need(value, torch.float32, (self.G,), "value")
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_step(self._h, _p(policy), _p(value), _p(self.planes), _p(self.needs_eval), _stream()))

    def root_counts(self, with_children=False):
        counts = torch.empty((self.G, self.A), dtype=torch.int32, device=self.device)
        cw = torch.empty((self.G, self.A), dtype=torch.float32, device=self.device) if with_children else None
        cp = torch.empty((self.G, self.A), dtype=torch.float32, device=self.device) if with_children else None
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_root_counts(self._h, _p(counts), _p(cw), _p(cp), _stream()))
        return (counts, cw, cp) if with_children else counts

    def root_policy(self, temperature_zero=False):
        pi = torch.empty((self.G, self.A), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_root_policy(self._h, int(bool(temperature_zero)), _p(pi), _stream()))
        return pi

    def root_stats(self):
        visits = torch.empty(self.G, dtype=torch.int32, device=self.device)
        wsum = torch.empty(self.G, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_root_stats(self._h, _p(visits), _p(wsum), _stream()))
        return visits, wsum

    def boards(self):
        out = torch.empty((self.G, self.R, self.C), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_get_boards(self._h, _p(out), _stream()))
        return out

    def status(self):
        """sync; raises YYError(YY_E_ARENA) when a game's arena overflowed; returns the counters."""
        n = ct.c_int32(0)
        ctr = (ct.c_uint64 * 8)()
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_status(self._h, ct.byref(n), ctr))
        return dict(zip(self.COUNTERS, [int(x) for x in ctr[:8]]))

    def set_book(self, book):
        """Shallow leaves are looked up in `book` (OpeningBook, or None to unset) before the evaluator is asked; the context
        keeps a reference so that the table outlives its use."""
        if book is not None and (book.R, book.C) != (self.R, self.C):
            raise _lib.YYError(-1, "the book was built for another board size")
        with torch.cuda.device(self.device):
            if book is None:
                check(lib().yy_mcts_set_book(self._h, None, None, None, None, 0, 0))
            else:
                check(lib().yy_mcts_set_book(self._h, _p(book.meta), _p(book.table_keys), _p(book.value), _p(book.policy),
                                             book.cap, book.max_stones))
        self.book = book
        self.book_version = getattr(self, "book_version", 0) + 1      # captured steps hold the table's pointers: LockstepSearch re-captures

    def bind_evaluator(self, evaluator):
        """Evaluations kept across searches (keep_evaluations) and the opening book are results of ONE network: the engine-level
        searches (self_play.LockstepSearch: SelfPlayEngine, MCTS, Arena) bind the context to their evaluator object; a search with
        another one clears the cache and drops the book instead of silently mixing two networks' numbers.  The low-level
        search() below takes any callable per call and does not bind: there clear_evaluation_cache() is the caller's job."""
        owner = getattr(self, "_evaluator_owner", None)
        if owner is not None and owner is not evaluator:
            if self.keep_evaluations:
                self.clear_evaluation_cache()
            if getattr(self, "book", None) is not None:
                self.set_book(None)
        self._evaluator_owner = evaluator

    def clear_evaluation_cache(self):
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_cache_clear(self._h, _stream()))

    def reset_counters(self):
        with torch.cuda.device(self.device):
            check(lib().yy_mcts_reset_counters(self._h, _stream()))

    # -- the whole of MCTS.search for G games (mcts.py:275-343)
    def search(self, boards, root_players, evaluator, num_sims, noise=None, eps=0.25, active=None, fused=True):
        """evaluator(planes f32[G,5,R,C]) -> (policy f32[G,A] softmax, value f32[G]) on device.
        Runs 1 + num_sims evaluator calls exactly like the reference (root call, then one per
        simulation) and returns the root visit counts int32 [G,A]."""
        if num_sims > self.max_sims:
            raise _lib.YYError(-1, f"num_sims {num_sims} > max_sims {self.max_sims} the context was sized for")
        self.begin(boards, root_players, active)
        policy, _ = evaluator(self.planes)
        self.expand_root(policy, noise, eps)
        self.select()
        for s in range(num_sims):
            policy, value = evaluator(self.planes)
            if fused and s + 1 < num_sims:
                self.step(policy, value)
            else:
                self.expand_backup(policy, value)
                if s + 1 < num_sims:
                    self.select()
        return self.root_counts()

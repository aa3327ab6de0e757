"""Policy/value CNN of the reference under PyTorch-ROCm, plus the batched evaluator the lockstep
engine feeds.

`YinYangNeuralNetwork` mirrors src/yin_yang/ai/neural_network.py:35-237 (same constructor, same
parameter names so reference checkpoints load unchanged, same predict/board_to_input/save/load
surface).  The architecture is the reference's: 5 input planes -> 3x3 stem -> N residual blocks of
two 3x3 convs -> policy head (1x1 to 32 ch, FC to A) and value head (1x1 to 32 ch, FC 256, FC 1,
tanh).  What is new here is the batched path: `predict_batch` / `BatchedEvaluator` evaluate all G
leaf positions of a lockstep step in ONE forward on the device (the reference evaluates one board
per call on the CPU, self_play.py:54-59).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

H3_BOARDS = ((6, 6), (8, 8), (12, 12))      # board shapes the 32x32x16 split-f16 kernels cover (csrc/yy_tower_h3r.hip: A/B partner of the general kernel)
G_MAX_CELLS = 144                # the general split-f16 tower (csrc/yy_tower_g.hip): any R x C board up to this many cells
G_CHANNELS = (32, 64, 96, 128)   # ... and these widths
G_SPLIT_WG = 256                 # live rows <= this many workgroups of the small form (= ONE round of workgroups on the 256 CUs): the small
                                 # form runs, else the large one.  One launch alone on the chip (tools/tower_rows_sweep.py, 8x8): one-board form
                                 # 209 us up to 256 rows, 415 up to 512; two-board form 347 us up to 512 rows.  With two lanes on the
                                 # evaluation-reuse leg at 4096 games (bench.py --split-wg, one box; rows per launch rise from ~12 to ~800 inside a
                                 # search, tools/reuse_rows_hist.py): 192 -> 6 781 positions/s, 256 -> 7 118, 320 -> 7 073, 384 -> 7 054,
                                 # 512 -> 6 990, 640 -> 6 846
G_HINT_BIG_ONLY = 4                # an owner whose last move averaged >= this many times the split per step launches the large form only
                                 # (config 2 to completion with the engine's defaults, one box: 2 -> 6 284 / 6 334 positions/s, 4 -> 6 474, 8 -> 6 460)
G_AUTO_MAX_WG = 2048             # batches up to this many small-form workgroups enqueue both forms, gated on the device-side row count
HEAD_CHANNELS = 32
VALUE_HIDDEN = 256
INPUT_PLANES = 5


class ResidualBlock(nn.Module):
    """conv-bn-relu-conv-bn + skip, relu (neural_network.py:16-33); attribute names fixed by the
    checkpoint format."""

    def __init__(self, channels):
        super().__init__()
        for i in (1, 2):
            setattr(self, f"conv{i}", nn.Conv2d(channels, channels, 3, padding=1))
            setattr(self, f"bn{i}", nn.BatchNorm2d(channels))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class YinYangNeuralNetwork(nn.Module):
    def __init__(self, game, num_channels=128, num_res_blocks=10):
        super().__init__()
        self.game = game
        self.board_size = tuple(game.getBoardSize())
        self.action_size = game.getActionSize()
        self.input_channels = INPUT_PLANES
        cells = self.board_size[0] * self.board_size[1]
        self.conv1 = nn.Conv2d(INPUT_PLANES, num_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_channels)
        self.res_blocks = nn.ModuleList(ResidualBlock(num_channels) for _ in range(num_res_blocks))
        self.policy_conv = nn.Conv2d(num_channels, HEAD_CHANNELS, 1)
        self.policy_bn = nn.BatchNorm2d(HEAD_CHANNELS)
        self.policy_fc = nn.Linear(HEAD_CHANNELS * cells, self.action_size)
        self.value_conv = nn.Conv2d(num_channels, HEAD_CHANNELS, 1)
        self.value_bn = nn.BatchNorm2d(HEAD_CHANNELS)
        self.value_fc1 = nn.Linear(HEAD_CHANNELS * cells, VALUE_HIDDEN)
        self.value_fc2 = nn.Linear(VALUE_HIDDEN, 1)
        # Xavier-normal weights, zero biases (neural_network.py:85-92); module order == reference
        # order so torch.manual_seed(s) gives the same initial weights as the reference
        for mod in self.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_normal_(mod.weight)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)

    # ---- forward: logits + tanh value (neural_network.py:94-123)
    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        for blk in self.res_blocks:
            x = blk(x)
        p = F.relu(self.policy_bn(self.policy_conv(x))).flatten(1)
        v = F.relu(self.value_bn(self.value_conv(x))).flatten(1)
        return self.policy_fc(p), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))

    # ---- batched evaluation on the module's device
    @torch.no_grad()
    def predict_batch(self, planes):
        """planes float32 [G,5,R,C] on the module's device -> (softmax policy f32 [G,A], value f32 [G])."""
        self.eval()
        logits, value = self.forward(planes)
        return F.softmax(logits, dim=1), value.reshape(-1)

    # ---- single-board API of the reference (neural_network.py:125-154)
    def predict(self, board):
        dev = next(self.parameters()).device
        x = self.board_to_input(board).unsqueeze(0).to(dev)
        pol, val = self.predict_batch(x)
        return pol[0].cpu().numpy(), val.cpu().numpy()[0]

    def board_to_input(self, board):
        """5 planes float32 (neural_network.py:156-196).  On a ROCm device this is the HIP encode
        kernel; for a CPU module the planes are computed with torch ops on the host."""
        arr = np.ascontiguousarray(board.get_board(), dtype=np.int8)
        dev = next(self.parameters()).device
        if dev.type == "cuda":
            from . import engine
            return engine.encode_planes(torch.from_numpy(arr[None]).to(dev))[0]
        b = torch.from_numpy(arr)
        n, m = arr.shape
        occ = (b != 0)
        rows = (occ.sum(1).double() / m).float()[:, None].expand(n, m)
        cols = (occ.sum(0).double() / n).float()[None, :].expand(n, m)
        return torch.stack([(b == 0).float(), (b == 1).float(), (b == -1).float(), rows, cols])

    # ---- checkpoint format of the reference (neural_network.py:198-237)
    def save_model(self, filename):
        d = os.path.dirname(filename)
        if d:
            os.makedirs(d, exist_ok=True)
        torch.save({"state_dict": self.state_dict(), "board_size": self.board_size,
                    "action_size": self.action_size}, filename)

    def load_model(self, filename):
        if not os.path.exists(filename):
            raise FileNotFoundError(f"Model file {filename} not found")
        ckpt = torch.load(filename, map_location="cpu", weights_only=True)
        self.load_state_dict(ckpt["state_dict"])


def fold_batchnorm(conv, bn):
    """(weight, bias) of the conv with the eval-mode BatchNorm folded in."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * scale.reshape(-1, 1, 1, 1)
    b = (conv.bias - bn.running_mean) * scale + bn.bias
    return w.detach(), b.detach()


def pack_tower(net):
    """Fold eval-mode BatchNorm and pack the stem + residual-block convolutions of `net` (8x8 boards,
    128 channels) into the fragment order csrc/yy_tower.hip streams:
    weights int16(bf16 bits) [n_chunks, 8192] with chunk = [ks 4][ntile 4][h 2][c 32][j 8],
    cout = ntile*32 + c, cin = half*64 + ks*16 + h*8 + j; layer 0 (stem) has 9 chunks (one per tap,
    5 input planes zero-padded to 16 channels), every other layer 18 (tap-major, then half);
    bias float32 [n_layers, 128]."""
    convs = [(net.conv1, net.bn1)]
    for blk in net.res_blocks:
        convs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
    chunks, biases = [], []
    for li, (conv, bn) in enumerate(convs):
        w, b = fold_batchnorm(conv, bn)                       # [128, cin, 3, 3], [128]
        w = w.float().cpu()
        cout, cin = w.shape[0], w.shape[1]
        assert cout == 128 and w.shape[2:] == (3, 3) and (cin == 128 or li == 0)
        wp = torch.zeros((128, 128, 3, 3))
        wp[:, :cin] = w
        for tap in range(9):
            wt = wp[:, :, tap // 3, tap % 3]                  # [cout, cin]
            t = wt.reshape(4, 32, 2, 4, 2, 8)                 # nt, c, half, ks, h, j
            t = t.permute(2, 3, 0, 4, 1, 5).contiguous()      # half, ks, nt, h, c, j
            for half in range(1 if li == 0 else 2):
                chunks.append(t[half].reshape(-1))
        biases.append(b.float().cpu())
    wq = torch.stack(chunks).to(torch.bfloat16).view(torch.int16).contiguous()
    return wq, torch.stack(biases).contiguous()


def pack_tower_f32(net):
    """float32 packing for csrc/yy_tower_f32.hip: chunk = one tap x 32 input channels = [m 4][nt 4][h 2][i 32][s 4]
    float32 with cout = nt*32 + i and cin = quarter*32 + m*8 + h*4 + s; the stem has one chunk per tap (5 planes padded to
    8 channels, m = 0 only), every other layer 36 (tap-major, then quarter)."""
    convs = [(net.conv1, net.bn1)]
    for blk in net.res_blocks:
        convs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
    chunks, biases = [], []
    for li, (conv, bn) in enumerate(convs):
        w, b = fold_batchnorm(conv, bn)
        w = w.float().cpu()
        wp = torch.zeros((128, 128, 3, 3))
        wp[:, :w.shape[1]] = w
        for tap in range(9):
            t = wp[:, :, tap // 3, tap % 3].reshape(4, 32, 4, 4, 2, 4)      # nt, i, quarter, m, h, s
            t = t.permute(2, 3, 0, 4, 1, 5).contiguous()                   # quarter, m, nt, h, i, s
            for quarter in range(1 if li == 0 else 4):
                chunks.append(t[quarter].reshape(-1))
        biases.append(b.float().cpu())
    return torch.stack(chunks).contiguous(), torch.stack(biases).contiguous()


ACT_EXP = 3          # split-f16 tower: activations (and the tower's bias rows) live times 2^ACT_EXP (csrc/yy_tower_g.hip)


def _pow2_exponent(t, target=14):
    """k such that max|t| * 2^k lies in [2^(target-1), 2^target): an exact scaling that keeps the float16 lo parts normal."""
    m = float(t.abs().max())
    if not np.isfinite(m) or m <= 0.0:
        return 0
    return int(target - 1 - np.floor(np.log2(m)))


def split_f16(t):
    """x -> (hi, lo) float16 with x == hi + lo to 22 significant bits (csrc/yy_tower_g.hip): hi = f16(x), lo = f16(x - hi)."""
    t = t.float()
    hi = t.to(torch.float16)
    if not bool(torch.isfinite(hi).all()):
        raise ValueError("split-f16 evaluator: a scaled weight exceeds the float16 range")
    lo = (t - hi.float()).to(torch.float16)
    return hi, lo


def pack_tower_h3(net):
    """Split-f16 packing for the 32x32x16 kernel (csrc/yy_tower_h3r.hip, evaluator mode "f16x3r"; re-ordered by pack_tower_h3r): every folded float32 weight w, times 2^kw, becomes
    hi = f16(w'), lo = f16(w' - hi); chunk = one tap x 32 input channels = [ks 2][part 2][nt 4][h 2][c 32][j 8] f16 with
    cout = nt*32 + c and cin = quarter*32 + ks*16 + h*8 + j (part 0 = hi, 1 = lo); the stem has one chunk per tap (5 planes
    padded to 16 channels, ks = 0 only), every other layer 36 (tap-major, then quarter).
    Returns (int16 [n_chunks, 8192] f16 bits, float32 bias [n_layers, 128] times 2^ACT_EXP, kw)."""
    convs = [(net.conv1, net.bn1)]
    for blk in net.res_blocks:
        convs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
    folded = [fold_batchnorm(conv, bn) for conv, bn in convs]
    kw = _pow2_exponent(torch.cat([w.float().reshape(-1) for w, _ in folded]))
    chunks, biases = [], []
    for li, (w, b) in enumerate(folded):
        w = w.float().cpu()
        wp = torch.zeros((128, 128, 3, 3))
        wp[:, :w.shape[1]] = w
        hi, lo = split_f16(torch.ldexp(wp, torch.tensor(kw)))
        for tap in range(9):
            parts = []
            for t in (hi, lo):
                t = t[:, :, tap // 3, tap % 3].reshape(4, 32, 4, 2, 2, 8)         # nt, c, quarter, ks, h, j
                parts.append(t.permute(2, 3, 0, 4, 1, 5))                          # quarter, ks, nt, h, c, j
            both = torch.stack(parts, dim=2).contiguous()                          # quarter, ks, part, nt, h, c, j
            for quarter in range(1 if li == 0 else 4):
                chunks.append(both[quarter].reshape(-1))
        biases.append(torch.ldexp(b.float().cpu(), torch.tensor(ACT_EXP)))
    return torch.stack(chunks).view(torch.int16).contiguous(), torch.stack(biases).contiguous(), kw


def pack_heads_h3(net):
    """The two 1x1 head convolutions for the split-f16 kernels: two chunks [ks 4][part 2][nt 2][h 2][c 32][j 8] f16 of the
    weights times 2^kh (nt 0 = policy channels, nt 1 = value channels, cin = chunk*64 + ks*16 + h*8 + j) and one UNSCALED bias
    row [policy 32 | value 32 | zeros].  Returns (int16 [2, 8192], float32 [1, 128], kh)."""
    wp, bp = fold_batchnorm(net.policy_conv, net.policy_bn)      # [32,128,1,1]
    wv, bv = fold_batchnorm(net.value_conv, net.value_bn)
    w = torch.cat([wp, wv]).float().cpu()
    kh = _pow2_exponent(w)
    w = torch.ldexp(w, torch.tensor(kh)).reshape(2, 32, 2, 4, 2, 8)              # nt, c, chunk, ks, h, j
    parts = [t.permute(2, 3, 0, 4, 1, 5) for t in split_f16(w)]                  # chunk, ks, nt, h, c, j
    both = torch.stack(parts, dim=2).contiguous()                                # chunk, ks, part, nt, h, c, j
    bias = torch.zeros(1, 128)
    bias[0, :32], bias[0, 32:64] = bp.float().cpu(), bv.float().cpu()
    return both.reshape(2, -1).view(torch.int16).contiguous(), bias, kh


def pack_tower_h3r(net):
    """pack_tower_h3 re-ordered "wave-major" for csrc/yy_tower_h3r.hip (weights loaded global -> VGPR by the wave that uses
    them): chunk = [nt 4][ks 2][part 2][h 2][c 32][j 8] f16, so that the 4 KB of output-channel quarter nt are contiguous."""
    wq, bq, kw = pack_tower_h3(net)
    n = wq.shape[0]
    return wq.view(n, 2, 2, 4, 512).permute(0, 3, 1, 2, 4).contiguous().view(n, 8192), bq, kw


def pack_heads_h3r(net):
    """pack_heads_h3 re-ordered for csrc/yy_tower_h3r.hip: [head 2][ks 8][part 2][h 2][c 32][j 8] f16 (cin = ks*16 + h*8 + j)."""
    hw, hb, kh = pack_heads_h3(net)                                  # [chunk 2][ks 4][part 2][nt 2][512]
    return hw.view(2, 4, 2, 2, 512).permute(3, 0, 1, 2, 4).contiguous().view(2, 8192), hb, kh


def tower_convs(net):
    convs = [(net.conv1, net.bn1)]
    for blk in net.res_blocks:
        convs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
    return convs


def pack_tower_g(net):
    """Split-f16 packing for csrc/yy_tower_g.hip (v_mfma_f32_16x16x32_f16; any board, 32/64/96/128 channels): every folded
    float32 weight w, times 2^kw, becomes hi = f16(w'), lo = f16(w' - hi).  A chunk is one tap x 32 input channels for all
    output channels = [wave CH/32][M block 2][part 2][lane 64][j 8] f16 with cout = wave*32 + mblock*16 + lane%16 and
    cin = kq*32 + (lane//16)*8 + j (part 0 = hi, 1 = lo): the A operand of the MFMA in register order, 4 KB per wave.
    Chunk order [layer][kq][tap]; the stem has kq = 0 only (5 planes zero-padded to 32 channels).
    Returns (int16 [n_chunks, CH*64] f16 bits, float32 bias [n_layers, CH] times 2^ACT_EXP, kw)."""
    ch = net.conv1.out_channels
    if ch % 32 or not 32 <= ch <= 128:
        raise ValueError("split-f16 tower: channels must be 32, 64, 96 or 128")
    nw = ch // 32
    folded = [fold_batchnorm(conv, bn) for conv, bn in tower_convs(net)]
    kw = _pow2_exponent(torch.cat([w.float().reshape(-1) for w, _ in folded]))
    chunks, biases = [], []
    for li, (w, b) in enumerate(folded):
        w = w.float().cpu()
        cin = w.shape[1]
        kq_n = 1 if li == 0 else nw
        wp = torch.zeros((ch, 32 * kq_n, 3, 3))
        wp[:, :cin] = w
        hi, lo = split_f16(torch.ldexp(wp, torch.tensor(kw)))
        both = torch.stack([hi, lo])                                     # part, cout, cin, ky, kx
        both = both.reshape(2, nw, 2, 16, kq_n, 4, 8, 9)                  # part, wave, mb, m, kq, kg, j, tap
        both = both.permute(4, 7, 1, 2, 0, 5, 3, 6).contiguous()          # kq, tap, wave, mb, part, kg, m, j
        chunks.append(both.reshape(kq_n * 9, -1))
        biases.append(torch.ldexp(b.float().cpu(), torch.tensor(ACT_EXP)))
    return torch.cat(chunks).view(torch.int16).contiguous(), torch.stack(biases).contiguous(), kw


def pack_heads_g(net):
    """The two 1x1 head convolutions for csrc/yy_tower_g.hip: [unit 4 = head*2 + M block][kq CH/32][part 2][lane 64][j 8] f16 of the
    folded weights times 2^kh (channel of the head = mblock*16 + lane%16, cin = kq*32 + (lane//16)*8 + j) and the UNSCALED bias
    [policy 32 | value 32].  Returns (int16 [4*CH/32*2*512], float32 [64], kh)."""
    wp, bp = fold_batchnorm(net.policy_conv, net.policy_bn)      # [32,CH,1,1]
    wv, bv = fold_batchnorm(net.value_conv, net.value_bn)
    if wp.shape[0] != HEAD_CHANNELS or wv.shape[0] != HEAD_CHANNELS:
        raise ValueError("split-f16 tower: the head convolutions must have 32 channels")
    w = torch.cat([wp, wv]).float().cpu().reshape(2 * HEAD_CHANNELS, -1)
    nw = w.shape[1] // 32
    kh = _pow2_exponent(w)
    hi, lo = split_f16(torch.ldexp(w, torch.tensor(kh)))
    both = torch.stack([hi, lo]).reshape(2, 2, 2, 16, nw, 4, 8)         # part, head, mb, m, kq, kg, j
    both = both.permute(1, 2, 4, 0, 5, 3, 6).contiguous()               # head, mb, kq, part, kg, m, j
    return both.reshape(-1).view(torch.int16).contiguous(), torch.cat([bp, bv]).float().cpu().contiguous(), kh


def pack_fc_heads(net):
    """policy_fc and value_fc1 for csrc/yy_fc_heads.hip: the outputs of each head are cut into slices of at most 64 (a job);
    a slice's weights, times 2^kw and split into hi / lo float16, are stored [k-step ceil(K/128)*4][wave 4][part 2][lane 64][j 8]
    with output = first + wave*16 + lane%16 and k = kstep*32 + (lane//16)*8 + j (zero beyond the slice / beyond K).
    Returns (int16 weights, float32 bias [A + H], int32 jobs [n_jobs, 4] = (head, first output, outputs, first 8 KB block), kw)."""
    mats = [net.policy_fc.weight.detach().float().cpu(), net.value_fc1.weight.detach().float().cpu()]
    K = mats[0].shape[1]
    ksteps = ((K + 127) // 128) * 4
    kw = _pow2_exponent(torch.cat([m.reshape(-1) for m in mats]))
    blocks, jobs = [], []
    for head, w in enumerate(mats):
        for first in range(0, w.shape[0], 64):
            n = min(64, w.shape[0] - first)
            wp = torch.zeros((64, ksteps * 32))
            wp[:n, :K] = w[first:first + n]
            hi, lo = split_f16(torch.ldexp(wp, torch.tensor(kw)))
            both = torch.stack([hi, lo]).reshape(2, 4, 16, ksteps, 4, 8)            # part, wave, m, kstep, kg, j
            blocks.append(both.permute(3, 1, 0, 4, 2, 5).contiguous().reshape(-1))   # kstep, wave, part, kg, m, j
            jobs.append((head, first, n, len(jobs) * ksteps))
    bias = torch.cat([net.policy_fc.bias.detach().float().cpu(), net.value_fc1.bias.detach().float().cpu()]).contiguous()
    return torch.cat(blocks).view(torch.int16).contiguous(), bias, torch.tensor(jobs, dtype=torch.int32), kw


def tower_g_forms(cells, channels):
    """(column blocks, boards per workgroup) choices of the general split-f16 tower for a board of `cells` cells:
    (large-batch form, small-batch form).  A form covers 16 * nb (board, cell) columns; boards * cells of them are real."""
    from . import engine
    forms = engine.tower_g_available(channels)
    best = small = None
    for nb in forms:
        tb = (16 * nb) // cells
        if tb < 1:
            continue
        eff = tb * cells / (16.0 * nb)
        if best is None or (eff, nb) > (best[2], best[0]):      # least padding; ties: the larger tile (more weight reuse)
            best = (nb, tb, eff)
        if small is None or (tb, nb) < (small[1], small[0]):   # fewest boards per workgroup: most workgroups for few rows
            small = (nb, tb, eff)
    if best is None:
        raise ValueError("split-f16 tower: no kernel form for %d cells" % cells)
    return best[:2], small[:2]


def pack_heads(net):
    """The two 1x1 head convolutions (policy_conv/policy_bn, value_conv/value_bn) as one extra chunk
    [ks 8][nt 2][h 2][c 32][j 8] (nt 0 = policy channels, nt 1 = value channels, cin = ks*16 + h*8 + j)
    and one extra bias row [policy 32 | value 32 | zeros]."""
    wp, bp = fold_batchnorm(net.policy_conv, net.policy_bn)      # [32,128,1,1]
    wv, bv = fold_batchnorm(net.value_conv, net.value_bn)
    w = torch.cat([wp, wv]).float().cpu().reshape(2, 32, 8, 2, 8)   # nt, c, ks, h, j
    chunk = w.permute(2, 0, 3, 1, 4).contiguous().reshape(1, -1)    # ks, nt, h, c, j
    bias = torch.zeros(1, 128)
    bias[0, :32], bias[0, 32:64] = bp.float().cpu(), bv.float().cpu()
    return chunk.to(torch.bfloat16).view(torch.int16).contiguous(), bias


def reference_precision_mode(net):
    """The fastest evaluator mode that is float32-ACCURATE for this network (the reference evaluates in float32,
    neural_network.py:125-154): "f16x3" where the split-f16 kernels cover the shape (f16x3_covers), else "fp32" (the module itself)."""
    return "f16x3" if f16x3_covers(net) else "fp32"


def f16x3_covers(net):
    """True where the split-f16 kernels (csrc/yy_tower_g.hip + yy_fc_heads.hip) cover the network: any board of at most 144 cells,
    32 / 64 / 96 / 128 channels, at most 10 residual blocks, 32-channel head convolutions.
    Range: activations live times 2^ACT_EXP in float16 pairs, so an activation (or head feature) above 65504 / 2^ACT_EXP = 8188
    becomes inf, then NaN; the tree kernel turns a NaN prior / value into the game's sticky error flag and the engine raises at
    the end of that MOVE (SelfPlayEngine.finish_move reads the context's status once per move).  Networks of this architecture
    (BatchNorm after every convolution) stay orders of magnitude below it: the largest activation over the round's soaks was
    ~10; a network that does not should be evaluated with mode "fp32"."""
    R, C = net.board_size
    return (R * C <= G_MAX_CELLS and net.conv1.out_channels in G_CHANNELS and len(net.res_blocks) <= 10
            and net.policy_conv.out_channels == HEAD_CHANNELS and net.value_conv.out_channels == HEAD_CHANNELS)


class BatchedEvaluator:
    """Callable evaluator for BatchedMCTS.search: planes f32 [G,5,R,C] -> (policy f32 [G,A], value f32 [G]).

    mode "auto" (default): `reference_precision_mode(net)` -- "f16x3" where the kernels cover the shape, else "fp32".
    mode "f16x3": float32 ACCURACY on the f16 matrix cores (csrc/yy_tower_g.hip + yy_fc_heads.hip): activations and weights as
    hi + lo float16 pairs (22 significant bits), three MFMAs per product term, f32 accumulation / bias / residual, 1x1 head
    convs fused into the tower kernel, FC heads as one split-f16 GEMM kernel of our own + one finish kernel.  Any board of at
    most 144 cells, 32 / 64 / 96 / 128 channels.  Searches driven by it return the reference's visit counts
    (tests/test_gpu_mcts.py::test_live_gpu_evaluator_search_vs_reference_pi); supports row compaction; a row's results do not
    depend on the batch (row_independent).
    mode "f16x3r": the same evaluator with the round-2 32x32x16 tower kernel (csrc/yy_tower_h3r.hip; 6x6 / 8x8 / 12x12, 128
    channels) in place of the general one: the A/B partner for timing.
    mode "fp32": the module as is (same arithmetic as predict()).
    mode "fp32t": the same float32 weights, but the stem + residual tower run in the hand-written exact-f32 MFMA kernel
    (csrc/yy_tower_f32.hip; 8x8 boards, 128 channels); heads by torch in float32.  Differs from "fp32" only by summation order.
    mode "bf16"/"fp16": inference-only fast path -- eval-mode BatchNorm folded into the convs,
    channels-last activations, reduced-precision MFMA convolutions (MIOpen implicit GEMM), softmax/tanh
    in fp32.  In bf16 mode with `fused_epilogue` (default) every convolution is issued WITHOUT bias and
    followed by ONE hand-written HIP pass (csrc k_bias_act: bias + residual + ReLU in place) instead of
    the 3-4 elementwise kernels PyTorch would launch.
    """

    def __init__(self, net, mode="auto", fused_epilogue=True, tower=True, fused_heads=True):
        self.net = net.eval()
        if mode == "auto":            # reference precision, fastest available kernel
            mode = reference_precision_mode(net)
        self.mode = mode
        self.device = next(net.parameters()).device
        self.fused = bool(fused_epilogue) and mode == "bf16"
        # the LDS-resident MFMA tower kernels (csrc/yy_tower.hip 8x8, yy_towerq.hip 6x6 / 12x12 / small 8x8 batches) cover the stem + residual
        # blocks (+ head convs) for 128 channels; other shapes use MIOpen convolutions + the fused epilogue
        self.tower = (bool(tower) and mode == "bf16" and tuple(net.board_size) in ((6, 6), (8, 8), (12, 12))
                      and net.conv1.out_channels == 128 and len(net.res_blocks) <= 10)
        if mode in ("f16x3", "f16x3r"):
            if not f16x3_covers(net):
                raise ValueError("f16x3 needs a board of at most 144 cells, 32/64/96/128 channels, at most 10 residual blocks, 32-channel heads")
            dev = self.device
            R, C = net.board_size
            self.h3_layers = 1 + 2 * len(net.res_blocks)
            # the general kernel (csrc/yy_tower_g.hip): weights in MFMA operand order, the kernel forms for this board
            wq, bq, kw = pack_tower_g(net)
            hw, hb, kh = pack_heads_g(net)
            self.g_w, self.g_b, self.g_hw, self.g_hb = wq.to(dev), bq.to(dev), hw.to(dev), hb.to(dev)
            self.g_exps = (kw, kh, ACT_EXP)       # weights x 2^kw, head weights x 2^kh, activations x 2^ACT_EXP
            self.g_big, self.g_small = tower_g_forms(R * C, net.conv1.out_channels)
            self.g_split = G_SPLIT_WG * self.g_small[1]
            # policy_fc + value_fc1 as one split-f16 GEMM kernel (csrc/yy_fc_heads.hip), fixed summation order per output
            fw, fb, jobs, kf = pack_fc_heads(net)
            self.fc_w, self.fc_b, self.fc_jobs, self.fc_exps = fw.to(dev), fb.to(dev), jobs.to(dev), (kf, ACT_EXP)
            self.n_actions, self.n_hidden = net.policy_fc.out_features, net.value_fc1.out_features
            f32 = lambda t: t.detach().float().contiguous().to(dev)
            self.fc2_w, self.fc2_b = f32(net.value_fc2.weight.reshape(-1)), f32(net.value_fc2.bias.reshape(1))
            self.use_h3r = False
            if mode == "f16x3r":                  # A/B partner: the 32x32x16 register-ring tower of round 2 (6x6 / 8x8 / 12x12, 128 channels)
                if tuple(net.board_size) not in H3_BOARDS or net.conv1.out_channels != 128:
                    raise ValueError("f16x3r needs 6x6, 8x8 or 12x12 boards and 128 channels")
                self.h3_b = torch.cat([pack_tower_h3(net)[1], pack_heads_h3(net)[1]]).contiguous().to(dev)
                self.h3r_w = pack_tower_h3r(net)[0].to(dev)
                self.h3r_hw = pack_heads_h3r(net)[0].to(dev)
                kw_r, kh_r = pack_tower_h3(net)[2], pack_heads_h3(net)[2]
                self.h3_exps = (kw_r, kh_r, ACT_EXP)
                self.use_h3r = True
                self.mode = mode = "f16x3"
            self._hint = {}                      # owner token -> live rows of that owner's last compacted launch (rows_hint())
            self.supports_compaction = True      # __call__(planes, needs_eval=...) evaluates only the flagged rows
            self.supports_static = True          # __call__(..., static=True): results in buffers kept per batch size
            self._static = {}
            self.tower = False
            # A row's (policy, value) is a function of that row's planes alone, bit for bit, BY CONSTRUCTION: the tower computes
            # each board in a fixed order whatever its workgroup form or position in the batch, and the FC heads are our own
            # fixed-order kernel (no library GEMM whose tiling could follow the batch size).  Pinned by
            # tests/test_gpu_network.py::test_evaluator_rows_do_not_depend_on_the_batch.  The search may therefore reuse
            # evaluations across launches of different sizes (YY_FLAG_REUSE_*, the opening book).
            self.row_independent = True
            return
        if mode == "fp32t":
            if tuple(net.board_size) != (8, 8) or net.conv1.out_channels != 128 or len(net.res_blocks) > 11:
                raise ValueError(mode + " needs 8x8 boards, 128 channels, at most 11 residual blocks")
            wq, bq = pack_tower_f32(net)
            self.f32_w, self.f32_b = wq.to(self.device), bq.to(self.device)
            self.f32_layers = 1 + 2 * len(net.res_blocks)
            # float32 heads in four GEMM-shaped steps: both 1x1 head convs as ONE [G*64,128] x [128,64] product, then
            # policy_fc and value_fc1 as ONE product over [policy features | value features] (block weights)
            (pw, pb), (vw, vb) = fold_batchnorm(net.policy_conv, net.policy_bn), fold_batchnorm(net.value_conv, net.value_bn)
            self.hconv_w = torch.cat([pw, vw]).float().reshape(pw.shape[0] + vw.shape[0], -1).t().contiguous().to(self.device)
            self.hconv_b = torch.cat([pb, vb]).float().contiguous().to(self.device)
            A, F_, Hd = net.policy_fc.out_features, net.policy_fc.in_features, net.value_fc1.out_features
            wc = torch.zeros((A + Hd, 2 * F_), dtype=torch.float32, device=self.device)
            wc[:A, :F_] = net.policy_fc.weight.detach()
            wc[A:, F_:] = net.value_fc1.weight.detach()
            self.fc_cat_w = wc.contiguous()
            self.fc_cat_b = torch.cat([net.policy_fc.bias.detach(), net.value_fc1.bias.detach()]).float().contiguous()
            self.fc2_w = net.value_fc2.weight.detach().float().reshape(-1, 1).contiguous()
            self.fc2_b = net.value_fc2.bias.detach().float().reshape(1).contiguous()
            self.n_actions = A
            self.tower = False
            return
        if mode != "fp32":
            self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[mode]
            self._fold()
        if self.tower:
            wq, bq = pack_tower(net)
            self.tower_w, self.tower_b = wq.to(self.device), bq.to(self.device)
            self.tower_layers = 1 + 2 * len(net.res_blocks)
            self.fused_heads = bool(fused_heads) and net.policy_conv.out_channels == 32
            if self.fused_heads:
                hw, hb = pack_heads(net)
                self.towerh_w = torch.cat([wq, hw]).contiguous().to(self.device)
                self.towerh_b = torch.cat([bq, hb]).contiguous().to(self.device)
                # policy_fc and value_fc1 as ONE GEMM over [policy features | value features] (block weights)
                A, F_ = net.policy_fc.out_features, net.policy_fc.in_features
                Hd = net.value_fc1.out_features
                wc = torch.zeros((A + Hd, 2 * F_), dtype=torch.float32, device=self.device)
                wc[:A, :F_] = net.policy_fc.weight.detach()
                wc[A:, F_:] = net.value_fc1.weight.detach()
                self.fc_cat_w = wc.to(self.dtype).contiguous()
                self.fc_cat_b = torch.cat([net.policy_fc.bias.detach(), net.value_fc1.bias.detach()]).to(self.dtype).contiguous()
                self.fc2_w = net.value_fc2.weight.detach().float().reshape(-1).contiguous()
                self.fc2_b = net.value_fc2.bias.detach().float().reshape(1).contiguous()
                self.n_actions = A

    def form_key(self, owner):
        """Which tower form(s) a compacted call of this owner would enqueue now: part of the key of a captured step."""
        hint = getattr(self, "_hint", {}).get(owner)
        if hint is None or not hasattr(self, "g_split"):
            return 0
        return 1 if hint >= 2 * self.g_split else 0

    def rows_hint(self, owner, mean_rows):
        """Tell the evaluator how many rows per step `owner`'s last move evaluated on average (the engine knows it from the tree
        context's counters where it waits for the move anyway): decides the tower form of that owner's next launches."""
        if hasattr(self, "_hint"):
            self._hint[owner] = float(mean_rows)

    def _fold(self):
        n, dt = self.net, self.dtype
        cl = torch.channels_last

        def prep(conv, bn):
            w, b = fold_batchnorm(conv, bn)
            return w.to(dt).contiguous(memory_format=cl), (b.float().contiguous() if self.fused else b.to(dt))

        self.stem = prep(n.conv1, n.bn1)
        self.blocks = [(prep(blk.conv1, blk.bn1), prep(blk.conv2, blk.bn2)) for blk in n.res_blocks]
        self.phead = prep(n.policy_conv, n.policy_bn)
        self.vhead = prep(n.value_conv, n.value_bn)
        self.pfc = (n.policy_fc.weight.detach().to(dt), n.policy_fc.bias.detach().to(dt))
        self.vfc1 = (n.value_fc1.weight.detach().to(dt), n.value_fc1.bias.detach().to(dt))
        self.vfc2 = (n.value_fc2.weight.detach().float(), n.value_fc2.bias.detach().float())

    def _conv(self, x, wb, padding, residual=None):
        """relu(conv(x) + bias (+ residual)) on channels-last tensors."""
        w, b = wb
        if not self.fused:
            y = F.conv2d(x, w, b, padding=padding)
            return F.relu(y if residual is None else y + residual)
        from . import engine
        y = F.conv2d(x, w, None, padding=padding)
        return engine.bias_act_(y, b, residual, relu=True)

    @torch.no_grad()
    def __call__(self, planes, needs_eval=None, static=False):
        """needs_eval (uint8 [G], modes with `supports_compaction` only): evaluate just the flagged rows -- the tower launch
        gathers them, the other rows of the returned (policy, value) must not be read.
        static (f16x3): write the row list and the results into buffers the evaluator keeps per (batch size, owner) instead of fresh
        zero-filled tensors (four fill kernels per call less); the returned tensors are overwritten by the owner's next call.
        True = one shared owner; any other hashable value = that owner's private buffers (one per LockstepSearch: searches on
        different HIP streams must not share them -- a stream id would not do, graph captures share one capture stream)."""
        if self.mode == "f16x3":
            from . import engine
            G = planes.shape[0]
            K = 32 * planes.shape[2] * planes.shape[3]
            rows = n = pol = val = feats = logits = hidden = None
            if static:
                key = (G, 0 if static is True else static)             # static = an owner token: engines on different streams must not share buffers
                buf = self._static.get(key)
                if buf is None:
                    dev = planes.device
                    z = lambda shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
                    buf = self._static[key] = (z(G, torch.int32), z(1, torch.int32), z((G, self.n_actions)), z(G), z((G, 2, K)),
                                             z((G, self.n_actions)), z((G, self.n_hidden)))
                rows, n, pol, val, feats, logits, hidden = buf
            if needs_eval is not None:
                rows, n = engine.compact_rows(needs_eval, rows, n)
            else:
                rows = n = None
            if self.use_h3r:
                feats = engine.tower_heads_forward_h3r(planes, self.h3r_w, self.h3r_hw, self.h3_b, self.h3_layers, self.h3_exps, rows, n, feats)
            else:
                tg = lambda form, gate, out: engine.tower_g(planes, self.g_w, self.g_b, self.h3_layers, self.g_exps, form[0], form[1],
                                                            self.g_hw, self.g_hb, rows, n, out, gate)
                if self.g_small == self.g_big or G > G_AUTO_MAX_WG * self.g_small[1] or (rows is None and G > self.g_split):
                    feats = tg(self.g_big, (-1, 0x7FFFFFFF), feats)
                elif G <= self.g_split:
                    feats = tg(self.g_small, (-1, 0x7FFFFFFF), feats)
                else:
                    # The live row count is only known on the device: both forms are enqueued, each gated on it (same bits either
                    # way; the count swings widely between the steps of one search, and picking per step beats picking one form
                    # per move: 6 146 against 5 342 positions/s on the evaluation-reuse leg; ONE launch holding both forms was
                    # built and measured too: no gain over the two gated launches, 6 104 against 6 107).  Only when
                    # the owner's last move averaged far more rows per step than the split (rows_hint) is the small form's launch
                    # dropped: an unneeded gated launch is not free under lanes, its empty workgroups queue behind the other
                    # lane's running ones.
                    hint = self._hint.get(static) if static is not True and static is not False else None
                    if hint is not None and hint >= G_HINT_BIG_ONLY * self.g_split:
                        feats = tg(self.g_big, (-1, 0x7FFFFFFF), feats)
                    else:
                        feats = tg(self.g_small, (-1, self.g_split), feats)
                        feats = tg(self.g_big, (self.g_split, 0x7FFFFFFF), feats)
            logits, hidden = engine.fc_heads(feats, self.fc_w, self.fc_b, self.fc_jobs, self.n_actions, self.n_hidden, self.fc_exps, n,
                                             logits, hidden)
            return engine.head_finish_f32(logits, hidden, self.fc2_w, self.fc2_b, rows, n, pol, val)
        if needs_eval is not None:
            raise ValueError(f"evaluator mode {self.mode} does not take needs_eval")
        if self.mode == "fp32":
            return self.net.predict_batch(planes)
        if self.mode == "fp32t":
            from . import engine
            x = engine.tower_forward_f32(planes, self.f32_w, self.f32_b, self.f32_layers)           # NCHW view of [G,8,8,128] memory
            G = x.shape[0]
            cells = x.shape[2] * x.shape[3]
            hc = torch.addmm(self.hconv_b, x.permute(0, 2, 3, 1).reshape(G * cells, -1), self.hconv_w)   # [G*cells, 64]
            feats = torch.relu_(hc).view(G, cells, -1).transpose(1, 2).reshape(G, -1)     # [G, (head, ch, cell)] = NCHW flatten
            h = torch.addmm(self.fc_cat_b, feats, self.fc_cat_w.t())                      # [G, A + 256]
            A = self.n_actions
            value = torch.tanh(torch.addmm(self.fc2_b, torch.relu(h[:, A:]), self.fc2_w)).reshape(-1)
            return F.softmax(h[:, :A], dim=1), value
        if self.tower and self.fused_heads:
            from . import engine
            feats = engine.tower_heads_forward(planes, self.towerh_w, self.towerh_b, self.tower_layers)
            hcat = F.linear(feats.view(feats.shape[0], -1), self.fc_cat_w, self.fc_cat_b)    # [G, A + 256] bf16
            return engine.head_finish(hcat, self.n_actions, self.fc2_w, self.fc2_b)
        if self.tower:
            from . import engine
            x = engine.tower_forward(planes, self.tower_w, self.tower_b, self.tower_layers)
        else:
            x = planes.to(self.dtype).contiguous(memory_format=torch.channels_last)
            x = self._conv(x, self.stem, 1)
            for (c1, c2) in self.blocks:
                y = self._conv(x, c1, 1)
                x = self._conv(y, c2, 1, residual=x)
        p = self._conv(x, self.phead, 0).contiguous().flatten(1)      # NCHW flatten order as the reference
        v = self._conv(x, self.vhead, 0).contiguous().flatten(1)
        logits = F.linear(p, *self.pfc).float()
        h = F.relu(F.linear(v, *self.vfc1)).float()
        value = torch.tanh(F.linear(h, *self.vfc2)).reshape(-1)
        return F.softmax(logits, dim=1), value

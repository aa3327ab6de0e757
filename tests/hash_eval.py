"""Exact dyadic hash evaluator (test helper).  Same integer hash as
oracle/yy_oracle.c:yyo_hash_eval and tests/golden/make_golden.py: priors (1+x)/2^(pbits+6) and a
value k/2^(vbits-1) are exact in float32, so no transcendental enters an MCTS parity check."""
import numpy as np


def hash_eval_batch(boards, pbits, vbits):
    """boards int8[G,R,C] -> (policy f32[G,A], value f32[G]); vectorised over G."""
    b = np.asarray(boards, np.int8)
    G = b.shape[0]
    flat = b.reshape(G, -1)
    A = flat.shape[1]
    M = np.uint64(0xFFFFFFFF)
    h = np.full(G, 0x9E3779B9, np.uint64)
    code = (flat.astype(np.int64) & 3).astype(np.uint64)
    for i in range(A):
        h = (((h ^ code[:, i]) * np.uint64(16777619)) + np.uint64(i)) & M
    a = np.arange(A, dtype=np.uint64)[None, :]
    x = (h[:, None] + a * np.uint64(0x9E3779B1)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x2C1B3C6D)) & M
    x ^= x >> np.uint64(12)
    pol = ((np.uint64(1) + (x & np.uint64((1 << pbits) - 1))).astype(np.float32)
           / np.float32(1 << (pbits + 6))).astype(np.float32)
    y = h ^ (h >> np.uint64(16))
    y = (y * np.uint64(0x045D9F3B)) & M
    y ^= y >> np.uint64(13)
    half = 1 << (vbits - 1)
    val = (((y & np.uint64((1 << vbits) - 1)).astype(np.int64) - half) / half).astype(np.float32)
    return pol, val


def hash_eval_np(board, pbits, vbits):
    p, v = hash_eval_batch(np.asarray(board)[None], pbits, vbits)
    return p[0], np.float32(v[0])


def planes_to_boards(planes):
    """planes f32[G,5,R,C] (board_to_input layout) -> int8[G,R,C]."""
    p = np.asarray(planes)
    return (p[:, 1] - p[:, 2]).astype(np.int8)


def hash_eval_torch(planes, pbits, vbits):
    """The same hash evaluator as torch int64 ops on the device: planes f32 [G,5,R,C] -> (policy, value).
    Lets full-size batches (G = 4096) be searched without a host round trip per simulation."""
    import torch
    G = planes.shape[0]
    b = (planes[:, 1] - planes[:, 2]).to(torch.int64).reshape(G, -1)
    A = b.shape[1]
    M = 0xFFFFFFFF
    code = b & 3
    h = torch.full((G,), 0x9E3779B9, dtype=torch.int64, device=planes.device)
    for i in range(A):
        h = (((h ^ code[:, i]) * 16777619) + i) & M
    a = torch.arange(A, dtype=torch.int64, device=planes.device)[None, :]
    x = (h[:, None] + a * 0x9E3779B1) & M
    x = x ^ (x >> 15)
    x = (x * 0x2C1B3C6D) & M
    x = x ^ (x >> 12)
    pol = (1 + (x & ((1 << pbits) - 1))).to(torch.float32) / float(1 << (pbits + 6))
    y = h ^ (h >> 16)
    y = (y * 0x045D9F3B) & M
    y = y ^ (y >> 13)
    half = 1 << (vbits - 1)
    val = ((y & ((1 << vbits) - 1)) - half).to(torch.float32) / float(half)
    return pol.contiguous(), val.contiguous()

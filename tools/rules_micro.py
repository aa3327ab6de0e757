#!/usr/bin/env python3
"""Rules micro-benchmark: the packed rules kernel k_mask_terminal_bb on N positions drawn from the golden
rules fixture (random legal play at random plies + adversarial fills, the G1 distribution), tiled to N.
Per position: board in (2 bitboards), both colours' legal masks + the game-ended code out
  algorithmic bytes = Bb + 2*Mb + 1 = 16*NW + 16*NW + 1   (33 B at 8x8, 97 B at 12x12)
Prints positions/s and the achieved fraction of the 8 TB/s HBM roofline; also times the int8-tensor API kernels
(k_valid_mask for both colours + k_game_ended), which move 64 + 2*64 + 8 bytes per 8x8 position.
    python tools/rules_micro.py [R] [N_millions] [launches]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg
from yinyang_game_alphazero_amd import engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else 16 << 20
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", f"rules_{R}x{R}.npz"))
base = torch.from_numpy(z["boards"]).cuda()                       # [10240, R, R] int8
bb, bw = engine.pack_boards(base)
reps = (N + base.shape[0] - 1) // base.shape[0]
black = bb.repeat(1, reps)[:, :N].contiguous()
white = bw.repeat(1, reps)[:, :N].contiguous()
nw = black.shape[0]
out = (torch.empty_like(black), torch.empty_like(black), torch.empty(N, dtype=torch.int8, device="cuda"))


def timed(fn, launches):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(launches):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / launches * 1e-3


dt = timed(lambda: engine.mask_terminal_bb(black, white, R, R, out=out), L)
# check against the fixture on the first copy (bit-exact masks for both colours)
m1, m2, res = out
ref1 = engine.pack_boards(torch.from_numpy(z["mask_p1"].reshape(-1, R, R).astype(np.int8)).cuda())[0]
ref2 = engine.pack_boards(torch.from_numpy(z["mask_m1"].reshape(-1, R, R).astype(np.int8)).cuda())[0]
n0 = base.shape[0]
assert torch.equal(m1[:, :n0], ref1) and torch.equal(m2[:, :n0], ref2), "mask mismatch against the reference fixture"
bytes_pos = 16 * nw + 16 * nw + 1
line = {"kernel": "k_mask_terminal_bb", "board": f"{R}x{R}", "positions": N, "launch_ms": dt * 1e3,
        "positions_per_s": N / dt, "algorithmic_bytes_per_position": bytes_pos, "achieved_GBps": N * bytes_pos / dt / 1e9,
        "hbm_peak_GBps": 8000.0, "frac": N * bytes_pos / dt / 1e9 / 8000.0}
# the int8-tensor API path on a smaller tile (64 B boards): both masks + outcome = 3 launches
n8 = min(N, 4 << 20)
boards8 = base.repeat(reps, 1, 1)[:n8].contiguous()
pl = torch.ones(n8, dtype=torch.int8, device="cuda")
mi = -pl
dt8 = timed(lambda: (engine.valid_mask(boards8, pl), engine.valid_mask(boards8, mi), engine.game_ended(boards8, pl)), max(3, L // 4))
b8 = R * R * 3 + R * R * 2 + 8 + 2
line["int8_api"] = {"positions": n8, "ms_for_3_kernels": dt8 * 1e3, "positions_per_s": n8 / dt8,
                    "bytes_per_position": b8, "achieved_GBps": n8 * b8 / dt8 / 1e9}
print(json.dumps(line))

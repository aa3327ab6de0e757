"""Host restatement (test helper) of the counter-based streams of csrc/yy_selfplay.hip: Philox4x32-10 keyed by the seed with
counter (game lo, game hi, ply << 8 | purpose, element), and the move choice built on it (self_play.py:143-160)."""
import numpy as np

M32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c = [int(x) & M32 for x in counter]
    k0, k1 = int(key[0]) & M32, int(key[1]) & M32
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k1) & M32, p0 & M32]
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c


def draw(seed, game, ply, purpose, element):
    game &= 0xFFFFFFFFFFFFFFFF
    return philox4x32_10([game & M32, game >> 32, ((ply << 8) | purpose) & M32, element], [seed & M32, (seed >> 32) & M32])


def u01(hi, lo):
    return float(((hi << 32) | lo) >> 11) * 2.0 ** -53


def sample_action(seed, game, ply, pi, mask, thr):
    """pi float64 [A], mask {0,1} [A] -> action, the same float64 operations in the same order as k_sample_actions."""
    r = draw(seed, game, ply, 1, 0)
    u = u01(r[0], r[1])
    A = len(pi)
    pick = -1
    if ply < thr:
        tot, legal = 0.0, 0
        for a in range(A):
            tot += float(pi[a]) if mask[a] else 0.0
            legal += 1 if mask[a] else 0
        if tot > 0.0:
            target, c = u * tot, 0.0
            for a in range(A):
                w = float(pi[a]) if mask[a] else 0.0
                c += w
                if w > 0.0:
                    pick = a
                    if c > target:
                        break
        elif legal > 0:
            k = min(int(u * legal), legal - 1)
            for a in range(A):
                if mask[a]:
                    if k == 0:
                        pick = a
                        break
                    k -= 1
    else:
        mx = float(np.max(pi))
        best = [a for a in range(A) if float(pi[a]) == mx]
        pick = best[min(int(u * len(best)), len(best) - 1)]
    return pick

// yy_selfplay.hip -- the two random draws of the episode loop as counter-based streams keyed by the GLOBAL game index.
//
// Reference: the root Dirichlet noise np.random.dirichlet([alpha] * k) at the first move of a game
// (src/yin_yang/ai/mcts.py:298-312, self_play.py:131) and the move choice np.random.choice (self_play.py:143-160) both
// draw from ONE global numpy stream in the order one process happens to reach them.  A batched engine that did the same
// (one generator over the whole [G, A] batch) makes a game's noise and moves depend on its slot, on G, on which other games
// finished and on the world size.  Here every draw is a pure function of
//     (seed, global game index, ply, purpose, element index)
// through Philox4x32-10, so game g's transcript is the same whatever slot, batch or rank plays it (SURVEY 8(d) config 2:
// per-game seeds 1000 + g; 8(e): results independent of W).
//   purpose 0 = Dirichlet noise: Gamma(alpha) per legal cell (Marsaglia-Tsang, with the U^(1/alpha) boost for alpha < 1),
//               normalised over the game's legal cells in float64;
//   purpose 1 = move choice: ONE uniform u in [0, 1) per (game, ply); temperature 1: first action whose running sum of
//               pi[a] * mask[a] (ascending a, float64) exceeds u * total (uniform over the legal moves if the total is 0);
//               temperature 0: the floor(u * n)-th of the n actions tied at max(pi).
// One game per wavefront; the inverse CDF runs on lane 0 (A <= 192) so that its float64 sums have one fixed order, which the
// host restatement in tests/ reproduces bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace sp {

struct U4 {
    uint32_t x, y, z, w;
};
// Philox4x32-10 (Salmon et al., SC'11): counter c, key k
__device__ __forceinline__ U4 philox(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// 53-bit uniform in [0, 1) from two words
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}
// key = seed; counter = (game index lo, game index hi, ply << 8 | purpose, element)
__device__ __forceinline__ U4 draw(uint64_t seed, int64_t game, int ply, int purpose, uint32_t element) {
    const U4 c = {(uint32_t)game, (uint32_t)((uint64_t)game >> 32), ((uint32_t)ply << 8) | (uint32_t)purpose, element};
    return philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Gamma(alpha, 1): Marsaglia-Tsang on alpha' = alpha (+1 when alpha < 1), attempt t uses element = cell * 64 + t
__device__ double gamma_draw(uint64_t seed, int64_t game, int ply, int cell, double alpha) {
    const bool boost = alpha < 1.0;
    const double a = boost ? alpha + 1.0 : alpha;
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;     // value if 64 attempts were all rejected (probability < 1e-80)
    double ub = 0.5;
    for (int t = 0; t < 64; t++) {
        const U4 r = draw(seed, game, ply, 0, (uint32_t)(cell * 64 + t));
        const U4 q = draw(seed, game, ply, 2, (uint32_t)(cell * 64 + t));
        const double u1 = 1.0 - u01(r.x, r.y), u2 = u01(r.z, r.w);      // u1 in (0, 1]
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        const double v1 = 1.0 + c * z;
        if (v1 <= 0.0) continue;
        const double v = v1 * v1 * v1;
        const double u3 = 1.0 - u01(q.x, q.y);
        if (log(u3) < 0.5 * z * z + d - d * v + d * log(v)) {
            g = d * v;
            ub = 1.0 - u01(q.z, q.w);
            break;
        }
    }
    return boost ? g * pow(ub, 1.0 / alpha) : g;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// noise[g, a] = Dirichlet(alpha) over the legal cells of the games with draw[g] != 0 (zeros elsewhere: the tree kernel
// treats an all-zero row as "no noise", mcts.py:298)
__global__ void __launch_bounds__(64) k_root_noise(uint64_t seed, const int64_t *__restrict__ game_id,
                                                   const int32_t *__restrict__ ply, const uint8_t *__restrict__ drawf,
                                                   const uint8_t *__restrict__ mask, int A, double alpha,
                                                   double *__restrict__ noise) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const bool on = drawf[g] != 0;
    double x[3] = {0.0, 0.0, 0.0}, part = 0.0;
    if (on) {
        const int64_t gid = game_id[g];
        const int p = ply[g];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int a = j * 64 + lane;
            if (a < A && mask[(size_t)g * A + a]) {
                x[j] = gamma_draw(seed, gid, p, a, alpha);
                part += x[j];
            }
        }
    }
    // one fixed summation order: lanes' partial sums (cells a, a+64, a+128) combined by the xor butterfly
    const double tot = wave_sum_f64(part);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int a = j * 64 + lane;
        if (a < A) noise[(size_t)g * A + a] = (on && tot > 0.0) ? x[j] / tot : 0.0;
    }
}

// self_play.py:143-160; action = -1 for games that are not searching
__global__ void __launch_bounds__(64) k_sample_actions(uint64_t seed, const int64_t *__restrict__ game_id,
                                                       const int32_t *__restrict__ ply, const uint8_t *__restrict__ searching,
                                                       const double *__restrict__ pi, const uint8_t *__restrict__ mask, int A,
                                                       int temperature_threshold, int32_t *__restrict__ action) {
    const int g = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (!searching[g]) {
        action[g] = -1;
        return;
    }
    const double *p = pi + (size_t)g * A;
    const uint8_t *m = mask + (size_t)g * A;
    const U4 r = draw(seed, game_id[g], ply[g], 1, 0u);
    const double u = u01(r.x, r.y);
    int pick = -1;
    if (ply[g] < temperature_threshold) {            // temperature 1: p = pi * valid / sum (:150-160)
        double tot = 0.0;
        int legal = 0;
        for (int a = 0; a < A; a++) {
            tot += m[a] ? p[a] : 0.0;
            legal += m[a] != 0;
        }
        if (tot > 0.0) {
            const double target = u * tot;
            double c = 0.0;
            for (int a = 0; a < A; a++) {
                const double w = m[a] ? p[a] : 0.0;
                c += w;
                if (w > 0.0) {
                    pick = a;                           // remembers the last action with mass (rounding guard)
                    if (c > target) break;
                }
            }
        } else if (legal > 0) {                         // uniform over the legal moves (:156-158)
            int k = min((int)(u * (double)legal), legal - 1);
            for (int a = 0; a < A; a++)
                if (m[a] && k-- == 0) { pick = a; break; }
        }
    } else {                                            // temperature 0: random arg-max (:143-146)
        double mx = p[0];
        for (int a = 1; a < A; a++) mx = fmax(mx, p[a]);
        int n = 0;
        for (int a = 0; a < A; a++) n += (p[a] == mx);
        int k = min((int)(u * (double)n), n - 1);
        for (int a = 0; a < A; a++)
            if (p[a] == mx && k-- == 0) { pick = a; break; }
    }
    action[g] = pick;
}

}   // namespace sp

extern "C" int yy_selfplay_root_noise(uint64_t seed, const int64_t *game_id, const int32_t *ply, const uint8_t *draw,
                                      const uint8_t *mask, int G, int A, double alpha, double *noise, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!game_id || !ply || !draw || !mask || !noise || G < 0 || A < 1 || A > 192 || !(alpha > 0.0))
        return yy_tower_set_err(YY_E_INVALID, "yy_selfplay_root_noise: bad argument");
    sp::k_root_noise<<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(seed, game_id, ply, draw, mask, A, alpha, noise);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_selfplay_root_noise: launch failed");
    return YY_OK;
}

extern "C" int yy_selfplay_sample_actions(uint64_t seed, const int64_t *game_id, const int32_t *ply, const uint8_t *searching,
                                          const double *pi, const uint8_t *mask, int G, int A, int temperature_threshold,
                                          int32_t *action, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!game_id || !ply || !searching || !pi || !mask || !action || G < 0 || A < 1 || A > 192)
        return yy_tower_set_err(YY_E_INVALID, "yy_selfplay_sample_actions: bad argument");
    sp::k_sample_actions<<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(seed, game_id, ply, searching, pi, mask, A,
                                                                   temperature_threshold, action);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_selfplay_sample_actions: launch failed");
    return YY_OK;
}

// yy_fc_heads.hip -- the two fully connected heads of the evaluator, policy_fc and value_fc1, as ONE hand-written split-f16
// GEMM kernel on the rows a launch really evaluated (src/yin_yang/ai/neural_network.py:115 policy_fc, :120 value_fc1; the
// reference runs them in float32 on the CPU, one board per call).
//
// Why not a library GEMM: the engine's evaluation reuse (pass values, per-game cache, opening book) feeds results computed in
// a batch of one size into searches that would have computed them in a batch of another size.  That is only sound when a
// row's (policy, value) is a function of that row's planes alone, bit for bit.  The tower kernels have that property by
// construction; a library GEMM picks its tile shape / split-K from the batch size.  Here every output element is one fixed
// chain: k ascending in steps of 32 through v_mfma_f32_16x16x32_f16, never split, whatever the row's position or the batch
// size -- the property holds by construction (tests/test_gpu_network.py::test_evaluator_rows_do_not_depend_on_the_batch).
// It also reads the device-side row count, so rows that were not evaluated cost nothing.
//
// Numerics as in the tower (yy_tower_g.hip): x = hi + lo float16 pairs (22 significant bits), weights times 2^kw, features
// times 2^ka, acc1 += w_hi*x_hi, acc2 += w_lo*x_hi + w_hi*x_lo, out = (acc1 + acc2) * 2^-(kw+ka) + bias.
//
// Work split: job = (tile of 64 -- 32 for batches of at most 2048 rows -- dense rows, slice of <= 64 outputs of one head);
// wave w = outputs [16w, 16w+16) of the slice x the tile's rows (16-row blocks).  Features f32 [row][head][K] are staged through LDS in chunks of 128 k, split into
// hi / lo on the way (rows of 256 B at a stride of 288 B: conflict-free for this MFMA's B-operand reads), double buffered,
// one barrier per chunk; the weights of a slice ([k-step][wave][part][lane][8 f16], network.pack_fc_heads) go global ->
// register in MFMA operand order, one chunk (four k-steps) ahead.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace fch {

constexpr int KC = 128, RS = 288;

__device__ __forceinline__ void split_pair(const f32x2 a, uint32_t &hi, uint32_t &lo) {
    const f16x2 h = __builtin_convertvector(a, f16x2);
    const f32x2 r = a - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

struct Job {
    int head, out0, nout, wchunk;      // wchunk: first 8 KB k-step block of the slice's weights
};

template <int ROWS>      // dense rows per tile: 64, or 32 for small batches (twice the workgroups, half the feature bytes per workgroup)
__global__ void __launch_bounds__(256, 2)
k_fc_heads(const float *__restrict__ feats, const unsigned char *__restrict__ wpk, const float *__restrict__ bias,
           const Job *__restrict__ jobs, int n_jobs, float *__restrict__ logits, float *__restrict__ hidden,
           const int *__restrict__ n_rows, int G, int K, int A, int H, float in_scale, float out_scale) {
    constexpr int PART_BYTES = ROWS * RS, STAGE_BYTES = 2 * PART_BYTES, NBR = ROWS / 16, NI = ROWS / 8;
    __shared__ __attribute__((aligned(256))) unsigned char lds[2 * STAGE_BYTES];
    const int n_live = n_rows ? min(*n_rows, G) : G;
    const int tile = blockIdx.x / n_jobs, jb = blockIdx.x - tile * n_jobs;
    const int r0 = tile * ROWS;
    if (r0 >= n_live) return;                                // whole workgroup, before any barrier
    const Job job = jobs[jb];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int n_chunks = (K + KC - 1) / KC;
    const bool active = wave * 16 < job.nout;                // a wave past the slice's outputs only helps staging

    // staging: thread t moves float4 #(t % 32) of rows t/32 + 8*i (i = 0 .. ROWS/8 - 1) of the chunk
    const int sq = threadIdx.x & 31, sr = threadIdx.x >> 5;
    auto stage_load = [&](f32x4 (&fr)[NI], int c) {
        const int k = c * KC + sq * 4;
        const int kc = k < K ? k : 0;                                             // unconditional loads (no branch between the
        const float km = k < K ? 1.0f : 0.0f;                                     // prefetches); k >= K contributes zeros
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int row = min(r0 + sr + 8 * i, n_live - 1);                    // rows past the live count repeat the last one (never stored)
            fr[i] = *(const f32x4 *)(feats + ((size_t)row * 2 + job.head) * K + kc) * km;
        }
    };
    auto stage_store = [&](const f32x4 (&fr)[NI], int s) {
        unsigned char *base = lds + s * STAGE_BYTES + sq * 8;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const f32x4 v = fr[i] * in_scale;
            uint32_t h01, l01, h23, l23;
            split_pair((f32x2){v[0], v[1]}, h01, l01);
            split_pair((f32x2){v[2], v[3]}, h23, l23);
            *(u32x2 *)(base + (sr + 8 * i) * RS) = (u32x2){h01, h23};
            *(u32x2 *)(base + PART_BYTES + (sr + 8 * i) * RS) = (u32x2){l01, l23};
        }
    };
    const unsigned char *wbase = wpk + ((size_t)job.wchunk * 4 + wave) * 2048 + lane * 16;   // + ks * 8192; hi, lo 1 KB apart
    f32x4 acc1[NBR], acc2[NBR];
#pragma unroll
    for (int nb = 0; nb < NBR; nb++) acc1[nb] = acc2[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this wave's weight fragments of one chunk: [k-step 4]{hi, lo}; loaded a whole chunk (768 MFMA cycles) ahead
    auto load_wc = [&](f16x8 (&w)[KC / 32][2], int c) {
#pragma unroll
        for (int ks = 0; ks < KC / 32; ks++) {
            const unsigned char *p = wbase + (size_t)(c * (KC / 32) + ks) * 8192;
            w[ks][0] = __builtin_bit_cast(f16x8, *(const u32x4 *)p);
            w[ks][1] = __builtin_bit_cast(f16x8, *(const u32x4 *)(p + 1024));
        }
    };
    // One chunk: 16 groups (k-step, row block) of {2 ds_read_b128, 3 MFMAs}; the fragments of group g + 4 are requested right
    // after the MFMAs of group g (a rotating window of four), so the LDS latency is paid once per chunk, not once per group.
    auto compute = [&](const f16x8 (&w)[KC / 32][2], int stage) {
        if (!active) return;
        const unsigned char *xs = lds + stage * STAGE_BYTES + n16 * RS + kg * 16;
        constexpr int NG = (KC / 32) * NBR, P = 4;
        f16x8 xh[P], xl[P];
        auto rd = [&](int g, int slot) {
            const int ks = g / NBR, nb = g % NBR;
            xh[slot] = __builtin_bit_cast(f16x8, *(const u32x4 *)(xs + nb * 16 * RS + ks * 64));
            xl[slot] = __builtin_bit_cast(f16x8, *(const u32x4 *)(xs + PART_BYTES + nb * 16 * RS + ks * 64));
        };
#pragma unroll
        for (int g = 0; g < P; g++) rd(g, g);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * P, 0);      // the window's first fill stays in front of the first MFMA
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int ks = g / NBR, nb = g % NBR, slot = g % P;
            const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks][1], xh[slot], acc2[nb], 0, 0, 0);
            acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks][0], xh[slot], acc1[nb], 0, 0, 0);
            acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks][0], xl[slot], a, 0, 0, 0);
            if (g + P < NG) rd(g + P, slot);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
    // Software pipeline, two chunks per trip so that every buffer has a fixed name: the feature rows of chunk c + 3 and the
    // weight fragments of chunk c + 2 are requested while chunk c computes, i.e. two chunks (~1 us) before they are used --
    // the feature tensor was just written by the tower launch and comes from the Infinity Cache / HBM.  Chunk indices past
    // the end are clamped (they re-read the last chunk; never stored, never multiplied).
    const int last = n_chunks - 1;
    f32x4 f0[NI], f1[NI];                    // f0: the odd chunk after the one computing, f1: the even chunk after that
    f16x8 w0[KC / 32][2], w1[KC / 32][2];    // w0: weights of the even chunk computing / to come, w1: of the odd one
    stage_load(f1, 0);
    load_wc(w0, 0);
    stage_load(f0, min(1, last));
    load_wc(w1, min(1, last));
    stage_store(f1, 0);
    stage_load(f1, min(2, last));
    __syncthreads();
    for (int c = 0; c < n_chunks; c += 2) {
        compute(w0, 0);                                          // chunk c (even) from stage 0
        if (c + 1 < n_chunks) stage_store(f0, 1);                // chunk c + 1 -> stage 1
        stage_load(f0, min(c + 3, last));
        load_wc(w0, min(c + 2, last));
        __syncthreads();
        if (c + 1 >= n_chunks) break;
        compute(w1, 1);                                          // chunk c + 1 (odd) from stage 1
        if (c + 2 < n_chunks) stage_store(f1, 0);                // chunk c + 2 -> stage 0
        stage_load(f1, min(c + 4, last));
        load_wc(w1, min(c + 3, last));
        __syncthreads();
    }
    if (!active) return;
    // lane: outputs o0 + 0..3 (o0 = out0 + 16*wave + 4*kg) of dense row r0 + nb*16 + n16
    float *out = job.head == 0 ? logits : hidden;
    const int W = job.head == 0 ? A : H, boff = job.head == 0 ? 0 : A;
    const int o0 = job.out0 + wave * 16 + kg * 4;
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = (o0 + i < job.out0 + job.nout) ? bias[boff + o0 + i] : 0.0f;
#pragma unroll
    for (int nb = 0; nb < NBR; nb++) {
        const int row = r0 + nb * 16 + n16;
        if (row < n_live) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (o0 + i < job.out0 + job.nout) out[(size_t)row * W + o0 + i] = __builtin_fmaf(acc1[nb][i] + acc2[nb][i], out_scale, b[i]);
        }
    }
}

}   // namespace fch

// feats f32 [G,2,K] (dense rows: the tower's out_heads); wpk = network.pack_fc_heads (f16 [n_jobs][ceil(K/128)*4][4][2][64][8]
// times 2^weight_exp); bias f32 [A + H] (policy_fc.bias, value_fc1.bias); jobs int32 [n_jobs][4] = (head, first output, outputs
// <= 64, first 8 KB weight block) on the device; logits f32 [G,A], hidden f32 [G,H]; n_rows (device, or NULL = G): rows past it
// are neither read nor written.  A row's outputs do not depend on G, *n_rows or its position.
extern "C" int yy_nn_fc_heads_f16x3(const float *feats, const void *wpk, const float *bias, const int32_t *jobs, int n_jobs,
                                    float *logits, float *hidden, const int32_t *n_rows, int G, int K, int A, int H,
                                    int weight_exp, int act_exp, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!feats || !wpk || !bias || !jobs || !logits || !hidden || G < 0 || n_jobs < 1 || K < 32 || (K & 31) || A < 1 || H < 1)
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_fc_heads_f16x3: bad argument");
    // tiles of 32 rows for batches up to 2048 rows (a lane of the engine), 64 above: the same bits either way (one fixed
    // k-ascending chain per output), only the work split differs
    const float in_scale = ldexpf(1.0f, act_exp), out_scale = ldexpf(1.0f, -(weight_exp + act_exp));
    if (G <= 2048)
        fch::k_fc_heads<32><<<dim3(((G + 31) / 32) * n_jobs), dim3(256), 0, (hipStream_t)s>>>(
            feats, (const unsigned char *)wpk, bias, (const fch::Job *)jobs, n_jobs, logits, hidden, n_rows, G, K, A, H, in_scale, out_scale);
    else
        fch::k_fc_heads<64><<<dim3(((G + 63) / 64) * n_jobs), dim3(256), 0, (hipStream_t)s>>>(
            feats, (const unsigned char *)wpk, bias, (const fch::Job *)jobs, n_jobs, logits, hidden, n_rows, G, K, A, H, in_scale, out_scale);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_fc_heads_f16x3: launch failed");
    return YY_OK;
}

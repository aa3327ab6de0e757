// yy_tower_h3r.hip -- the round-2 split-f16 (float32-accurate) tower on v_mfma_f32_32x32x16_f16 with the weight stream kept in
// REGISTERS, now the A/B partner of the general kernel (yy_tower_g.hip; evaluator mode "f16x3r"): wave w = output channels [32w, 32w+32) x all (board, cell) columns of the workgroup.  Template
// <R, TB, D, NV>: TB boards of R x R cells per workgroup, a ring of D weight chunks per wave, NV of them in VGPRs:
//   <8, 2, 9, 4>  8x8, two boards = four 32-column tiles: the headline evaluator kernel (BASELINE config 2);
//   <6, 4, 3, 2>  6x6, four boards, and <12, 1, 3, 2> 12x12, one board: 144 columns = five tiles (the last half padding);
//                 five {acc1, acc2} pairs leave room for a 3-chunk ring (configs 1 and 4).
//
// Why.  In the cout-quarter form every wave multiplies only ITS quarter of each weight chunk, so staging weights through LDS
// buys nothing: a wave can load its own fragments global -> register in exactly the MFMA operand layout (one coalesced 1 KB
// global_load_dwordx4 per fragment) and keep a ring of D chunks (8x8: D = 9 = 144 registers; a 512-register wave has them to
// spare).  That removes the LDS-DMA issue slots and the per-chunk ring handshake, takes 2 of the 10 ds_read_b128 per k-step
// away, and deepens the prefetch from 3 chunks (~1.1 us of MFMA time) to 9 (~3.3 us): the 12 MB weight stream does not fit an
// XCD's 4 MB L2, every round of workgroups re-streams it from the Infinity Cache, and the workgroups of an XCD run in step,
// so every chunk exposed that latency to the 3-deep LDS ring.  Measured on 4096 boards: 3.44 ms against 3.65 ms (cout-quarter
// form with wave-private LDS rings) and 3.72 ms (board x cout-half form, shared LDS ring) in the same process.
// LDS now holds only the activations ({hi, lo} x (columns + a zero row) x 272 B), the bias table and (8x8) the f32 residual.
//
// Numerics (as yy_tower_g.hip): x = hi + lo with hi = f16(x), lo = f16(x - hi); weights are stored times 2^kw and activations (and
// the bias table) live times 2^ka so that the lo parts stay in float16's normal range; per tile one f32 accumulator takes
// w_hi*x_hi and a second one w_lo*x_hi + w_hi*x_lo; the epilogue is fma(acc1 + acc2, 2^-kw, bias).  Weight layout ("wave-major",
// network.pack_tower_h3r): chunk = one tap x 32 input channels = [nt 4][ks 2][part 2][h 2][c 32][j 8] f16 (4 KB per wave,
// contiguous), cin = quarter*32 + ks*16 + h*8 + j, cout = nt*32 + c; stem: one chunk per tap (ks 0 only carries data), every
// other layer 36 (tap-major, then quarter); heads: [head 2][ks 8][part 2][h 2][c 32][j 8] (cin = ks*16 + h*8 + j).
// Reference: src/yin_yang/ai/neural_network.py:16-33, 94-119 (float32 on the CPU).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define HR_CH 128
#define HR_ROW_BYTES 272
#define HR_CHUNK_BYTES 16384
#define HR_MAX_LAYERS 22

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace thr {

template <int R_, int TB_, int D_, int NV_> struct Geo {
    static constexpr int R = R_, TB = TB_, D = D_, NV = NV_, CELLS = R_ * R_, NCOL = TB_ * R_ * R_, CT = (NCOL + 31) / 32;
    static constexpr int PART_BYTES = (NCOL + 1) * HR_ROW_BYTES;    // one part (hi or lo) of every column + its zero row
    static constexpr int ZERO_OFF = NCOL * HR_ROW_BYTES;
    static constexpr int BIAS_OFF = 2 * PART_BYTES;
    // the f32 residual of a block input lives in LDS where it fits (8x8: 4 waves x 16 KB, lane-private 16-B slots, so the
    // 64 registers it would occupy across the MFMA loop are free); 6x6 / 12x12 keep it in registers
    static constexpr bool RES_LDS = (CT == 4);
    static constexpr int RES_OFF = BIAS_OFF + HR_MAX_LAYERS * HR_CH * 4;
    static constexpr int LDS_BYTES = RES_OFF + (RES_LDS ? 4 * 16 * 1024 : 0);
    static_assert(36 % D_ == 0 && 9 % D_ == 0, "ring depth must divide the chunks of a layer (36) and of the stem (9)");
    static_assert(PART_BYTES + 256 < 65536 && LDS_BYTES <= 163840, "LDS layout");
};

// two f32 -> packed (hi, hi) and (lo, lo) f16 pairs: hi = f16(x) (round to nearest even), lo = f16(x - hi); x - hi is exact in f32
__device__ __forceinline__ void split_pair(const f32x2 a, uint32_t &hi, uint32_t &lo) {
    const f16x2 h = __builtin_convertvector(a, f16x2);
    const f32x2 r = a - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ f32x2 join_pair(const uint32_t hi, const uint32_t lo) {
    const f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
    const f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
    return h + l;
}

// this wave's fragments of one weight chunk: [ks 2]{hi, lo}, 16 registers
struct WChunk {
    f16x8 h[2], l[2];
};
// The four 1 KB fragment loads of a chunk, written as asm so that they stay where they are put (left to itself the scheduler
// sinks a plain load towards its use, which is the opposite of a prefetch) and so that the ring can be waited for with ONE
// counted vmcnt: wbase = weights + chunk * 16 KB (uniform, SGPRs), voff = wave * 4096 + lane * 16.  The first NV ring entries
// live in VGPRs, the others in AGPRs (an MFMA reads its A operand from either; the accumulators leave AGPRs free): the VGPR
// half alone cannot hold the ring beside the activation fragments and the residual.
#define HR_LOAD_ASM(C)                                                                                            \
    asm volatile("global_load_dwordx4 %0, %4, %5\n\t"                                                             \
                 "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"                                                 \
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"                                                 \
                 "global_load_dwordx4 %3, %4, %5 offset:3072"                                                      \
                 : "=&" C(w.h[0]), "=&" C(w.l[0]), "=&" C(w.h[1]), "=&" C(w.l[1])                                  \
                 : "v"(voff), "s"(wbase)                                                                          \
                 : "memory")
__device__ __forceinline__ void load_w(WChunk &w, const unsigned char *wbase, uint32_t voff, bool in_vgpr) {
    if (in_vgpr) HR_LOAD_ASM("v");
    else HR_LOAD_ASM("a");
}
// All loads of a chunk have landed once at most 4 * (D - 1) younger loads are outstanding (loads return in order; every ring
// entry is refilled right after its use, the tail of the stream re-reads its last chunk, so the count is uniform).  The
// fragments are tied to the wait as in/out operands: no MFMA that reads them can be scheduled above it.
template <int D> __device__ __forceinline__ void wait_w(WChunk &w, bool in_vgpr) {
    if (in_vgpr) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w.h[0]), "+v"(w.l[0]), "+v"(w.h[1]), "+v"(w.l[1]) : "n"(4 * (D - 1)));
    else asm volatile("s_waitcnt vmcnt(%4)" : "+a"(w.h[0]), "+a"(w.l[0]), "+a"(w.h[1]), "+a"(w.l[1]) : "n"(4 * (D - 1)));
}
// activation fragments of one k-step: hi / lo of the column tiles
template <int CT> struct XFrags {
    f16x8 h[CT], l[CT];
};
template <class GEO>
__device__ __forceinline__ void load_x(XFrags<GEO::CT> &f, const unsigned char *lds, int quarter, int ks, const uint32_t (&cb)[GEO::CT]) {
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) {
        f.h[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cb[tt] + quarter * 64 + ks * 32));
        f.l[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cb[tt] + GEO::PART_BYTES + quarter * 64 + ks * 32));
    }
}
// acc1 += w_hi * x_hi ;  acc2 += w_lo * x_hi + w_hi * x_lo.  Two accumulators because every MFMA rounds its accumulator once:
// the large sum is rounded once per k-step (as in an f32 dot product) and the 2^-11-times-smaller corrections round among
// themselves; one shared accumulator measured 1.7x the error (three roundings of the large sum per k-step).
template <int CT, bool ZERO>
__device__ __forceinline__ void mma(f32x16 (&acc1)[CT], f32x16 (&acc2)[CT], const f16x8 wh, const f16x8 wl, const XFrags<CT> &x) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < CT; tt++) {
        const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, x.h[tt], ZERO ? z : acc2[tt], 0, 0, 0);
        acc1[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.h[tt], ZERO ? z : acc1[tt], 0, 0, 0);
        acc2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.l[tt], a, 0, 0, 0);
    }
}
// the 2*CT LDS reads of the next k-step inside the 3*CT MFMAs of this one
template <int CT> __device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int j = 0; j < CT; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * CT, 0);
}
// per-lane geometry: column tt*32 + c (board = col / CELLS, cell = col % CELLS); pad columns have an empty tap mask
template <class GEO> struct LaneGeo {
    uint32_t rowbase[GEO::CT], okmask[GEO::CT], zbase;
};
template <class GEO> __device__ __forceinline__ void make_lane_geo(LaneGeo<GEO> &g, int c, int h) {
    g.zbase = (uint32_t)GEO::ZERO_OFF + (uint32_t)(h * 16);
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) {
        const int col = tt * 32 + c;
        const int cell = col % GEO::CELLS;
        const int y = cell / GEO::R, x = cell - y * GEO::R;
        g.rowbase[tt] = (uint32_t)(col * HR_ROW_BYTES) + (uint32_t)(h * 16);
        uint32_t m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
            if (((unsigned)sy < (unsigned)GEO::R) && ((unsigned)sx < (unsigned)GEO::R)) m |= 1u << tap;
        }
        g.okmask[tt] = (col < GEO::NCOL) ? m : 0u;
    }
}
template <class GEO> __device__ __forceinline__ void tap_geo(int tap, const LaneGeo<GEO> &g, uint32_t (&cb)[GEO::CT]) {
    const int shift = ((tap / 3 - 1) * GEO::R + (tap % 3 - 1)) * HR_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) cb[tt] = (g.okmask[tt] & bit) ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
}

// One layer: NCH chunks = 9 taps x Q quarters (stem: Q = 1 and one k-step of 16 zero-padded channels), processed in groups of
// D with the ring position = chunk index within the group (a compile-time register).  After its last use a ring entry is
// refilled with the chunk D further down the stream (the next group / the next layer).
template <class GEO, bool STEM>
__device__ __forceinline__ void run_layer(f32x16 (&acc1)[GEO::CT], f32x16 (&acc2)[GEO::CT], WChunk (&W)[GEO::D], const unsigned char *lds,
                                          const unsigned char *weights, uint32_t voff, int &chunk, int n_tower, const LaneGeo<GEO> &geo) {
    constexpr int Q = STEM ? 1 : 4, KS = STEM ? 1 : 2, NCH = 9 * Q, D = GEO::D, NG = NCH / D, CT = GEO::CT;
    uint32_t cb[CT];
    tap_geo<GEO>(0, geo, cb);
    XFrags<CT> cur;
    load_x<GEO>(cur, lds, 0, 0, cb);
#pragma unroll 1
    for (int g = 0; g < NG; g++) {
#pragma unroll
        for (int j = 0; j < D; j++, chunk++) {
            const int i = g * D + j;                          // chunk within the layer
            const int quarter = STEM ? 0 : (i & 3);
            const bool last = (i == NCH - 1);
            const int ni = last ? i : i + 1;
            uint32_t ncb[CT];
            tap_geo<GEO>(STEM ? ni : (ni >> 2), geo, ncb);
            const int nquarter = STEM ? 0 : (ni & 3);
            wait_w<D>(W[j], j < GEO::NV);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                XFrags<CT> nxt;
                const bool has_next = (ks + 1 < KS) || !last;
                if (ks + 1 < KS) load_x<GEO>(nxt, lds, quarter, ks + 1, cb);
                else if (!last) load_x<GEO>(nxt, lds, nquarter, 0, ncb);
                if (i == 0 && ks == 0) mma<CT, true>(acc1, acc2, W[j].h[ks], W[j].l[ks], cur);
                else mma<CT, false>(acc1, acc2, W[j].h[ks], W[j].l[ks], cur);
                if (has_next) {
                    interleave_hint<CT>();
                    cur = nxt;
                }
            }
            // refill the ring entry just consumed (the tail of the stream re-reads the last chunk: uniform vmcnt)
            load_w(W[j], weights + (size_t)min(chunk + D, n_tower - 1) * HR_CHUNK_BYTES, voff, j < GEO::NV);
#pragma unroll
            for (int tt = 0; tt < CT; tt++) cb[tt] = ncb[tt];
        }
    }
}

template <int R_, int TB_, int D_, int NV_>
__global__ void __launch_bounds__(256, 1)
k_tower_h3r(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const unsigned char *__restrict__ head_w,
            const float *__restrict__ bias, float *__restrict__ out, float *__restrict__ out_heads,
            const int *__restrict__ rows, const int *__restrict__ n_rows, int G, int n_layers, float in_scale,
            float acc_scale, float head_scale, float out_scale, int gate_lo, int gate_hi) {
    using GEO = Geo<R_, TB_, D_, NV_>;
    constexpr int CT = GEO::CT, CELLS = GEO::CELLS, NCOL = GEO::NCOL, TB = GEO::TB, D = GEO::D;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GEO::LDS_BYTES];
    const int n_live = n_rows ? min(*n_rows, G) : G;
    if (n_live <= gate_lo || n_live > gate_hi) return;     // the launch only runs for row counts in (gate_lo, gate_hi]: see yy_nn_tower_heads_f16x3_auto
    const int g0 = blockIdx.x * TB;                        // first dense row of the workgroup
    if (g0 >= n_live) return;                              // whole workgroup, before any barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nh = wave;                                   // this wave's output-channel quarter
    const int h = lane >> 5, c = lane & 31;
    const int n_tower = 9 + 36 * (n_layers - 1);

    // the weight stream starts first: D chunks in flight before anything else is touched
    const uint32_t voff = (uint32_t)(wave * 4096 + lane * 16);
    WChunk W[D];
#pragma unroll
    for (int j = 0; j < D; j++) load_w(W[j], weights + (size_t)j * HR_CHUNK_BYTES, voff, j < GEO::NV);

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * HR_CH; i += 256) ((float *)(lds + GEO::BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 128)   // the zero rows of both parts
        ((uint32_t *)(lds + GEO::ZERO_OFF + (threadIdx.x >> 6) * GEO::PART_BYTES))[threadIdx.x & 63] = 0u;
    if (threadIdx.x < NCOL) {   // 5 planes -> channels 0..4 of a 16-channel zero-padded input
        const int col = threadIdx.x, gb = g0 + col / CELLS, cell = col % CELLS;
        const bool live = gb < n_live;
        const int src = live ? (rows ? rows[gb] : gb) : 0;
        float p[6];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = live ? planes[((size_t)src * 5 + k) * CELLS + cell] * in_scale : 0.0f;
        p[5] = 0.0f;
        uint32_t hi[3], lo[3];
#pragma unroll
        for (int k = 0; k < 3; k++) split_pair((f32x2){p[2 * k], p[2 * k + 1]}, hi[k], lo[k]);
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + col * HR_ROW_BYTES) = (u32x4){hi[0], hi[1], hi[2], 0u};
        *(u32x4 *)(lds + col * HR_ROW_BYTES + 16) = z;
        *(u32x4 *)(lds + GEO::PART_BYTES + col * HR_ROW_BYTES) = (u32x4){lo[0], lo[1], lo[2], 0u};
        *(u32x4 *)(lds + GEO::PART_BYTES + col * HR_ROW_BYTES + 16) = z;
    }

    LaneGeo<GEO> geo;
    make_lane_geo<GEO>(geo, c, h);
    f32x4 res[CT][4];      // residual x of this wave's 32 couts, f32 (registers: only when !GEO::RES_LDS)
    unsigned char *res_lds = lds + GEO::RES_OFF + (wave * 16) * 1024 + lane * 16;   // slot (q * CT + tt) * 1 KB, lane-private
    int chunk = 0;
    // (acc1 + acc2) * 2^-kw + bias (+ residual) + ReLU in f32, split again, back to LDS (activations, bias table and residual
    // live in the 2^ka-scaled domain); KEEP: the output is a block input x, kept for the skip; resolved once per layer
    auto epilogue = [&](const int L, f32x16 (&acc1)[CT], f32x16 (&acc2)[CT], auto conv2_tag, auto keep_tag) {
        constexpr bool CONV2 = decltype(conv2_tag)::value, KEEP = decltype(keep_tag)::value;
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bq[q] = *(const f32x4 *)(lds + GEO::BIAS_OFF + (L * HR_CH + nh * 32 + 8 * q + 4 * h) * 4);
        f32x4 rl[GEO::RES_LDS && CONV2 ? CT : 1][4];
        if (GEO::RES_LDS && CONV2) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int tt = 0; tt < CT; tt++) rl[tt][q] = *(const f32x4 *)(res_lds + (q * CT + tt) * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = nh * 32 + 8 * q + 4 * h;       // this lane's 4 couts
            const f32x4 b = bq[q];
#pragma unroll
            for (int tt = 0; tt < CT; tt++) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = __builtin_fmaf(acc1[tt][4 * q + i] + acc2[tt][4 * q + i], acc_scale, b[i]);
                if (CONV2) v += GEO::RES_LDS ? rl[GEO::RES_LDS && CONV2 ? tt : 0][q] : res[tt][q];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.0f);
                if (KEEP) {
                    if (GEO::RES_LDS) *(f32x4 *)(res_lds + (q * CT + tt) * 1024) = v;
                    else res[tt][q] = v;
                }
                uint32_t h01, l01, h23, l23;
                split_pair((f32x2){v[0], v[1]}, h01, l01);
                split_pair((f32x2){v[2], v[3]}, h23, l23);
                const int col = tt * 32 + c;
                if (NCOL % 32 == 0 || col < NCOL) {
                    *(u32x2 *)(lds + col * HR_ROW_BYTES + co * 2) = (u32x2){h01, h23};
                    *(u32x2 *)(lds + GEO::PART_BYTES + col * HR_ROW_BYTES + co * 2) = (u32x2){l01, l23};
                }
            }
        }
    };
    {   // the stem (its own code: 9 one-k-step chunks), outside the loop over the 128 -> 128 layers
        f32x16 acc1[CT], acc2[CT];
        __syncthreads();                                   // the prologue's LDS writes are visible
        run_layer<GEO, true>(acc1, acc2, W, lds, weights, voff, chunk, n_tower, geo);
        __syncthreads();                                   // every wave has finished reading the input
        epilogue(0, acc1, acc2, std::false_type{}, std::true_type{});
    }
    for (int L = 1; L < n_layers; L++) {
        f32x16 acc1[CT], acc2[CT];
        __syncthreads();                                   // the previous layer's epilogue is visible
        run_layer<GEO, false>(acc1, acc2, W, lds, weights, voff, chunk, n_tower, geo);
        __syncthreads();                                   // every wave has finished reading this layer's input
        if ((L & 1) == 0) epilogue(L, acc1, acc2, std::true_type{}, std::true_type{});      // second conv of a block: + skip, keep
        else epilogue(L, acc1, acc2, std::false_type{}, std::false_type{});
    }
    __syncthreads();
    if (out_heads) {
        // 1x1 head convs (neural_network.py:113, 118): wave w: head (w & 1), column tiles [t0, t0 + HT) with t0 = (w >> 1) * HT
        // (a surplus tile of waves 2, 3 is a clamped duplicate, never stored)
        constexpr int HT = (CT + 1) / 2;
        const int head = wave & 1, t0 = (wave >> 1) * HT, nt_cnt = (wave >> 1) ? CT - HT : HT;
        const unsigned char *hw = head_w + head * 16384 + lane * 16;       // [head][ks 8][part 2][1 KB]
        f32x16 h1[HT], h2[HT];
        uint32_t xb[HT];
#pragma unroll
        for (int t = 0; t < HT; t++) xb[t] = (uint32_t)(min((t0 + t) * 32 + c, NCOL - 1) * HR_ROW_BYTES + h * 16);
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const f16x8 wh = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 2048));
            const f16x8 wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 2048 + 1024));
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const f16x8 xh = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[t] + ks * 32));
                const f16x8 xl = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[t] + GEO::PART_BYTES + ks * 32));
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, ks == 0 ? z : h2[t], 0, 0, 0);
                h1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, ks == 0 ? z : h1[t], 0, 0, 0);
                h2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a, 0, 0, 0);
            }
        }
        // features f32 [row][head][channel 32][cell CELLS] (the reference's NCHW flatten order)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(lds + GEO::BIAS_OFF + (n_layers * HR_CH + head * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const int col = (t0 + t) * 32 + c;
                if (t < nt_cnt && col < NCOL) {
                    const int gb = g0 + col / CELLS, cell = col % CELLS;
                    if (gb < n_live) {
                        float *o = out_heads + (((size_t)gb * 2 + head) * 32 + 8 * q + 4 * h) * CELLS + cell;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            o[i * CELLS] = fmaxf(__builtin_fmaf(h1[t][4 * q + i] + h2[t][4 * q + i], head_scale, b[i]), 0.0f);
                    }
                }
            }
        }
        return;
    }
    for (int p = threadIdx.x; p < NCOL * 32; p += 256) {   // activations [column][128] f32 = (hi + lo) * 2^-ka
        const int col = p >> 5, ch4 = p & 31;
        if (g0 + col / CELLS < n_live) {
            const u32x2 ph = *(const u32x2 *)(lds + col * HR_ROW_BYTES + ch4 * 8);
            const u32x2 pl = *(const u32x2 *)(lds + GEO::PART_BYTES + col * HR_ROW_BYTES + ch4 * 8);
            const f32x2 v01 = join_pair(ph.x, pl.x), v23 = join_pair(ph.y, pl.y);
            *(f32x4 *)(out + ((size_t)g0 * CELLS + col) * HR_CH + ch4 * 4) = (f32x4){v01.x, v01.y, v23.x, v23.y} * out_scale;
        }
    }
}

}   // namespace thr

template <int R_, int TB_, int D_, int NV_>
static int launch_hr(const float *planes, const void *weights, const void *head_w, const float *bias, float *out, float *out_heads,
                     const int *rows, const int *n_rows, int G, int n_layers, const float (&sc)[4], yy_stream_t s,
                     int gate_lo = -1, int gate_hi = 0x7FFFFFFF) {
    thr::k_tower_h3r<R_, TB_, D_, NV_><<<dim3((G + TB_ - 1) / TB_), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, (const unsigned char *)head_w, bias, out, out_heads, rows, n_rows, G, n_layers, sc[0],
        sc[1], sc[2], sc[3], gate_lo, gate_hi);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_f16x3_regs: launch failed");
    return YY_OK;
}

// weights: wave-major f16 chunks [9 + 36*(n_layers-1)][8192] (network.pack_tower_h3r); head_w f16 [2][8192] (pack_heads_h3r) or
// NULL with out_heads NULL; bias f32 [n_layers (+1), 128]; planes f32 [G,5,R,R]; out f32 [G,R,R,128] or out_heads f32 [G,2,32,R*R].
extern "C" int yy_nn_tower_f16x3_regs(const float *planes, const void *weights, const void *head_w, const float *bias, float *out,
                                      float *out_heads, const int32_t *rows, const int32_t *n_rows, int G, int R, int C,
                                      int channels, int n_layers, int weight_exp, int head_exp, int act_exp, yy_stream_t s) {
    if (G == 0) return YY_OK;
    // {input scale 2^ka, accumulator scale 2^-kw, head accumulator scale 2^-(kh+ka), output scale 2^-ka}
    const float sc[4] = {ldexpf(1.0f, act_exp), ldexpf(1.0f, -weight_exp), ldexpf(1.0f, -(head_exp + act_exp)), ldexpf(1.0f, -act_exp)};
    if (!planes || !weights || !bias || (!out && !out_heads) || (out_heads && !head_w) || G < 0 || (rows && !n_rows))
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_f16x3_regs: bad argument");
    if (R != C || (R != 6 && R != 8 && R != 12) || channels != HR_CH || n_layers < 1 ||
        n_layers + (out_heads ? 1 : 0) > HR_MAX_LAYERS || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED,
                                "yy_nn_tower_f16x3_regs: needs 6x6, 8x8 or 12x12 boards, 128 channels, at most 10 residual blocks");
    if (R == 8) return launch_hr<8, 2, 9, 4>(planes, weights, head_w, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
    if (R == 6) return launch_hr<6, 4, 3, 2>(planes, weights, head_w, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
    return launch_hr<12, 1, 3, 2>(planes, weights, head_w, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
}

// yy_tower_h3q.hip -- the split-f16 (float32-accurate) tower of yy_tower_h3.hip in the "output-channel quarter" form of
// yy_towerq.hip, for the board sizes whose cells do not fill whole 64-cell wave tiles: 6x6 (BASELINE config 1, four boards
// per workgroup) and 12x12 (config 4, one board per workgroup).  Both are 144 (board, cell) columns = five 32-column MFMA
// tiles, the last one half padding (the pad columns read the zero row and are never stored: 10 % of the MFMAs).
//
// Numerics are those described in yy_tower_h3.hip: x = hi + lo with hi, lo float16 (weights stored times 2^kw, activations
// times 2^ka); w_hi*x_hi into one f32 accumulator, w_lo*x_hi + w_hi*x_lo into a second (v_mfma_f32_32x32x16_f16), f32 bias / residual /
// ReLU, re-split.  Same weight values as yy_tower_h3r.hip (network.pack_tower_h3 / pack_heads_h3; that kernel reads them in
// wave-major order), same accumulation order per output element, so a board's bits do not depend on which kernel or
// workgroup evaluated it.
// Wave w owns output channels [32w, 32w+32) for ALL columns: 5 x {acc1, acc2} accumulators (160 registers) + the f32 residual
// (80); per k-step 10 activation + 2 weight fragment reads for 15 MFMAs.  LDS: 2 x (144 + 1 zero) rows x 272 B (77 KB) + 4-slot x 16 KB
// ring + bias = 152.5 KB.  Reference: src/yin_yang/ai/neural_network.py:16-33, 94-119 (float32 on the CPU).
#include <hip/hip_runtime.h>
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

#define HQ_CH 128
#define HQ_ROW_BYTES 272
#define HQ_CHUNK_BYTES 16384
#define HQ_MAX_LAYERS 22                     // bias rows: 21 tower layers + 1 head row

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace thq {

template <int R_, int TB_, int NSLOT_> struct Geo {
    static constexpr int R = R_, TB = TB_, NSLOT = NSLOT_, CELLS = R_ * R_, NCOL = TB_ * R_ * R_, CT = (NCOL + 31) / 32;
    // one part (hi or lo) of every column + a zero row behind it: the lo row of ANY hi row (the zero row included) sits
    // PART_BYTES further, so a lane keeps one base per tile and the lo read is an immediate offset
    static constexpr int PART_BYTES = (NCOL + 1) * HQ_ROW_BYTES;
    static constexpr int ZERO_OFF = NCOL * HQ_ROW_BYTES;
    static constexpr int ACT_BYTES = 2 * PART_BYTES;
    static constexpr int RING_OFF = ACT_BYTES;
    static constexpr int BIAS_OFF = RING_OFF + NSLOT * HQ_CHUNK_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + HQ_MAX_LAYERS * HQ_CH * 4;
    static_assert(PART_BYTES + 256 < 65536, "lo offset must fit the ds_read immediate");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
template <class GEO>
__device__ __forceinline__ void issue_chunk(const unsigned char *wchunk, unsigned char *lds, int slot, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int piece = (r * 4 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + piece + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + GEO::RING_OFF + slot * HQ_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
// two f32 -> packed (hi, hi) and (lo, lo) f16 pairs: hi = f16(x) (round to nearest even), lo = f16(x - hi); x - hi is exact
// in f32 (written as fma(f32(hi), -1, x): one v_fma_mix_f32 per element)
__device__ __forceinline__ void split_pair(const f32x2 a, uint32_t &hi, uint32_t &lo) {
    const f16x2 h = __builtin_convertvector(a, f16x2);
    const f32x2 r = {__builtin_fmaf((float)h.x, -1.0f, a.x), __builtin_fmaf((float)h.y, -1.0f, a.y)};
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ f32x2 join_pair(const uint32_t hi, const uint32_t lo) {
    const f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
    const f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
    return h + l;
}
// Ring protocol of one chunk.  The chunk layout [ks 2][part 2][nt 4][1 KB] and the piece order of issue_chunk make wave w
// load exactly the four 1 KB pieces (nt == w) that wave w itself reads: the weight ring is WAVE-PRIVATE, so a chunk costs
// the wave one counted vmcnt wait and NO workgroup barrier (barriers remain only around the epilogue, where the waves
// exchange activations).  Invariant on entry: the wave's pieces of `chunk` have landed.  Waits for its pieces of chunk+1
// (the last k-step of a chunk prefetches the next chunk's first fragments), then refills the slot chunk-1 used -- every read of
// that slot by this wave has returned (its fragments were consumed by MFMAs earlier in program order).
template <class GEO>
__device__ __forceinline__ void chunk_sync(const unsigned char *weights, unsigned char *lds, int chunk, int n_chunks, bool first,
                                           int wave, int lane) {
    constexpr int AHEAD = GEO::NSLOT - 1;                 // chunks issued beyond the current one
    if (chunk + 1 < n_chunks) {
        const int newer = min(AHEAD - 2, n_chunks - 2 - chunk);   // chunks younger than chunk+1 that may stay in flight
        if (newer >= 2) wait_vmcnt<8>();
        else if (newer == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    }
    if (first) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the previous layer's epilogue (or the prologue) is visible to every wave
        asm volatile("" ::: "memory");
    }
    if (chunk + AHEAD < n_chunks)
        issue_chunk<GEO>(weights + (size_t)(chunk + AHEAD) * HQ_CHUNK_BYTES, lds, (chunk + AHEAD) % GEO::NSLOT, wave, lane);
}

template <int CT> struct Frags {
    f16x8 xh[CT], xl[CT], wh, wl;
};
template <class GEO>
__device__ __forceinline__ void load_frags(Frags<GEO::CT> &f, const unsigned char *lds, int slot, int quarter, int ks,
                                           const uint32_t (&cbase)[GEO::CT], int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    // chunk: [ks 2][part 2][nt 4][h 2][c 32][j 8]; this wave's cout tile is nt = nh
    const unsigned char *wslot = lds + GEO::RING_OFF + slot * HQ_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 8192 + nh * 1024;
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) {
        f.xh[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cbase[tt] + quarter * 64 + ks * 32));
        f.xl[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cbase[tt] + GEO::PART_BYTES + quarter * 64 + ks * 32));
    }
    f.wh = __builtin_bit_cast(f16x8, *(const u32x4 *)wslot);
    f.wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(wslot + 4096));
}
// acc1 += w_hi * x_hi ;  acc2 += w_lo * x_hi + w_hi * x_lo   (same order as yy_tower_h3r.hip; why two accumulators: there)
template <int CT, bool ZERO>
__device__ __forceinline__ void mma_ct(f32x16 (&acc1)[CT], f32x16 (&acc2)[CT], const Frags<CT> &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < CT; tt++) {
        const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl, f.xh[tt], ZERO ? z : acc2[tt], 0, 0, 0);
        acc1[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xh[tt], ZERO ? z : acc1[tt], 0, 0, 0);
        acc2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xl[tt], a, 0, 0, 0);
    }
}
// the 2*CT + 2 reads of the next k-step inside the 3*CT MFMAs of this one
template <int CT> __device__ __forceinline__ void interleave_hint() {
    constexpr int NR = 2 * CT + 2, PAIRS = NR / 2, REST = 3 * CT - PAIRS;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
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
        g.rowbase[tt] = (uint32_t)(col * HQ_ROW_BYTES) + (uint32_t)(h * 16);
        uint32_t m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
            if (((unsigned)sy < (unsigned)GEO::R) && ((unsigned)sx < (unsigned)GEO::R)) m |= 1u << tap;
        }
        g.okmask[tt] = (col < GEO::NCOL) ? m : 0u;
    }
}
template <class GEO>
__device__ __forceinline__ void tap_geo(int tap, const LaneGeo<GEO> &g, uint32_t (&cbase)[GEO::CT]) {
    const int shift = ((tap / 3 - 1) * GEO::R + (tap % 3 - 1)) * HQ_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) cbase[tt] = (g.okmask[tt] & bit) ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
}

template <class GEO, bool STEM>
__device__ __forceinline__ void run_layer(f32x16 (&acc1)[GEO::CT], f32x16 (&acc2)[GEO::CT], unsigned char *lds,
                                          const unsigned char *weights, int &chunk, int n_chunks, const LaneGeo<GEO> &geo, int nh,
                                          int wave, int lane) {
    constexpr int Q = STEM ? 1 : 4, KS = STEM ? 1 : 2, NCH = 9 * Q, CT = GEO::CT;
    uint32_t cb[CT];
    tap_geo<GEO>(0, geo, cb);
    Frags<CT> cur;
    for (int i = 0; i < NCH; i++, chunk++) {
        const int quarter = STEM ? 0 : (i & 3);
        chunk_sync<GEO>(weights, lds, chunk, n_chunks, i == 0, wave, lane);
        if (i == 0) load_frags<GEO>(cur, lds, chunk % GEO::NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        const int ni = last ? i : i + 1;
        uint32_t ncb[CT];
        tap_geo<GEO>(STEM ? ni : (ni >> 2), geo, ncb);
        const int nquarter = STEM ? 0 : (ni & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags<CT> nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags<GEO>(nxt, lds, chunk % GEO::NSLOT, quarter, ks + 1, cb, nh, lane);
            else if (!last) load_frags<GEO>(nxt, lds, (chunk + 1) % GEO::NSLOT, nquarter, 0, ncb, nh, lane);
            if (i == 0 && ks == 0) mma_ct<CT, true>(acc1, acc2, cur);
            else mma_ct<CT, false>(acc1, acc2, cur);
            if (has_next) {
                interleave_hint<CT>();
                cur = nxt;
            }
        }
#pragma unroll
        for (int tt = 0; tt < CT; tt++) cb[tt] = ncb[tt];   // (same tap while the quarter advances)
    }
}

template <int R_, int TB_, int NSLOT_>
__global__ void __launch_bounds__(256, 1)
k_tower_h3q(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
            float *__restrict__ out, float *__restrict__ out_heads, const int *__restrict__ rows,
            const int *__restrict__ n_rows, int G, int n_layers, float in_scale, float acc_scale, float head_scale,
            float out_scale, int gate_lo, int gate_hi) {
    using GEO = Geo<R_, TB_, NSLOT_>;
    constexpr int CT = GEO::CT, CELLS = GEO::CELLS, NCOL = GEO::NCOL, TB = GEO::TB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GEO::LDS_BYTES];
    const int n_live = n_rows ? min(*n_rows, G) : G;
    if (n_live <= gate_lo || n_live > gate_hi) return;     // see yy_nn_tower_heads_f16x3_auto
    const int g0 = blockIdx.x * TB;                        // first dense row of the workgroup
    if (g0 >= n_live) return;                              // whole workgroup, before any barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nh = wave;                                   // this wave's output-channel quarter
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * HQ_CH; i += 256)
        ((float *)(lds + GEO::BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 128)   // the zero rows of both parts
        ((uint32_t *)(lds + GEO::ZERO_OFF + (threadIdx.x >> 6) * GEO::PART_BYTES))[threadIdx.x & 63] = 0u;
    for (int col = threadIdx.x; col < NCOL; col += 256) {  // 5 planes -> channels 0..4 of a 16-channel zero-padded input
        const int gb = g0 + col / CELLS, cell = col % CELLS;
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
        *(u32x4 *)(lds + col * HQ_ROW_BYTES) = (u32x4){hi[0], hi[1], hi[2], 0u};
        *(u32x4 *)(lds + col * HQ_ROW_BYTES + 16) = z;
        *(u32x4 *)(lds + GEO::PART_BYTES + col * HQ_ROW_BYTES) = (u32x4){lo[0], lo[1], lo[2], 0u};
        *(u32x4 *)(lds + GEO::PART_BYTES + col * HQ_ROW_BYTES + 16) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_tower = 9 + 36 * (n_layers - 1);
    const int n_chunks = n_tower + (out_heads ? 2 : 0);
#pragma unroll
    for (int pc = 0; pc < GEO::NSLOT - 1; pc++)
        if (pc < n_chunks) issue_chunk<GEO>(weights + (size_t)pc * HQ_CHUNK_BYTES, lds, pc % GEO::NSLOT, wave, lane);
    if (n_chunks >= GEO::NSLOT - 1) wait_vmcnt<4 * (GEO::NSLOT - 2)>();   // this wave's pieces of chunk 0
    else wait_vmcnt<0>();

    LaneGeo<GEO> geo;
    make_lane_geo<GEO>(geo, c, h);
    f32x4 res[CT][4];      // residual x of this wave's 32 couts, f32
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc1[CT], acc2[CT];
        if (L == 0) run_layer<GEO, true>(acc1, acc2, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        else run_layer<GEO, false>(acc1, acc2, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has finished reading this layer's input
        asm volatile("" ::: "memory");
        auto epilogue = [&](auto conv2_tag, auto keep_tag) {
            constexpr bool conv2 = decltype(conv2_tag)::value, keep = decltype(keep_tag)::value;   // resolved once per layer
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const f32x4 *)(lds + GEO::BIAS_OFF + (L * HQ_CH + nh * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = nh * 32 + 8 * q + 4 * h;       // this lane's 4 couts
                const f32x4 b = bq[q];
#pragma unroll
                for (int tt = 0; tt < CT; tt++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = __builtin_fmaf(acc1[tt][4 * q + i] + acc2[tt][4 * q + i], acc_scale, b[i]);
                    if (conv2) v += res[tt][q];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.0f);
                    if (keep) res[tt][q] = v;
                    uint32_t h01, l01, h23, l23;
                    split_pair((f32x2){v[0], v[1]}, h01, l01);
                    split_pair((f32x2){v[2], v[3]}, h23, l23);
                    const int col = tt * 32 + c;
                    if (NCOL % 32 == 0 || col < NCOL) {
                        *(u32x2 *)(lds + col * HQ_ROW_BYTES + co * 2) = (u32x2){h01, h23};
                        *(u32x2 *)(lds + GEO::PART_BYTES + col * HQ_ROW_BYTES + co * 2) = (u32x2){l01, l23};
                    }
                }
            }
        };
        if (L == 0) epilogue(std::false_type{}, std::true_type{});                    // stem: a block input
        else if ((L & 1) == 0) epilogue(std::true_type{}, std::true_type{});          // second conv of a block: + skip, keep
        else epilogue(std::false_type{}, std::false_type{});
    }
    if (out_heads) {
        // 1x1 head convs: two chunks [ks 4][part 2][nt 2][h 2][c 32][j 8]; wave w: head (w & 1), column tiles
        // [t0, t0 + HT) with t0 = (w >> 1) * HT (a surplus tile of waves 2, 3 is a clamped duplicate, never stored)
        constexpr int HT = (CT + 1) / 2;
        const int head = wave & 1, t0 = (wave >> 1) * HT, nt_cnt = (wave >> 1) ? CT - HT : HT;
        f32x16 h1[HT], h2[HT];
        uint32_t xb[HT];
#pragma unroll
        for (int t = 0; t < HT; t++) xb[t] = (uint32_t)(min((t0 + t) * 32 + c, NCOL - 1) * HQ_ROW_BYTES + h * 16);
#pragma unroll
        for (int hc = 0; hc < 2; hc++, chunk++) {
            chunk_sync<GEO>(weights, lds, chunk, n_chunks, hc == 0, wave, lane);
            const unsigned char *hw = lds + GEO::RING_OFF + (chunk % GEO::NSLOT) * HQ_CHUNK_BYTES + (h * 32 + c) * 16 + head * 1024;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const f16x8 wh = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 4096));
                const f16x8 wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 4096 + 2048));
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    const f16x8 xh = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[t] + hc * 128 + ks * 32));
                    const f16x8 xl = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[t] + GEO::PART_BYTES + hc * 128 + ks * 32));
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const bool zero = (hc == 0 && ks == 0);
                    const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, zero ? z : h2[t], 0, 0, 0);
                    h1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, zero ? z : h1[t], 0, 0, 0);
                    h2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a, 0, 0, 0);
                }
            }
        }
        // features f32 [row][head][channel 32][cell CELLS] (the reference's NCHW flatten order)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(lds + GEO::BIAS_OFF + (n_layers * HQ_CH + head * 32 + 8 * q + 4 * h) * 4);
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int p = threadIdx.x; p < NCOL * 32; p += 256) {   // activations [column][128] f32 = hi + lo * 2^-11, 4 channels per piece
        const int col = p >> 5, ch4 = p & 31;
        if (g0 + col / CELLS < n_live) {
            const u32x2 ph = *(const u32x2 *)(lds + col * HQ_ROW_BYTES + ch4 * 8);
            const u32x2 pl = *(const u32x2 *)(lds + GEO::PART_BYTES + col * HQ_ROW_BYTES + ch4 * 8);
            const f32x2 v01 = join_pair(ph.x, pl.x), v23 = join_pair(ph.y, pl.y);
            *(f32x4 *)(out + ((size_t)g0 * CELLS + col) * HQ_CH + ch4 * 4) = (f32x4){v01.x, v01.y, v23.x, v23.y} * out_scale;
        }
    }
}

}   // namespace thq

template <int R_, int TB_, int NSLOT_>
static int launch_hq(const float *planes, const void *weights, const float *bias, float *out, float *out_heads, const int *rows,
                     const int *n_rows, int G, int n_layers, const float *sc, yy_stream_t s, int gate_lo = -1,
                     int gate_hi = 0x7FFFFFFF) {
    thq::k_tower_h3q<R_, TB_, NSLOT_><<<dim3((G + TB_ - 1) / TB_), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, bias, out, out_heads, rows, n_rows, G, n_layers, sc[0], sc[1], sc[2], sc[3],
        gate_lo, gate_hi);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_f16x3: launch failed");
    return YY_OK;
}

// called by yy_tower_h3.hip: R = 6 (four boards per workgroup), 12 (one board per workgroup), 8 (two boards per workgroup)
extern "C" int yy_tower_h3q_launch(const float *planes, const void *weights, const float *bias, float *out, float *out_heads,
                                   const int *rows, const int *n_rows, int G, int R, int n_layers, const float *sc, yy_stream_t s) {
    if (n_layers + (out_heads ? 1 : 0) > HQ_MAX_LAYERS) return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_f16x3: too many layers");
    if (R == 6) return launch_hq<6, 4, 4>(planes, weights, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
    if (R == 12) return launch_hq<12, 1, 4>(planes, weights, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
    if (R == 8) {
        // small batches (arena matches, single-board MCTS.search): one board per workgroup spreads the work over twice as
        // many CUs while the chip is not full; same bits
        if (G <= 256) return launch_hq<8, 1, 5>(planes, weights, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
        return launch_hq<8, 2, 5>(planes, weights, bias, out, out_heads, rows, n_rows, G, n_layers, sc, s);
    }
    return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_f16x3: board size");
}

// 8x8, one board per workgroup, with a device-side gate on the live row count (yy_nn_tower_heads_f16x3_auto)
extern "C" int yy_tower_h3q_launch81_gated(const float *planes, const void *weights, const float *bias, float *out_heads,
                                           const int *rows, const int *n_rows, int G, int n_layers, const float *sc,
                                           int gate_lo, int gate_hi, yy_stream_t s) {
    if (n_layers + 1 > HQ_MAX_LAYERS) return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_f16x3: too many layers");
    return launch_hq<8, 1, 5>(planes, weights, bias, nullptr, out_heads, rows, n_rows, G, n_layers, sc, s, gate_lo, gate_hi);
}

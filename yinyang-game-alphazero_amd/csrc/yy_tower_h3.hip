// yy_tower_h3.hip -- the LDS-resident tower kernel (design: yy_tower.hip) at FLOAT32 accuracy on the F16 matrix cores
// ("split-f16", evaluator mode "f16x3"): the evaluator the benchmark and the reference-facing APIs default to.
//
// Reference computation: YinYangNeuralNetwork.forward, stem + residual blocks + the two 1x1 head convolutions
// (src/yin_yang/ai/neural_network.py:16-33, 94-119) in float32 on the CPU.  gfx950's f32-input MFMA runs at 1/16 of the
// f16 rate, so every activation x and every weight w is held as TWO f16 numbers
//     hi = f16(x),  lo = f16((x - hi) * 2^11)        (x == hi + lo * 2^-11 to 22 significant bits; f32 has 24)
// and each product is formed as  w_hi*x_hi  (accumulator 1)  +  2^-11 * (w_hi*x_lo + w_lo*x_hi)  (accumulator 2) with three
// v_mfma_f32_32x32x16_f16; the dropped w_lo*x_lo term is 2^-22 relative.  The 2^11 scaling keeps the lo parts in f16's
// normal range (an unscaled lo of a weight ~0.02 would be a subnormal f16 and lose its bits).  Bias, residual (exact f32 in
// registers) and ReLU are applied in f32 and the result is split again for the next layer.  Measured against a float64
// evaluation of the same folded network the activations agree to 3.7e-7 of scale -- the float32 module itself is at
// 3.2e-7 -- where the bf16 split (yy_tower_x3.hip: 16 significant bits) is at 8.9e-6.
//
// 8x8 boards, 128 channels.  A (board, part) pair is laid out exactly like a board of yy_tower.hip (64 cells x 272 B), so a
// workgroup holds TWO boards = 4 "virtual boards"; wave w owns board (w & 1) and output-channel half (w >> 1): a 64-cout x
// 64-cell tile = 2 x 2 x {acc1, acc2} accumulators.  A weight chunk is one tap x 32 input channels x 128 couts x {hi, lo} =
// 16 KB in fragment order [ks 2][part 2][nt 4][h 2][c 32][j 8]; per k-step a wave reads 4 activation + 4 weight fragments for
// 12 MFMAs.  5-slot LDS-DMA ring, one barrier per chunk, two extra barriers per layer around the epilogue.  The two head
// convolutions ride the ring as two more chunks ([ks 4][part 2][nt 2][h 2][c 32][j 8] each, cin = 64*chunk + 16*ks + 8*h + j)
// and leave float32 features [board][head 2][channel 32][cell 64] = the reference's NCHW flatten order (:114 / :119).
// Optional row gather: with (rows, n_rows) the workgroup evaluates planes[rows[i]] for i < *n_rows into dense output row i
// and workgroups past *n_rows exit at once -- the lockstep step compacts the leaves that need an evaluation (terminal
// revisits do not) so the launch carries no dead rows.
// Roofline: f16 MFMA; algorithmic FLOPs are the conv's (one product per term), issued MFMA FLOPs are 3x that.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define H_TB 2
#define H_CH 128
#define H_CELLS 64
#define H_ROW_BYTES 272                                      // 128 f16 channels + 16 B pad
#define H_ACT_BYTES (H_TB * 2 * H_CELLS * H_ROW_BYTES)       // 69632: [board 2][part 2][cell 64] rows
#define H_CHUNK_BYTES 16384
#define H_NSLOT 5
#define H_RING_OFF H_ACT_BYTES
#define H_BIAS_OFF (H_RING_OFF + H_NSLOT * H_CHUNK_BYTES)
#define H_MAX_LAYERS 23                                      // bias rows: tower layers + 1 head row
#define H_ZERO_OFF (H_BIAS_OFF + H_MAX_LAYERS * H_CH * 4)
#define H_LDS_BYTES (H_ZERO_OFF + 256)                        // 163584
#define H_LO_SCALE 2048.0f                                    // 2^11
#define H_LO_INV 0.00048828125f                               // 2^-11

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace th3 {

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
__device__ __forceinline__ void issue_chunk(const unsigned char *wchunk, unsigned char *lds, int slot, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int piece = (r * 4 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + piece + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + H_RING_OFF + slot * H_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
// byte offset of the row of (board, part, cell)
__device__ __forceinline__ uint32_t row_off(int board, int part, int cell) {
    return (uint32_t)(((board * 2 + part) * H_CELLS + cell) * H_ROW_BYTES);
}
// two f32 -> packed (hi, hi) and (lo, lo) f16 pairs: hi = f16(x) (round to nearest even), lo = f16((x - hi) * 2^11);
// x - hi is exact in f32
__device__ __forceinline__ void split_pair(const f32x2 a, uint32_t &hi, uint32_t &lo) {
    const f16x2 h = __builtin_convertvector(a, f16x2);
    const f32x2 r = (a - __builtin_convertvector(h, f32x2)) * H_LO_SCALE;
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
// packed (hi, hi), (lo, lo) -> two f32: hi + lo * 2^-11
__device__ __forceinline__ f32x2 join_pair(const uint32_t hi, const uint32_t lo) {
    const f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
    const f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
    return (f32x2){__builtin_fmaf(l.x, H_LO_INV, h.x), __builtin_fmaf(l.y, H_LO_INV, h.y)};
}

// The ring protocol of one chunk.  Invariant on entry: chunk `chunk` is visible in its slot to every wave (or becomes so at
// the barrier below when `first`).  Makes chunk+1 visible (counted vmcnt + barrier), then refills the slot chunk-1 used.
__device__ __forceinline__ void chunk_sync(const unsigned char *weights, unsigned char *lds, int chunk, int n_chunks, bool first,
                                           int wave, int lane) {
    if (chunk + 1 < n_chunks) {
        const int newer = min(2, n_chunks - 2 - chunk);   // chunks younger than chunk+1 still in flight
        if (newer == 2) wait_vmcnt<8>();
        else if (newer == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    }
    if (chunk + 1 < n_chunks || first) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // chunk+1 landed everywhere; everyone finished chunk-1 (and the epilogue)
        asm volatile("" ::: "memory");
    }
    if (chunk + 4 < n_chunks)
        issue_chunk(weights + (size_t)(chunk + 4) * H_CHUNK_BYTES, lds, (chunk + 4) % H_NSLOT, wave, lane);
}

// one k-step = 16 input channels of one tap: hi/lo activation fragments of both column tiles, hi/lo weight fragments of
// this wave's two cout tiles
struct Frags {
    f16x8 xh[2], xl[2], wh[2], wl[2];
};
struct Bases {
    uint32_t hi[2], lo[2];   // byte offsets (+ h*16) of the hi and lo rows of this lane's tap neighbour, or the zero row
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int quarter, int ks,
                                           const Bases &cbase, int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    // chunk: [ks 2][part 2][nt 4][h 2][c 32][j 8]; this wave's tiles are nt = 2*nh, 2*nh + 1
    const unsigned char *wslot = lds + H_RING_OFF + slot * H_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 8192 + nh * 2048;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        f.xh[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cbase.hi[tt] + quarter * 64 + ks * 32));
        f.xl[tt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cbase.lo[tt] + quarter * 64 + ks * 32));
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        f.wh[nt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(wslot + nt * 1024));
        f.wl[nt] = __builtin_bit_cast(f16x8, *(const u32x4 *)(wslot + 4096 + nt * 1024));
    }
}
// acc1 += w_hi * x_hi ;  acc2 += w_lo * x_hi + w_hi * x_lo   (acc2 carries the 2^11 scale of the lo parts)
template <bool ZERO> __device__ __forceinline__ void mma12(f32x16 (&acc1)[2][2], f32x16 (&acc2)[2][2], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl[nt], f.xh[tt], ZERO ? z : acc2[tt][nt], 0, 0, 0);
            acc1[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[nt], f.xh[tt], ZERO ? z : acc1[tt][nt], 0, 0, 0);
            acc2[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[nt], f.xl[tt], a, 0, 0, 0);
        }
}
// per-lane geometry, computed once per kernel: own rows (hi part; the lo part sits 64 rows further), on-board tap mask
struct LaneGeo {
    uint32_t rowbase[2], okmask[2], zbase;
};
__device__ __forceinline__ LaneGeo make_lane_geo(const int (&cy)[2], int cx, int board, int h) {
    LaneGeo g;
    g.zbase = (uint32_t)H_ZERO_OFF + (uint32_t)(h * 16);
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        g.rowbase[tt] = row_off(board, 0, cy[tt] * 8 + cx) + (uint32_t)(h * 16);
        uint32_t m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int sy = cy[tt] + tap / 3 - 1, sx = cx + tap % 3 - 1;
            if (((unsigned)sy < 8u) && ((unsigned)sx < 8u)) m |= 1u << tap;
        }
        g.okmask[tt] = m;
    }
    return g;
}
__device__ __forceinline__ void tap_geo(int tap, const LaneGeo &g, Bases &cbase) {
    const int shift = ((tap / 3 - 1) * 8 + (tap % 3 - 1)) * H_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        const bool ok = (g.okmask[tt] & bit) != 0;
        cbase.hi[tt] = ok ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
        cbase.lo[tt] = ok ? g.rowbase[tt] + (uint32_t)(shift + H_CELLS * H_ROW_BYTES) : g.zbase;
    }
}
// the 8 reads of the next k-step inside the 12 MFMAs of this one
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
}

// A layer = 9 taps x Q quarter-chunks (Q = 4, two k-steps each; stem: 1 chunk per tap, one k-step = 16 padded channels).
template <bool STEM>
__device__ __forceinline__ void run_layer(f32x16 (&acc1)[2][2], f32x16 (&acc2)[2][2], unsigned char *lds,
                                          const unsigned char *weights, int &chunk, int n_chunks, const LaneGeo &geo, int nh,
                                          int wave, int lane) {
    constexpr int Q = STEM ? 1 : 4, KS = STEM ? 1 : 2, NCH = 9 * Q;
    Bases cb;
    tap_geo(0, geo, cb);
    Frags cur;
    for (int i = 0; i < NCH; i++, chunk++) {
        const int quarter = STEM ? 0 : (i & 3);
        chunk_sync(weights, lds, chunk, n_chunks, i == 0, wave, lane);
        if (i == 0) load_frags(cur, lds, chunk % H_NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        const int ni = last ? i : i + 1;
        Bases ncb;
        tap_geo(STEM ? ni : (ni >> 2), geo, ncb);
        const int nquarter = STEM ? 0 : (ni & 3);
        const bool next_tap = STEM || (quarter == 3);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % H_NSLOT, quarter, ks + 1, cb, nh, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % H_NSLOT, nquarter, 0, next_tap ? ncb : cb, nh, lane);
            if (i == 0 && ks == 0) mma12<true>(acc1, acc2, cur);
            else mma12<false>(acc1, acc2, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
        if (next_tap) cb = ncb;
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower_h3(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
           float *__restrict__ out, float *__restrict__ out_heads, const int *__restrict__ rows,
           const int *__restrict__ n_rows, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[H_LDS_BYTES];
    const int n_live = n_rows ? min(*n_rows, G) : G;          // rows this launch has to evaluate (uniform)
    if ((int)blockIdx.x * H_TB >= n_live) return;             // whole workgroup: nothing to do, before any barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int board = wave & 1, nh = wave >> 1;
    const int gb = blockIdx.x * H_TB + board;                  // dense output row
    const bool live = gb < n_live;
    const int src = live ? (rows ? rows[gb] : gb) : 0;         // input row
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * H_CH; i += 256) ((float *)(lds + H_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + H_ZERO_OFF))[threadIdx.x] = 0u;
    {   // wave (board, nh): part nh of the input; lane = cell: 5 planes -> channels 0..4 of a 16-channel zero-padded input
        float p[6];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = live ? planes[((size_t)src * 5 + k) * H_CELLS + lane] : 0.0f;
        p[5] = 0.0f;
        uint32_t hi[3], lo[3];
#pragma unroll
        for (int k = 0; k < 3; k++) split_pair((f32x2){p[2 * k], p[2 * k + 1]}, hi[k], lo[k]);
        const uint32_t *s = nh ? lo : hi;
        const u32x4 v0 = {s[0], s[1], s[2], 0u};
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + row_off(board, nh, lane)) = v0;
        *(u32x4 *)(lds + row_off(board, nh, lane) + 16) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_tower = 9 + 36 * (n_layers - 1);
    const int n_chunks = n_tower + (out_heads ? 2 : 0);
#pragma unroll
    for (int pc = 0; pc < 4; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * H_CHUNK_BYTES, lds, pc % H_NSLOT, wave, lane);
    if (n_chunks >= 4) wait_vmcnt<12>();
    else wait_vmcnt<0>();

    const int cy[2] = {c >> 3, 4 + (c >> 3)}, cx = c & 7;
    const LaneGeo geo = make_lane_geo(cy, cx, board, h);
    f32x4 res[2][2][4];   // residual x of this wave's 64 couts, f32
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc1[2][2], acc2[2][2];
        if (L == 0) run_layer<true>(acc1, acc2, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        else run_layer<false>(acc1, acc2, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has finished reading this layer's input
        asm volatile("" ::: "memory");
        const bool conv2 = (L >= 2) && ((L & 1) == 0);
        const bool keep = (L == 0) || conv2;
        // the lane's 8 bias vectors in one batch of back-to-back reads (no MFMA runs here: every stall is paid in full)
        f32x4 bq[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                bq[nt][q] = *(const f32x4 *)(lds + H_BIAS_OFF + (L * H_CH + (nh * 2 + nt) * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = (nh * 2 + nt) * 32 + 8 * q + 4 * h;   // this lane's 4 couts (accumulator rows 4q..4q+3)
                const f32x4 b = bq[nt][q];
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        v[i] = __builtin_fmaf(acc2[tt][nt][4 * q + i], H_LO_INV, acc1[tt][nt][4 * q + i]) + b[i];
                    if (conv2) v += res[tt][nt][q];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.0f);
                    if (keep) res[tt][nt][q] = v;
                    uint32_t h01, l01, h23, l23;
                    split_pair((f32x2){v[0], v[1]}, h01, l01);
                    split_pair((f32x2){v[2], v[3]}, h23, l23);
                    const u32x2 ph = {h01, h23}, pl = {l01, l23};
                    const int cell = tt * 32 + c;
                    *(u32x2 *)(lds + row_off(board, 0, cell) + co * 2) = ph;
                    *(u32x2 *)(lds + row_off(board, 1, cell) + co * 2) = pl;
                }
            }
    }
    if (out_heads) {
        // ---- policy_conv / value_conv (1x1, 128 -> 32 each; neural_network.py:59-60, 65-66, 113, 118) + bias + ReLU:
        // wave (board, nh): head nh of its board, 32 couts x 64 cells, two chunks of 64 input channels
        f32x16 h1[2], h2[2];
        const uint32_t xb[2] = {row_off(board, 0, c) + (uint32_t)(h * 16), row_off(board, 0, 32 + c) + (uint32_t)(h * 16)};
#pragma unroll
        for (int hc = 0; hc < 2; hc++, chunk++) {
            chunk_sync(weights, lds, chunk, n_chunks, hc == 0, wave, lane);
            const unsigned char *hw = lds + H_RING_OFF + (chunk % H_NSLOT) * H_CHUNK_BYTES + (h * 32 + c) * 16 + nh * 1024;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const f16x8 wh = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 4096));
                const f16x8 wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(hw + ks * 4096 + 2048));
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    const f16x8 xh = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[tt] + hc * 128 + ks * 32));
                    const f16x8 xl =
                        __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb[tt] + H_CELLS * H_ROW_BYTES + hc * 128 + ks * 32));
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const bool zero = (hc == 0 && ks == 0);
                    const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, zero ? z : h2[tt], 0, 0, 0);
                    h1[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, zero ? z : h1[tt], 0, 0, 0);
                    h2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a, 0, 0, 0);
                }
            }
        }
        if (live) {   // features f32 [row][head][channel 32][cell 64]: a wave store = two 128-B runs of cells
            float *o = out_heads + ((size_t)gb * 2 + nh) * 32 * H_CELLS;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b = *(const f32x4 *)(lds + H_BIAS_OFF + (n_layers * H_CH + nh * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = fmaxf(__builtin_fmaf(h2[tt][4 * q + i], H_LO_INV, h1[tt][4 * q + i]) + b[i], 0.0f);
                        o[(8 * q + 4 * h + i) * H_CELLS + tt * 32 + c] = v;
                    }
            }
        }
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (live) {   // activations [cell][128] f32 (channels-last) = hi + lo * 2^-11; 64 x 32 pieces of 4 channels per board, half per wave
        for (int p = nh * 1024 + lane; p < (nh + 1) * 1024; p += 64) {
            const int cell = p >> 5, ch4 = p & 31;
            const u32x2 ph = *(const u32x2 *)(lds + row_off(board, 0, cell) + ch4 * 8);
            const u32x2 pl = *(const u32x2 *)(lds + row_off(board, 1, cell) + ch4 * 8);
            const f32x2 v01 = join_pair(ph.x, pl.x), v23 = join_pair(ph.y, pl.y);
            const f32x4 v = {v01.x, v01.y, v23.x, v23.y};
            *(f32x4 *)(out + ((size_t)gb * H_CELLS + cell) * H_CH + ch4 * 4) = v;
        }
    }
}

}   // namespace th3

extern "C" int yy_tower_h3q_launch(const float *planes, const void *weights, const float *bias, float *out, float *out_heads,
                                   const int *rows, const int *n_rows, int G, int R, int n_layers, yy_stream_t s);   // yy_tower_h3q.hip

// 8x8 kernel form: 1 (default) = cout-quarter waves with wave-private weight rings (yy_tower_h3q.hip<8,2,5>: 3.56 ms per
// 4096-board launch), 0 = board x cout-half waves with the shared ring of this file (3.64 ms; kept as the A/B partner)
static int g_h3_form8 = 1;
extern "C" int yy_nn_tower_f16x3_set_form8(int form) {   // A/B measurements (tools/eval_micro.py); same bits either way
    g_h3_form8 = form ? 1 : 0;
    return YY_OK;
}

static int launch_h3(const float *planes, const void *weights, const float *bias, float *out, float *out_heads, const int *rows,
                     const int *n_rows, int G, int R, int C, int channels, int n_layers, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || (!out && !out_heads) || G < 0 || (rows && !n_rows))
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_f16x3: bad argument");
    if (R != C || (R != 6 && R != 8 && R != 12) || channels != H_CH || n_layers < 1 ||
        n_layers + (out_heads ? 1 : 0) > H_MAX_LAYERS || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED,
                                "yy_nn_tower_f16x3: needs 6x6, 8x8 or 12x12 boards, 128 channels, at most 10 residual blocks");
    if (R != 8 || g_h3_form8 == 1) return yy_tower_h3q_launch(planes, weights, bias, out, out_heads, rows, n_rows, G, R, n_layers, s);
    th3::k_tower_h3<<<dim3((G + H_TB - 1) / H_TB), dim3(256), 0, (hipStream_t)s>>>(planes, (const unsigned char *)weights, bias, out,
                                                                                 out_heads, rows, n_rows, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_f16x3: launch failed");
    return YY_OK;
}

// weights: f16 chunks [9 + 36*(n_layers-1)][8192] in fragment order (network.pack_tower_h3); bias f32 [n_layers,128];
// planes f32 [G,5,8,8]; out f32 [G,8,8,128].
extern "C" int yy_nn_tower_f16x3(const float *planes, const void *weights, const float *bias, float *out, int G, int R, int C,
                                 int channels, int n_layers, yy_stream_t s) {
    return launch_h3(planes, weights, bias, out, nullptr, nullptr, nullptr, G, R, C, channels, n_layers, s);
}

// the same + the two 1x1 head convolutions: weights hold two more chunks, bias one more row; out_heads f32 [G,2,32,64];
// rows / n_rows (device, or both NULL): evaluate planes[rows[i]] for i < *n_rows into out_heads row i.
extern "C" int yy_nn_tower_heads_f16x3(const float *planes, const void *weights, const float *bias, float *out_heads,
                                       const int32_t *rows, const int32_t *n_rows, int G, int R, int C, int channels,
                                       int n_layers, yy_stream_t s) {
    return launch_h3(planes, weights, bias, nullptr, out_heads, rows, n_rows, G, R, C, channels, n_layers, s);
}

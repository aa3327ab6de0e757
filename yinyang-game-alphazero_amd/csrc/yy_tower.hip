// yy_tower.hip -- the policy/value CNN's residual tower as ONE LDS-resident MFMA kernel (gfx950).
//
// Reference computation: YinYangNeuralNetwork.forward, stem + residual blocks
// (src/yin_yang/ai/neural_network.py:16-33, 94-110) at eval time with BatchNorm folded into the
// convolutions, on 8x8 boards with 128 channels, bf16 storage / f32 accumulation.
//
// MI355X design.  A 3x3 convolution on an 8x8 board never looks outside its board, so the whole
// tower of a board can run inside one workgroup with no inter-workgroup dependency:
//   * workgroup = 4 waves = 4 boards (wave w owns board w: its 64 cells are the MFMA columns);
//   * the boards' activations [64 cells][128 ch] bf16 live in LDS for ALL layers (68 KB, rows padded by
//     16 B so ds_read_b128 fragment reads are conflict-free and use immediate offsets); they never touch HBM;
//   * each layer is an implicit GEMM  D[cout][cell] = sum_{tap,cin} W[cout][tap,cin] * X[cin][cell+tap]
//     on v_mfma_f32_32x32x16_bf16 (weights = A operand, shifted activations = B operand; out-of-board
//     taps read a zero row); a wave accumulates its full 128 x 64 output tile in 128 accumulator
//     registers, then applies bias (pre-loaded into the accumulators), residual (kept packed in
//     registers) and ReLU and writes bf16 back to LDS;
//   * weights stream L2 -> LDS with global_load_lds_dwordx4 through a 5-slot ring of 16 KB chunks
//     (one chunk = one tap x 64 input channels, pre-packed on the host in fragment order), counted
//     s_waitcnt vmcnt + raw s_barrier, 4 chunks (~2 taps of compute) in flight;
//   * LDS: 68 KB activations (272-B padded rows) + 80 KB ring + 11.5 KB bias table + zero row = 160 KB -> 1 workgroup/CU,
//     one wave per SIMD with the whole register file.
// Roofline: MFMA (bf16 dense 2.5 PFLOP/s).  Algorithmic FLOPs per board: 2*9*16*128*64 (stem, K padded
// to 16) + (layers-1) * 2*9*128*128*64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TW_TB 4
#define TW_CH 128
#define TW_CELLS 64
#define TW_ROW_BYTES 272                                   // 256 B of channels + 16 B pad: consecutive cells land on
                                                           // consecutive 16-B bank slots (conflict-free b128 reads)
#define TW_ACT_BYTES (TW_TB * TW_CELLS * TW_ROW_BYTES)     // 69632
#define TW_CHUNK_BYTES 16384                               // [ks 4][ntile 4][h 2][c 32][j 8] bf16
#define TW_NSLOT 5
#define TW_RING_OFF TW_ACT_BYTES
#define TW_BIAS_OFF (TW_RING_OFF + TW_NSLOT * TW_CHUNK_BYTES)   // 147456
#define TW_MAX_LAYERS 23
#define TW_ZERO_OFF (TW_BIAS_OFF + TW_MAX_LAYERS * TW_CH * 4)   // 159744, 256 B of zeros
#define TW_LDS_BYTES 163840
#define YY_TOWER_TB1_MAX_G 256   // <= one 1-board workgroup per CU
#define YY_TOWER_TB2_MAX_G 512   // <= one 2-board workgroup per CU

extern "C" int yy_tower_set_err(int code, const char *msg);   // defined in yy_engine.hip

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    bf16x2 t;
    t[0] = (__bf16)a;
    t[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, t);
}
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t relu_pk(uint32_t p) {   // max(x, 0) on two packed bf16 as signed int16
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// byte offset of the 16-B slot holding channels [8*chunk, 8*chunk+8) of (board, cell)
__device__ __forceinline__ uint32_t act_off(int board, int cell, int chunk) {
    return (uint32_t)((board * TW_CELLS + cell) * TW_ROW_BYTES + chunk * 16);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// every thread moves 4 x 16 B of a 16 KB chunk: global (fragment order) -> LDS ring slot, no VGPR data
__device__ __forceinline__ void issue_chunk(const unsigned char *wchunk, unsigned char *lds, int slot, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int piece = (r * 4 + wave) * 1024;   // wave-uniform 1 KiB piece
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + piece + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + TW_RING_OFF + slot * TW_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}

// fragments of one k-step (16 input channels of one tap): 2 activation tiles + 4 weight tiles, 6 ds_read_b128
struct Frags {
    bf16x8 x[2], w[4];
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int half, int ks,
                                           const uint32_t (&cbase)[2], int lane) {
    const int h = lane >> 5, c = lane & 31;
    // cbase already contains this lane's h*16; (half, ks) are compile-time after unrolling -> immediate offsets
    const unsigned char *wslot = lds + TW_RING_OFF + slot * TW_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 4096;
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
        f.x[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase[tt] + half * 128 + ks * 32));
#pragma unroll
    for (int nt = 0; nt < 4; nt++) f.w[nt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wslot + nt * 1024));
}
__device__ __forceinline__ void mma8(f32x16 (&acc)[2][4], const Frags &f) {
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[nt], f.x[tt], acc[tt][nt], 0, 0, 0);
}
// first k-step of a layer: C = 0 (inline constant), no accumulator initialisation
__device__ __forceinline__ void mma8_zero(f32x16 (&acc)[2][4], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[nt], f.x[tt], z, 0, 0, 0);
}
// issue order hint: the 6 LDS reads of the NEXT k-step interleaved with the 8 MFMAs of the current one
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);       // 5 MFMAs cover the last reads' latency
}
// Per-lane geometry of the lane's two cells (MFMA columns), computed ONCE per kernel: the LDS offset of the cell's own row
// (+ this lane's h*16), a 9-bit mask of the taps whose neighbour is on the board, and the zero row.  The per-chunk tap
// geometry is then four full-rate VALU ops per cell (and, compare, add a wave-uniform shift, select) instead of the
// compare / multiply chain it used to re-derive inside the MFMA stream of every chunk.
struct LaneGeo {
    uint32_t rowbase[2], okmask[2], zbase;
    int cy[2], cx, wave, h;   // only for the A/B build of the former per-chunk derivation (YY_TOWER_DEBUG=8)
};
__device__ __forceinline__ LaneGeo make_lane_geo(const int (&cy)[2], int cx, int wave, int h) {
    LaneGeo g;
    g.cy[0] = cy[0], g.cy[1] = cy[1], g.cx = cx, g.wave = wave, g.h = h;
    g.zbase = (uint32_t)TW_ZERO_OFF + (uint32_t)(h * 16);
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        g.rowbase[tt] = (uint32_t)((wave * TW_CELLS + cy[tt] * 8 + cx) * TW_ROW_BYTES) + (uint32_t)(h * 16);
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
// LDS row base of the tap's neighbour cell for this lane's two cells (the zero row when off-board); `tap` is wave-uniform
template <int DBG>
__device__ __forceinline__ void tap_geo(int tap, const LaneGeo &g, uint32_t (&cbase)[2]) {
    if constexpr ((DBG & 8) != 0) {   // experiment: re-derive the geometry per chunk, as the kernel did before
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
            const int sy = g.cy[tt] + dy, sx = g.cx + dx;
            const bool ok = ((unsigned)sy < 8u) && ((unsigned)sx < 8u);
            cbase[tt] = (ok ? (uint32_t)((g.wave * TW_CELLS + sy * 8 + sx) * TW_ROW_BYTES) : (uint32_t)TW_ZERO_OFF) + (uint32_t)(g.h * 16);
        }
        return;
    }
    const int shift = ((tap / 3 - 1) * 8 + (tap % 3 - 1)) * TW_ROW_BYTES;   // scalar unit
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) cbase[tt] = (g.okmask[tt] & bit) ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
}

// One layer's chunks.  Invariant on entry and exit of every iteration: chunk `chunk` is visible in its
// ring slot to every wave.  Each iteration first makes chunk+1 visible (counted vmcnt + barrier), refills
// the slot chunk-1 used, then runs KS k-steps whose LDS reads are software-pipelined one k-step ahead,
// across the chunk boundary too (only the first k-step of a layer exposes its read latency).
template <int KS, int DBG>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[2][4], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, const LaneGeo &geo, int wave, int lane) {
    constexpr int NCH = (KS == 1) ? 9 : 18;
    uint32_t cb[2];
    tap_geo<DBG>(0, geo, cb);
    Frags cur;
    load_frags(cur, lds, chunk % TW_NSLOT, 0, 0, cb, lane);
    for (int i = 0; i < NCH; i++, chunk++) {
        const int half = (KS == 1) ? 0 : (i & 1);
        if (chunk + 1 < n_chunks) {
            const int newer = min(2, n_chunks - 2 - chunk);   // chunks younger than chunk+1 still in flight
            if (newer == 2) wait_vmcnt<8>();
            else if (newer == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // chunk+1 landed everywhere; everyone finished chunk-1
            asm volatile("" ::: "memory");
            if (chunk + 4 < n_chunks && !(DBG & 1))
                issue_chunk(weights + (size_t)(chunk + 4) * TW_CHUNK_BYTES, lds, (chunk + 4) % TW_NSLOT, wave, lane);
        }
        if constexpr ((DBG & 2) != 0) continue;   // experiment: weight stream + barriers only
        const bool last = (i == NCH - 1);
        uint32_t ncb[2];
        const int ni = last ? i : i + 1;
        tap_geo<DBG>((KS == 1) ? ni : (ni >> 1), geo, ncb);
        const int nhalf = (KS == 1) ? 0 : (ni & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % TW_NSLOT, half, ks + 1, cb, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % TW_NSLOT, nhalf, 0, ncb, lane);
            if (i == 0 && ks == 0) mma8_zero(acc, cur);
            else mma8(acc, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
        cb[0] = ncb[0];
        cb[1] = ncb[1];
    }
}

template <int DBG>
__global__ void __launch_bounds__(256, 1)
k_tower(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
        unsigned short *__restrict__ out, unsigned short *__restrict__ out_heads, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[TW_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gb = blockIdx.x * TW_TB + wave;              // this wave's board
    const int h = lane >> 5, c = lane & 31;

    // ---- prologue: bias table + zero row + input planes -> LDS (ordinary loads, drained before the ring starts)
    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * TW_CH; i += 256)
        ((float *)(lds + TW_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + TW_ZERO_OFF))[threadIdx.x] = 0u;
    {
        // lane = cell: 5 planes (neural_network.py:156-196) -> channels 0..4 of a 16-channel input, rest zero
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * TW_CELLS + lane] : 0.0f;
        u32x4 v0 = {pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], 0.0f), 0u};
        u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + act_off(wave, lane, 0)) = v0;
        *(u32x4 *)(lds + act_off(wave, lane, 1)) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // the policy/value 1x1 head convolutions ride the same ring as one extra 16 KB chunk after the tower
    const int n_chunks = 9 + 18 * (n_layers - 1) + (out_heads ? 1 : 0);
#pragma unroll
    for (int pc = 0; pc < 4; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * TW_CHUNK_BYTES, lds, pc % TW_NSLOT, wave, lane);

    // chunk 0 visible to everyone before the first layer (3 younger chunks may stay in flight)
    if (n_chunks >= 4) wait_vmcnt<12>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // this lane's two cells (MFMA columns): tile 0 = board rows 0-3, tile 1 = rows 4-7
    const int cy[2] = {c >> 3, 4 + (c >> 3)}, cx = c & 7;
    const LaneGeo geo = make_lane_geo(cy, cx, wave, h);
    uint32_t res[2][4][4][2];   // residual x, packed bf16 in the accumulator layout
    int chunk = 0;

    for (int L = 0; L < n_layers; L++) {
        f32x16 acc[2][4];
        if (L == 0) run_layer<1, DBG>(acc, lds, weights, chunk, n_chunks, geo, wave, lane);
        else run_layer<4, DBG>(acc, lds, weights, chunk, n_chunks, geo, wave, lane);
        if constexpr ((DBG & 4) != 0) {   // experiment: no epilogue (keep the accumulators alive)
            asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]));
            asm volatile("" ::"v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]));
            continue;
        }
        // ---- epilogue (wave-private: a wave reads and writes only its own board's cells):
        // + bias (+ residual), bf16 rounding, ReLU on the packed pair (v_pk_max_i16: bf16 is sign-magnitude).
        // No MFMA runs here, so every stall is paid in full: the 16 bias vectors of this lane are fetched in two batches of
        // 8 back-to-back reads (one LDS latency each batch instead of one per vector), the pair conversion is ONE
        // v_cvt_pk_bf16_f32 (vector convert), and the residual branch is resolved once per layer.
        const bool conv2 = (L >= 2) && ((L & 1) == 0);   // second conv of a block: + residual
        const bool keep = (L == 0) || conv2;             // output is a block input x: keep it for the skip
        const unsigned char *bl = lds + TW_BIAS_OFF + (L * TW_CH + 4 * h) * 4;
        auto epilogue_half = [&](const int nt0, auto with_res) {
            constexpr bool RES = decltype(with_res)::value;
            f32x4 b[2][4];
#pragma unroll
            for (int n2 = 0; n2 < 2; n2++)
#pragma unroll
                for (int q = 0; q < 4; q++) b[n2][q] = *(const f32x4 *)(bl + ((nt0 + n2) * 32 + 8 * q) * 4);
#pragma unroll
            for (int n2 = 0; n2 < 2; n2++) {
                const int nt = nt0 + n2;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        // this lane's rows of tile nt are couts nt*32 + 8q + 4h + i
                        f32x2 v01 = {acc[tt][nt][4 * q + 0] + b[n2][q][0], acc[tt][nt][4 * q + 1] + b[n2][q][1]};
                        f32x2 v23 = {acc[tt][nt][4 * q + 2] + b[n2][q][2], acc[tt][nt][4 * q + 3] + b[n2][q][3]};
                        if (RES) {
                            const uint32_t r0 = res[tt][nt][q][0], r1 = res[tt][nt][q][1];
                            v01 += (f32x2){bf_lo(r0), bf_hi(r0)};
                            v23 += (f32x2){bf_lo(r1), bf_hi(r1)};
                        }
                        const uint32_t p0 = relu_pk(__builtin_bit_cast(uint32_t, __builtin_convertvector(v01, bf16x2)));
                        const uint32_t p1 = relu_pk(__builtin_bit_cast(uint32_t, __builtin_convertvector(v23, bf16x2)));
                        if (keep) {
                            res[tt][nt][q][0] = p0;
                            res[tt][nt][q][1] = p1;
                        }
                        u32x2 pk = {p0, p1};
                        *(u32x2 *)(lds + act_off(wave, tt * 32 + c, nt * 4 + q) + h * 8) = pk;
                    }
            }
        };
        if (conv2) {
            epilogue_half(0, std::true_type{});
            epilogue_half(2, std::true_type{});
        } else {
            epilogue_half(0, std::false_type{});
            epilogue_half(2, std::false_type{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (out_heads) {
        // ---- policy_conv / value_conv (1x1, 128 -> 32 each; neural_network.py:59-60, 65-66, 113, 118) + bias + ReLU
        // chunk layout [ks 8][nt 2][h 2][c 32][j 8]: nt 0 = policy channels, nt 1 = value channels
        const unsigned char *hw = lds + TW_RING_OFF + (chunk % TW_NSLOT) * TW_CHUNK_BYTES + (h * 32 + c) * 16;
        const uint32_t xb[2] = {(uint32_t)((wave * TW_CELLS + c) * TW_ROW_BYTES + h * 16),
                                (uint32_t)((wave * TW_CELLS + 32 + c) * TW_ROW_BYTES + h * 16)};
        f32x16 hacc[2][2];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            bf16x8 xf[2], wf[2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) xf[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + xb[tt] + ks * 32));
#pragma unroll
            for (int nt = 0; nt < 2; nt++) wf[nt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(hw + (ks * 2 + nt) * 1024));
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    if (ks == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        hacc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[tt], z, 0, 0, 0);
                    } else {
                        hacc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[tt], hacc[tt][nt], 0, 0, 0);
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // stage [head][channel 32][cell 64] bf16 (the reference's NCHW flatten order, :114/:119) in this wave's
        // own (now dead) activation rows, then stream it out 1 KiB per wave store
        unsigned char *stg = lds + wave * TW_CELLS * TW_ROW_BYTES;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b = *(const f32x4 *)(lds + TW_BIAS_OFF + (n_layers * TW_CH + nt * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = fmaxf(hacc[tt][nt][4 * q + i] + b[i], 0.0f);
                        const int ch = 8 * q + 4 * h + i, cell = tt * 32 + c;
                        *(unsigned short *)(stg + ((nt * 32 + ch) * TW_CELLS + cell) * 2) = (unsigned short)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
            }
        __syncthreads();
        if (gb < G) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const u32x4 v = *(const u32x4 *)(stg + (i * 64 + lane) * 16);
                *(u32x4 *)(out_heads + (size_t)gb * 4096 + (i * 64 + lane) * 8) = v;
            }
        }
        return;
    }
    // ---- final activations -> HBM, [board][cell][128] bf16 (channels-last), 1 KiB per wave store
    __syncthreads();
    if (gb < G) {
        const int ch = lane & 15;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int cell = (lane >> 4) + 4 * i;
            const u32x4 v = *(const u32x4 *)(lds + act_off(wave, cell, ch));
            *(u32x4 *)(out + ((size_t)gb * TW_CELLS + cell) * TW_CH + ch * 8) = v;
        }
    }
}

// weights: bf16 chunks [9 + 18*(n_layers-1) (+1 head chunk)][8192] in fragment order (network.pack_tower);
// bias f32 [n_layers (+1)][128]; planes f32 [G,5,8,8]; out bf16 [G,8,8,128] or out_heads bf16 [G,2,32,64].
extern "C" int yy_tower6_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                int G, int n_layers, yy_stream_t s);   // yy_towerq.hip: 6x6 boards

extern "C" int yy_tower12q_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                  int G, int n_layers, yy_stream_t s);   // yy_towerq.hip: 12x12, wave = cout quarter

extern "C" int yy_tower8q_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                 int G, int n_layers, int tb, yy_stream_t s);   // yy_towerq.hip: 8x8, 1 or 2 boards per workgroup

static int launch_tower(const float *planes, const void *weights, const float *bias, void *out, void *out_heads, int G,
                        int R, int C, int channels, int n_layers, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || (!out && !out_heads) || G < 0)
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower: bad argument");
    const bool b8 = (R == 8 && C == 8), b12 = (R == 12 && C == 12), b6 = (R == 6 && C == 6);
    if (!(b8 || b12 || b6) || channels != TW_CH || n_layers < 1 || n_layers + (out_heads ? 1 : 0) > TW_MAX_LAYERS || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower: needs 6x6, 8x8 or 12x12 boards, 128 channels, at most 10 residual blocks");
    if (b12) return yy_tower12q_launch(planes, weights, bias, out, out_heads, G, n_layers, s);
    if (b6) return yy_tower6_launch(planes, weights, bias, out, out_heads, G, n_layers, s);
    // small batches: 4 boards per workgroup would leave most of the 256 CUs idle; spread the boards over more, lighter
    // workgroups (same results bit for bit)
    int tb = G <= YY_TOWER_TB1_MAX_G ? 1 : (G <= YY_TOWER_TB2_MAX_G ? 2 : 4);
    int dbg = 0;
#ifdef YY_TOWER_EXPERIMENTS   // timing experiments only (profiles/r01_k_tower_pmc.txt): never compiled into the shipped library
    static const int env_dbg = getenv("YY_TOWER_DEBUG") ? atoi(getenv("YY_TOWER_DEBUG")) : 0;
    static const int force_tb = getenv("YY_TOWER_TB") ? atoi(getenv("YY_TOWER_TB")) : 0;
    dbg = env_dbg;
    if (force_tb) tb = force_tb;
#endif
    if (tb == 1 || tb == 2) return yy_tower8q_launch(planes, weights, bias, out, out_heads, G, n_layers, tb, s);
    const dim3 grid((G + TW_TB - 1) / TW_TB), block(256);
    const unsigned char *w = (const unsigned char *)weights;
    unsigned short *o = (unsigned short *)out, *oh = (unsigned short *)out_heads;
    hipStream_t st = (hipStream_t)s;
    switch (dbg) {
#ifdef YY_TOWER_EXPERIMENTS
        case 1: k_tower<1><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 2: k_tower<2><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 4: k_tower<4><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 5: k_tower<5><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 6: k_tower<6><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 7: k_tower<7><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
        case 8: k_tower<8><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
#endif
        default: k_tower<0><<<grid, block, 0, st>>>(planes, w, bias, o, oh, G, n_layers); break;
    }
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower: launch failed");
    return YY_OK;
}

extern "C" int yy_nn_tower_bf16(const float *planes, const void *weights, const float *bias, void *out, int G, int R,
                                int C, int channels, int n_layers, yy_stream_t s) {
    return launch_tower(planes, weights, bias, out, nullptr, G, R, C, channels, n_layers, s);
}

extern "C" int yy_nn_tower_heads_bf16(const float *planes, const void *weights, const float *bias, void *out_heads, int G,
                                      int R, int C, int channels, int n_layers, yy_stream_t s) {
    return launch_tower(planes, weights, bias, nullptr, out_heads, G, R, C, channels, n_layers, s);
}

// yy_tower_x3.hip -- the LDS-resident tower kernel (design: yy_tower.hip) at float32-grade accuracy on the BF16 matrix
// cores: every activation x and every weight w is held as a pair of bf16 numbers (hi = bf16(x), lo = bf16(x - hi), together
// 16 mantissa bits) and each product is formed as  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  with three
// v_mfma_f32_32x32x16_bf16 into ONE f32 accumulator (the dropped w_lo*x_lo term is 2^-18 relative).  Bias, residual
// (kept as exact f32 in registers) and ReLU are applied in f32 and the result is split again for the next layer.
// Against the fp32 module the tower output agrees to ~1e-5 relative, i.e. it serves the fp32 parity evaluator, at
// 3/16 of the cost of the f32-input MFMA (bf16 MFMA is 16x the f32 rate on gfx950): `--nn bf16x3`.
//
// 8x8 boards, 128 channels.  A (board, part) pair is laid out exactly like a board of yy_tower.hip (64 cells x 272 B),
// so a workgroup holds TWO boards = 4 "virtual boards" (hi/lo of each); wave w owns board (w & 1) and output-channel half
// (w >> 1): a 64-cout x 64-cell tile = 2 x 2 accumulators.  A weight chunk is one tap x 32 input channels x 128 couts x
// {hi, lo} = 16 KB in fragment order [ks 2][part 2][nt 4][h 2][c 32][j 8]; per k-step a wave reads 4 activation + 4 weight
// fragments for 12 MFMAs.  5-slot LDS-DMA ring, one barrier per chunk, two extra barriers per layer around the epilogue.
// Roofline: bf16 MFMA; algorithmic FLOPs are the conv's (one product per term), issued MFMA FLOPs are 3x that.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define X_TB 2
#define X_CH 128
#define X_CELLS 64
#define X_ROW_BYTES 272                                      // 128 bf16 channels + 16 B pad
#define X_ACT_BYTES (X_TB * 2 * X_CELLS * X_ROW_BYTES)       // 69632: [board 2][part 2][cell 64] rows
#define X_CHUNK_BYTES 16384
#define X_NSLOT 5
#define X_RING_OFF X_ACT_BYTES
#define X_BIAS_OFF (X_RING_OFF + X_NSLOT * X_CHUNK_BYTES)
#define X_MAX_LAYERS 23
#define X_ZERO_OFF (X_BIAS_OFF + X_MAX_LAYERS * X_CH * 4)
#define X_LDS_BYTES (X_ZERO_OFF + 256)                        // 163584

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace tx3 {

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
                                         (__attribute__((address_space(3))) void *)(lds + X_RING_OFF + slot * X_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
// byte offset of the row of (board, part, cell)
__device__ __forceinline__ uint32_t row_off(int board, int part, int cell) {
    return (uint32_t)(((board * 2 + part) * X_CELLS + cell) * X_ROW_BYTES);
}
__device__ __forceinline__ float bf_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
// x -> (hi, lo) bf16 bit patterns, round-to-nearest-even both times
__device__ __forceinline__ void split_bf16(float x, uint32_t &hi, uint32_t &lo) {
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;                 // exact in f32
    const __bf16 l = (__bf16)r;
    hi = (uint32_t)__builtin_bit_cast(unsigned short, h);
    lo = (uint32_t)__builtin_bit_cast(unsigned short, l);
}

// one k-step = 16 input channels of one tap: hi/lo activation fragments of both column tiles, hi/lo weight fragments of
// this wave's two cout tiles
struct Frags {
    bf16x8 xh[2], xl[2], wh[2], wl[2];
};
struct Bases {
    uint32_t hi[2], lo[2];   // byte offsets (+ h*16) of the hi and lo rows of this lane's tap neighbour, or the zero row
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int quarter, int ks,
                                           const Bases &cbase, int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    // chunk: [ks 2][part 2][nt 4][h 2][c 32][j 8]; this wave's tiles are nt = 2*nh, 2*nh + 1
    const unsigned char *wslot = lds + X_RING_OFF + slot * X_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 8192 + nh * 2048;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        f.xh[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase.hi[tt] + quarter * 64 + ks * 32));
        f.xl[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase.lo[tt] + quarter * 64 + ks * 32));
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        f.wh[nt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wslot + nt * 1024));
        f.wl[nt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wslot + 4096 + nt * 1024));
    }
}
template <bool ZERO> __device__ __forceinline__ void mma12(f32x16 (&acc)[2][2], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            // small terms first, the dominant hi*hi last
            f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl[nt], f.xh[tt], ZERO ? z : acc[tt][nt], 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[nt], f.xl[tt], a, 0, 0, 0);
            acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[nt], f.xh[tt], a, 0, 0, 0);
        }
}
// per-lane geometry, computed once per kernel: own rows (hi part; the lo part sits 64 rows further), on-board tap mask
struct LaneGeo {
    uint32_t rowbase[2], okmask[2], zbase;
};
__device__ __forceinline__ LaneGeo make_lane_geo(const int (&cy)[2], int cx, int board, int h) {
    LaneGeo g;
    g.zbase = (uint32_t)X_ZERO_OFF + (uint32_t)(h * 16);
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
    const int shift = ((tap / 3 - 1) * 8 + (tap % 3 - 1)) * X_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        const bool ok = (g.okmask[tt] & bit) != 0;
        cbase.hi[tt] = ok ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
        cbase.lo[tt] = ok ? g.rowbase[tt] + (uint32_t)(shift + X_CELLS * X_ROW_BYTES) : g.zbase;
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
__device__ __forceinline__ void run_layer(f32x16 (&acc)[2][2], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, const LaneGeo &geo, int nh, int wave, int lane) {
    constexpr int Q = STEM ? 1 : 4, KS = STEM ? 1 : 2, NCH = 9 * Q;
    Bases cb;
    tap_geo(0, geo, cb);
    Frags cur;
    for (int i = 0; i < NCH; i++, chunk++) {
        const int quarter = STEM ? 0 : (i & 3);
        if (chunk + 1 < n_chunks) {
            const int newer = min(2, n_chunks - 2 - chunk);
            if (newer == 2) wait_vmcnt<8>();
            else if (newer == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
        }
        if (chunk + 1 < n_chunks || i == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // chunk+1 landed everywhere; everyone finished chunk-1 (and the epilogue)
            asm volatile("" ::: "memory");
        }
        if (chunk + 1 < n_chunks && chunk + 4 < n_chunks)
            issue_chunk(weights + (size_t)(chunk + 4) * X_CHUNK_BYTES, lds, (chunk + 4) % X_NSLOT, wave, lane);
        if (i == 0) load_frags(cur, lds, chunk % X_NSLOT, 0, 0, cb, nh, lane);
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
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % X_NSLOT, quarter, ks + 1, cb, nh, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % X_NSLOT, nquarter, 0, next_tap ? ncb : cb, nh, lane);
            if (i == 0 && ks == 0) mma12<true>(acc, cur);
            else mma12<false>(acc, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
        if (next_tap) cb = ncb;
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower_x3(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
           float *__restrict__ out, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[X_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int board = wave & 1, nh = wave >> 1;
    const int gb = blockIdx.x * X_TB + board;
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < n_layers * X_CH; i += 256) ((float *)(lds + X_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + X_ZERO_OFF))[threadIdx.x] = 0u;
    {   // wave (board, nh): part nh of the input; lane = cell: 5 planes -> channels 0..4 of a 16-channel zero-padded input
        uint32_t hi[5], lo[5];
#pragma unroll
        for (int k = 0; k < 5; k++) split_bf16((gb < G) ? planes[((size_t)gb * 5 + k) * X_CELLS + lane] : 0.0f, hi[k], lo[k]);
        const uint32_t *s = nh ? lo : hi;
        const u32x4 v0 = {s[0] | (s[1] << 16), s[2] | (s[3] << 16), s[4], 0u};
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + row_off(board, nh, lane)) = v0;
        *(u32x4 *)(lds + row_off(board, nh, lane) + 16) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 36 * (n_layers - 1);
#pragma unroll
    for (int pc = 0; pc < 4; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * X_CHUNK_BYTES, lds, pc % X_NSLOT, wave, lane);
    if (n_chunks >= 4) wait_vmcnt<12>();
    else wait_vmcnt<0>();

    const int cy[2] = {c >> 3, 4 + (c >> 3)}, cx = c & 7;
    const LaneGeo geo = make_lane_geo(cy, cx, board, h);
    f32x4 res[2][2][4];   // residual x of this wave's 64 couts, f32
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc[2][2];
        if (L == 0) run_layer<true>(acc, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        else run_layer<false>(acc, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
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
                bq[nt][q] = *(const f32x4 *)(lds + X_BIAS_OFF + (L * X_CH + (nh * 2 + nt) * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = (nh * 2 + nt) * 32 + 8 * q + 4 * h;   // this lane's 4 couts (accumulator rows 4q..4q+3)
                const f32x4 b = bq[nt][q];
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    f32x4 v = {acc[tt][nt][4 * q + 0] + b[0], acc[tt][nt][4 * q + 1] + b[1],
                               acc[tt][nt][4 * q + 2] + b[2], acc[tt][nt][4 * q + 3] + b[3]};
                    if (conv2) v += res[tt][nt][q];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.0f);
                    if (keep) res[tt][nt][q] = v;
                    // (hi, lo) split, two values per packed convert: hi = bf16(v), lo = bf16(v - hi)
                    const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
                    const bf16x2 h01 = __builtin_convertvector(a01, bf16x2), h23 = __builtin_convertvector(a23, bf16x2);
                    const f32x2 r01 = a01 - __builtin_convertvector(h01, f32x2), r23 = a23 - __builtin_convertvector(h23, f32x2);
                    const bf16x2 l01 = __builtin_convertvector(r01, bf16x2), l23 = __builtin_convertvector(r23, bf16x2);
                    const int cell = tt * 32 + c;
                    const u32x2 ph = {__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23)};
                    const u32x2 pl = {__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23)};
                    *(u32x2 *)(lds + row_off(board, 0, cell) + co * 2) = ph;
                    *(u32x2 *)(lds + row_off(board, 1, cell) + co * 2) = pl;
                }
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (gb < G) {   // activations [cell][128] f32 (channels-last) = hi + lo; 64 x 32 pieces of 4 channels per board, half per wave
        for (int p = nh * 1024 + lane; p < (nh + 1) * 1024; p += 64) {
            const int cell = p >> 5, ch4 = p & 31;
            const u32x2 ph = *(const u32x2 *)(lds + row_off(board, 0, cell) + ch4 * 8);
            const u32x2 pl = *(const u32x2 *)(lds + row_off(board, 1, cell) + ch4 * 8);
            const f32x4 v = {bf_to_f32(ph[0] & 0xFFFFu) + bf_to_f32(pl[0] & 0xFFFFu), bf_to_f32(ph[0] >> 16) + bf_to_f32(pl[0] >> 16),
                             bf_to_f32(ph[1] & 0xFFFFu) + bf_to_f32(pl[1] & 0xFFFFu), bf_to_f32(ph[1] >> 16) + bf_to_f32(pl[1] >> 16)};
            *(f32x4 *)(out + ((size_t)gb * X_CELLS + cell) * X_CH + ch4 * 4) = v;
        }
    }
}

}   // namespace tx3

// weights: bf16 chunks [9 + 36*(n_layers-1)][8192] in fragment order (network.pack_tower_x3); bias f32 [n_layers,128];
// planes f32 [G,5,8,8]; out f32 [G,8,8,128].
extern "C" int yy_nn_tower_bf16x3(const float *planes, const void *weights, const float *bias, float *out, int G, int R, int C,
                                  int channels, int n_layers, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || !out || G < 0) return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_bf16x3: bad argument");
    if (R != 8 || C != 8 || channels != X_CH || n_layers < 1 || n_layers > X_MAX_LAYERS || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_bf16x3: needs 8x8 boards, 128 channels, at most 11 residual blocks");
    tx3::k_tower_x3<<<dim3((G + X_TB - 1) / X_TB), dim3(256), 0, (hipStream_t)s>>>(planes, (const unsigned char *)weights, bias,
                                                                                out, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_bf16x3: launch failed");
    return YY_OK;
}

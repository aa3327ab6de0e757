// yy_tower_f32.hip -- the LDS-resident tower kernel (design: yy_tower.hip) in EXACT float32 on the f32-input MFMA
// v_mfma_f32_32x32x2_f32 (gfx950 has no TF32-like shortcut; this instruction is a k-ordered fmaf chain, bitwise f32).
// It is the fast form of the fp32 parity evaluator: same folded weights as the fp32 nn.Module, results differ from the
// library convolutions only by summation order.
//
// 8x8 boards, 128 channels.  f32 activations take 512 B per cell, so a workgroup holds TWO boards (66 KB with 16-B row
// padding); wave w owns board (w & 1) and output-channel half (w >> 1): a 64-cout x 64-cell tile = 2 x 2 accumulators,
// the residual stays in 64 f32 registers.  A weight chunk is one tap x 32 input channels x 128 couts = 16 KB in
// fragment order [m 4][nt 4][h 2][i 32][s 4]: one ds_read_b128 per operand feeds FOUR MFMA k-steps (lane half h and
// step s select input channel 8m + 4h + s for both operands).  5-slot LDS-DMA ring, one barrier per chunk (64 MFMAs
// of 64 cycles each per wave), two extra barriers per layer around the epilogue (both halves read all channels).
// Roofline: f32 MFMA, 157 TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define F_TB 2
#define F_CH 128
#define F_CELLS 64
#define F_ROW_BYTES 528                                      // 512 B of channels + 16 B pad (conflict-free b128 reads)
#define F_ACT_BYTES (F_TB * F_CELLS * F_ROW_BYTES)            // 67584
#define F_CHUNK_BYTES 16384
#define F_NSLOT 5
#define F_RING_OFF F_ACT_BYTES
#define F_BIAS_OFF (F_RING_OFF + F_NSLOT * F_CHUNK_BYTES)
#define F_MAX_LAYERS 23
#define F_ZERO_OFF (F_BIAS_OFF + F_MAX_LAYERS * F_CH * 4)
#define F_LDS_BYTES (F_ZERO_OFF + 512)                        // 161792

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace tf32 {

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
                                         (__attribute__((address_space(3))) void *)(lds + F_RING_OFF + slot * F_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
__device__ __forceinline__ uint32_t act_off(int board, int cell, int ch) {   // byte offset of channel ch of (board, cell)
    return (uint32_t)((board * F_CELLS + cell) * F_ROW_BYTES + ch * 4);
}
// one m-block = 8 input channels: 2 activation + 2 weight 16-B fragments feed 4 k-steps x (2 x 2) tiles = 16 MFMAs
struct Frags {
    f32x4 x[2], w[2];
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int quarter, int m,
                                           const uint32_t (&cbase)[2], int nh, int lane) {
    // cbase holds this lane's h*16; channels of this m-block start at quarter*32 + m*8
    const unsigned char *wslot = lds + F_RING_OFF + slot * F_CHUNK_BYTES + lane * 16 + m * 4096 + nh * 2048;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) f.x[tt] = *(const f32x4 *)(lds + cbase[tt] + quarter * 128 + m * 32);
#pragma unroll
    for (int nt = 0; nt < 2; nt++) f.w[nt] = *(const f32x4 *)(wslot + nt * 1024);
}
template <bool ZERO> __device__ __forceinline__ void mma16(f32x16 (&acc)[2][2], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
                acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[nt][s], f.x[tt][s], (ZERO && s == 0) ? z : acc[tt][nt], 0, 0, 0);
}
// per-lane geometry, computed once per kernel: own row offsets (+ h*16), on-board tap mask, zero row
struct LaneGeo {
    uint32_t rowbase[2], okmask[2], zbase;
};
__device__ __forceinline__ LaneGeo make_lane_geo(const int (&cy)[2], int cx, int board, int h) {
    LaneGeo g;
    g.zbase = (uint32_t)F_ZERO_OFF + (uint32_t)(h * 16);
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        g.rowbase[tt] = (uint32_t)((board * F_CELLS + cy[tt] * 8 + cx) * F_ROW_BYTES) + (uint32_t)(h * 16);
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
__device__ __forceinline__ void tap_geo(int tap, const LaneGeo &g, uint32_t (&cbase)[2]) {
    const int shift = ((tap / 3 - 1) * 8 + (tap % 3 - 1)) * F_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) cbase[tt] = (g.okmask[tt] & bit) ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
}

// A layer = 9 taps x Q quarter-chunks (Q = 4; stem: 1 chunk per tap, only its first m-block is non-zero / computed).
template <bool STEM>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[2][2], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, const LaneGeo &geo, int nh, int wave, int lane) {
    constexpr int Q = STEM ? 1 : 4, MB = STEM ? 1 : 4, NCH = 9 * Q;
    uint32_t cb[2];
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
            issue_chunk(weights + (size_t)(chunk + 4) * F_CHUNK_BYTES, lds, (chunk + 4) % F_NSLOT, wave, lane);
        if (i == 0) load_frags(cur, lds, chunk % F_NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        const int ni = last ? i : i + 1;
        uint32_t ncb[2];
        tap_geo(STEM ? ni : (ni >> 2), geo, ncb);
        const int nquarter = STEM ? 0 : (ni & 3);
        const bool next_tap = STEM || (quarter == 3);
#pragma unroll
        for (int m = 0; m < MB; m++) {
            Frags nxt;
            const bool has_next = (m + 1 < MB) || !last;
            if (m + 1 < MB) load_frags(nxt, lds, chunk % F_NSLOT, quarter, m + 1, cb, nh, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % F_NSLOT, nquarter, 0, next_tap ? ncb : cb, nh, lane);
            if (i == 0 && m == 0) mma16<true>(acc, cur);
            else mma16<false>(acc, cur);
            if (has_next) cur = nxt;
        }
        if (next_tap) {
            cb[0] = ncb[0];
            cb[1] = ncb[1];
        }
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower_f32(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
            float *__restrict__ out, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[F_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int board = wave & 1, nh = wave >> 1;
    const int gb = blockIdx.x * F_TB + board;
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < n_layers * F_CH; i += 256) ((float *)(lds + F_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 128) ((uint32_t *)(lds + F_ZERO_OFF))[threadIdx.x] = 0u;
    if (nh == 0) {   // lane = cell: 5 planes -> channels 0..4 of an 8-channel zero-padded input
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * F_CELLS + lane] : 0.0f;
        *(f32x4 *)(lds + act_off(board, lane, 0)) = (f32x4){p[0], p[1], p[2], p[3]};
        *(f32x4 *)(lds + act_off(board, lane, 4)) = (f32x4){p[4], 0.0f, 0.0f, 0.0f};
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 36 * (n_layers - 1);
#pragma unroll
    for (int pc = 0; pc < 4; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * F_CHUNK_BYTES, lds, pc % F_NSLOT, wave, lane);
    if (n_chunks >= 4) wait_vmcnt<12>();
    else wait_vmcnt<0>();

    const int cy[2] = {c >> 3, 4 + (c >> 3)}, cx = c & 7;
    const LaneGeo geo = make_lane_geo(cy, cx, board, h);
    f32x4 res[2][2][4];   // residual x of this wave's 64 couts, exact f32
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
                bq[nt][q] = *(const f32x4 *)(lds + F_BIAS_OFF + (L * F_CH + (nh * 2 + nt) * 32 + 8 * q + 4 * h) * 4);
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
                    *(f32x4 *)(lds + act_off(board, tt * 32 + c, co)) = v;
                }
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (gb < G) {   // activations [cell][128] f32 (channels-last): 64 x 32 pieces of 16 B per board, half per wave
        for (int p = nh * 1024 + lane; p < (nh + 1) * 1024; p += 64) {
            const int cell = p >> 5, ch4 = p & 31;
            *(f32x4 *)(out + ((size_t)gb * F_CELLS + cell) * F_CH + ch4 * 4) = *(const f32x4 *)(lds + act_off(board, cell, ch4 * 4));
        }
    }
}

}   // namespace tf32

// weights: f32 chunks [9 + 36*(n_layers-1)][4096] in fragment order (network.pack_tower_f32); bias f32 [n_layers,128];
// planes f32 [G,5,8,8]; out f32 [G,8,8,128].
extern "C" int yy_nn_tower_f32(const float *planes, const void *weights, const float *bias, float *out, int G, int R, int C,
                               int channels, int n_layers, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || !out || G < 0) return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_f32: bad argument");
    if (R != 8 || C != 8 || channels != F_CH || n_layers < 1 || n_layers > F_MAX_LAYERS || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_f32: needs 8x8 boards, 128 channels, at most 11 residual blocks");
    tf32::k_tower_f32<<<dim3((G + F_TB - 1) / F_TB), dim3(256), 0, (hipStream_t)s>>>(planes, (const unsigned char *)weights,
                                                                                  bias, out, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_f32: launch failed");
    return YY_OK;
}

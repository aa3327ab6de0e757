// yy_tower16.hip -- the LDS-resident tower kernel of yy_tower.hip on the v_mfma_f32_16x16x32_bf16 shape.
//
// Same design (4 boards per workgroup, activations LDS-resident in 272-B padded rows, 5-slot LDS-DMA weight
// ring, bias/residual/ReLU epilogue in registers); only the fragment geometry differs:
//   A (weights)      lane l: row r = l & 15 (cout), k = 8*(l >> 4) + j        -> chunk [ks 2][rt 8][g 4][r 16][j 8]
//   B (activations)  lane l: col c = l & 15 (cell), k = 8*(l >> 4) + j
//   D                lane l: col c = l & 15 (cell), rows 4*(l >> 4) + reg     -> 4 consecutive couts per lane
// A wave's 128 x 64 tile = 8 x 4 accumulators of 4 registers.  MI355X_MICROARCH.md (DVFS give-back, item 7)
// measures this shape holding a higher clock than 32x32x16 in LDS-fed bf16 loops at equal cycles per FLOP.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define TW_TB 4
#define TW_CH 128
#define TW_CELLS 64
#define TW_ROW_BYTES 272
#define TW_ACT_BYTES (TW_TB * TW_CELLS * TW_ROW_BYTES)
#define TW_CHUNK_BYTES 16384
#define TW_NSLOT 5
#define TW_RING_OFF TW_ACT_BYTES
#define TW_BIAS_OFF (TW_RING_OFF + TW_NSLOT * TW_CHUNK_BYTES)
#define TW_MAX_LAYERS 23
#define TW_ZERO_OFF (TW_BIAS_OFF + TW_MAX_LAYERS * TW_CH * 4)
#define TW_LDS_BYTES 163840

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace t16 {

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    bf16x2 t;
    t[0] = (__bf16)a;
    t[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ uint32_t relu_pk(uint32_t p) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t act_off(int board, int cell, int chunk) {
    return (uint32_t)((board * TW_CELLS + cell) * TW_ROW_BYTES + chunk * 16);
}
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
                                         (__attribute__((address_space(3))) void *)(lds + TW_RING_OFF + slot * TW_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}

// one k-step = 32 input channels of one tap: 4 activation tiles + 8 weight tiles = 12 ds_read_b128, 32 MFMAs
struct Frags {
    bf16x8 x[4], w[8];
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int half, int ks,
                                           const uint32_t (&cbase)[4], int lane) {
    const unsigned char *wslot = lds + TW_RING_OFF + slot * TW_CHUNK_BYTES + lane * 16 + ks * 8192;
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
        f.x[ct] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase[ct] + half * 128 + ks * 64));
#pragma unroll
    for (int rt = 0; rt < 8; rt++) f.w[rt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wslot + rt * 1024));
}
template <bool ZERO>
__device__ __forceinline__ void mma32(f32x4 (&acc)[4][8], const Frags &f) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int rt = 0; rt < 8; rt++)
            acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[rt], f.x[ct], ZERO ? z : acc[ct][rt], 0, 0, 0);
}
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int j = 0; j < 6; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
}
// lane's 4 cells: tile ct covers board rows 2ct, 2ct+1
__device__ __forceinline__ void tap_geo(int tap, int c, int wave, int g, uint32_t (&cbase)[4]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        const int sy = ct * 2 + (c >> 3) + dy, sx = (c & 7) + dx;
        const bool ok = ((unsigned)sy < 8u) && ((unsigned)sx < 8u);
        cbase[ct] = (ok ? (uint32_t)((wave * TW_CELLS + sy * 8 + sx) * TW_ROW_BYTES) : (uint32_t)TW_ZERO_OFF) + (uint32_t)(g * 16);
    }
}

template <int KS>
__device__ __forceinline__ void run_layer(f32x4 (&acc)[4][8], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, int c, int wave, int lane) {
    constexpr int NCH = (KS == 1) ? 9 : 18;
    const int g = lane >> 4;
    uint32_t cb[4];
    tap_geo(0, c, wave, g, cb);
    Frags cur;
    load_frags(cur, lds, chunk % TW_NSLOT, 0, 0, cb, lane);
    for (int i = 0; i < NCH; i++, chunk++) {
        const int half = (KS == 1) ? 0 : (i & 1);
        if (chunk + 1 < n_chunks) {
            const int newer = min(2, n_chunks - 2 - chunk);
            if (newer == 2) wait_vmcnt<8>();
            else if (newer == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (chunk + 4 < n_chunks)
                issue_chunk(weights + (size_t)(chunk + 4) * TW_CHUNK_BYTES, lds, (chunk + 4) % TW_NSLOT, wave, lane);
        }
        const bool last = (i == NCH - 1);
        uint32_t ncb[4];
        const int ni = last ? i : i + 1;
        tap_geo((KS == 1) ? ni : (ni >> 1), c, wave, g, ncb);
        const int nhalf = (KS == 1) ? 0 : (ni & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % TW_NSLOT, half, ks + 1, cb, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % TW_NSLOT, nhalf, 0, ncb, lane);
            if (i == 0 && ks == 0) mma32<true>(acc, cur);
            else mma32<false>(acc, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ct++) cb[ct] = ncb[ct];
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower16(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
          unsigned short *__restrict__ out, unsigned short *__restrict__ out_heads, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[TW_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gb = blockIdx.x * TW_TB + wave;
    const int g = lane >> 4, c = lane & 15;

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * TW_CH; i += 256)
        ((float *)(lds + TW_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + TW_ZERO_OFF))[threadIdx.x] = 0u;
    {
        // lane = cell: 5 planes -> channels 0..4 of a 32-channel (one k-step) zero-padded input
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * TW_CELLS + lane] : 0.0f;
        u32x4 v0 = {pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], 0.0f), 0u};
        u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + act_off(wave, lane, 0)) = v0;
        *(u32x4 *)(lds + act_off(wave, lane, 1)) = z;
        *(u32x4 *)(lds + act_off(wave, lane, 2)) = z;
        *(u32x4 *)(lds + act_off(wave, lane, 3)) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 18 * (n_layers - 1) + (out_heads ? 1 : 0);
#pragma unroll
    for (int pc = 0; pc < 4; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * TW_CHUNK_BYTES, lds, pc % TW_NSLOT, wave, lane);
    if (n_chunks >= 4) wait_vmcnt<12>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    uint32_t res[4][8][2];   // residual x, packed bf16, accumulator layout
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x4 acc[4][8];
        if (L == 0) run_layer<1>(acc, lds, weights, chunk, n_chunks, c, wave, lane);
        else run_layer<2>(acc, lds, weights, chunk, n_chunks, c, wave, lane);
        const bool conv2 = (L >= 2) && ((L & 1) == 0);
        const bool keep = (L == 0) || conv2;
#pragma unroll
        for (int rt = 0; rt < 8; rt++) {
            // this lane's rows of tile rt are couts rt*16 + 4g + i
            const f32x4 b = *(const f32x4 *)(lds + TW_BIAS_OFF + (L * TW_CH + rt * 16 + 4 * g) * 4);
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                f32x2 v01 = {acc[ct][rt][0] + b[0], acc[ct][rt][1] + b[1]};
                f32x2 v23 = {acc[ct][rt][2] + b[2], acc[ct][rt][3] + b[3]};
                if (conv2) {
                    const uint32_t r0 = res[ct][rt][0], r1 = res[ct][rt][1];
                    v01 += (f32x2){bf_lo(r0), bf_hi(r0)};
                    v23 += (f32x2){bf_lo(r1), bf_hi(r1)};
                }
                const uint32_t p0 = relu_pk(pack_bf16(v01[0], v01[1]));
                const uint32_t p1 = relu_pk(pack_bf16(v23[0], v23[1]));
                if (keep) {
                    res[ct][rt][0] = p0;
                    res[ct][rt][1] = p1;
                }
                u32x2 pk = {p0, p1};
                *(u32x2 *)(lds + act_off(wave, ct * 16 + c, rt * 2 + (g >> 1)) + (g & 1) * 8) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (out_heads) {
        // 1x1 head convs: chunk layout [ks 4][rt 4][g 4][r 16][j 8]; couts 0..31 policy, 32..63 value
        const unsigned char *hw = lds + TW_RING_OFF + (chunk % TW_NSLOT) * TW_CHUNK_BYTES + lane * 16;
        uint32_t xb[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) xb[ct] = (uint32_t)((wave * TW_CELLS + ct * 16 + c) * TW_ROW_BYTES + g * 16);
        f32x4 hacc[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int ct = 0; ct < 4; ct++) xf[ct] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + xb[ct] + ks * 64));
#pragma unroll
            for (int rt = 0; rt < 4; rt++) wf[rt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(hw + (ks * 4 + rt) * 1024));
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int rt = 0; rt < 4; rt++) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    hacc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt], xf[ct], ks == 0 ? z : hacc[ct][rt], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned char *stg = lds + wave * TW_CELLS * TW_ROW_BYTES;
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
            const f32x4 b = *(const f32x4 *)(lds + TW_BIAS_OFF + (n_layers * TW_CH + rt * 16 + 4 * g) * 4);
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float v = fmaxf(hacc[ct][rt][i] + b[i], 0.0f);
                    const int chg = rt * 16 + 4 * g + i, cell = ct * 16 + c;   // chg: 0..31 policy, 32..63 value
                    *(unsigned short *)(stg + (chg * TW_CELLS + cell) * 2) = (unsigned short)(pack_bf16(v, 0.0f) & 0xFFFFu);
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

}   // namespace t16

extern "C" int yy_tower16_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                 int G, int n_layers, yy_stream_t s) {
    t16::k_tower16<<<dim3((G + TW_TB - 1) / TW_TB), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, bias, (unsigned short *)out, (unsigned short *)out_heads, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower: launch failed");
    return YY_OK;
}

// yy_tower12.hip -- the LDS-resident tower kernel (see yy_tower.hip for the design) for 12x12 boards, 128 channels.
//
// A 12x12 board has 144 cells = 4.5 MFMA column tiles of 32, so the tile is padded to 160 cells (the 16 pad columns
// read the zero row and are never written) and a workgroup holds TWO boards: wave w owns board (w & 1) and
// output-channel half (w >> 1), i.e. a 64-cout x 160-cell tile = 2 x 5 accumulators (160 registers) + the packed
// residual (80).  Both halves of a board read all 128 input channels, hence two workgroup barriers around each
// layer's epilogue.  LDS: 2 x 144 rows x 272 B activations (76.5 KB) + 4-slot x 16 KB weight ring + bias table +
// zero row = 152 KB; the weight chunk format and the numerics are those of the 8x8 kernel.
// Algorithmic FLOPs per board: 2*9*16*128*144 (stem) + layers * 2*9*128*128*144 (+ 2*128*64*144 heads).
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
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define T12_R 12
#define T12_CELLS 144
#define T12_CT 5                                            // column tiles of 32 cells (160 >= 144)
#define T12_TB 2
#define T12_CH 128
#define T12_ROW_BYTES 272
#define T12_ACT_BYTES (T12_TB * T12_CELLS * T12_ROW_BYTES)   // 78336
#define T12_CHUNK_BYTES 16384
#define T12_NSLOT 4
#define T12_RING_OFF T12_ACT_BYTES
#define T12_BIAS_OFF (T12_RING_OFF + T12_NSLOT * T12_CHUNK_BYTES)
#define T12_MAX_LAYERS 23
#define T12_ZERO_OFF (T12_BIAS_OFF + T12_MAX_LAYERS * T12_CH * 4)
#define T12_LDS_BYTES (T12_ZERO_OFF + 256)                   // 155904

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace t12 {

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
    return (uint32_t)((board * T12_CELLS + cell) * T12_ROW_BYTES + chunk * 16);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void issue_chunk(const unsigned char *wchunk, unsigned char *lds, int slot, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int piece = (r * 4 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + piece + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + T12_RING_OFF + slot * T12_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
struct Frags {
    bf16x8 x[T12_CT], w[2];
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int half, int ks,
                                           const uint32_t (&cbase)[T12_CT], int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    const unsigned char *wslot = lds + T12_RING_OFF + slot * T12_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 4096 + nh * 2048;
#pragma unroll
    for (int tt = 0; tt < T12_CT; tt++)
        f.x[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase[tt] + half * 128 + ks * 32));
#pragma unroll
    for (int nt = 0; nt < 2; nt++) f.w[nt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wslot + nt * 1024));
}
template <bool ZERO> __device__ __forceinline__ void mma10(f32x16 (&acc)[T12_CT][2], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < T12_CT; tt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
            acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[nt], f.x[tt], ZERO ? z : acc[tt][nt], 0, 0, 0);
}
__device__ __forceinline__ void interleave_hint() {   // 7 reads of the next k-step inside the 10 MFMAs of this one
#pragma unroll
    for (int j = 0; j < 3; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
}
// this lane's cell in column tile tt is tt*32 + c; cells >= 144 are padding (always the zero row)
__device__ __forceinline__ void tap_geo(int tap, int c, int board, int h, uint32_t (&cbase)[T12_CT]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int tt = 0; tt < T12_CT; tt++) {
        const int cell = tt * 32 + c;
        const int y = cell / T12_R, x = cell - y * T12_R;
        const int sy = y + dy, sx = x + dx;
        const bool ok = (cell < T12_CELLS) && ((unsigned)sy < (unsigned)T12_R) && ((unsigned)sx < (unsigned)T12_R);
        cbase[tt] = (ok ? (uint32_t)((board * T12_CELLS + sy * T12_R + sx) * T12_ROW_BYTES) : (uint32_t)T12_ZERO_OFF) + (uint32_t)(h * 16);
    }
}

template <int KS>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[T12_CT][2], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, int c, int board, int nh, int wave, int lane) {
    constexpr int NCH = (KS == 1) ? 9 : 18;
    const int h = lane >> 5;
    uint32_t cb[T12_CT];
    tap_geo(0, c, board, h, cb);
    Frags cur;
    for (int i = 0; i < NCH; i++, chunk++) {
        const int half = (KS == 1) ? 0 : (i & 1);
        if (chunk + 1 < n_chunks) {
            if (n_chunks - 2 - chunk >= 1) wait_vmcnt<4>();   // chunk+2 may stay in flight
            else wait_vmcnt<0>();
        }
        if (chunk + 1 < n_chunks || i == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // chunk+1 landed everywhere; everyone finished chunk-1 (and the epilogue)
            asm volatile("" ::: "memory");
        }
        if (chunk + 1 < n_chunks && chunk + 3 < n_chunks)
            issue_chunk(weights + (size_t)(chunk + 3) * T12_CHUNK_BYTES, lds, (chunk + 3) % T12_NSLOT, wave, lane);
        if (i == 0) load_frags(cur, lds, chunk % T12_NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        uint32_t ncb[T12_CT];
        const int ni = last ? i : i + 1;
        tap_geo((KS == 1) ? ni : (ni >> 1), c, board, h, ncb);
        const int nhalf = (KS == 1) ? 0 : (ni & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % T12_NSLOT, half, ks + 1, cb, nh, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % T12_NSLOT, nhalf, 0, ncb, nh, lane);
            if (i == 0 && ks == 0) mma10<true>(acc, cur);
            else mma10<false>(acc, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
#pragma unroll
        for (int tt = 0; tt < T12_CT; tt++) cb[tt] = ncb[tt];
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower12(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
          unsigned short *__restrict__ out, unsigned short *__restrict__ out_heads, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[T12_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int board = wave & 1, nh = wave >> 1;
    const int gb = blockIdx.x * T12_TB + board;
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * T12_CH; i += 256)
        ((float *)(lds + T12_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + T12_ZERO_OFF))[threadIdx.x] = 0u;
    if (nh == 0) {   // 5 planes -> channels 0..4 of a 16-channel zero-padded input, lane = cell (3 passes of 64)
        for (int cell = lane; cell < T12_CELLS; cell += 64) {
            float p[5];
#pragma unroll
            for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * T12_CELLS + cell] : 0.0f;
            u32x4 v0 = {pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], 0.0f), 0u};
            u32x4 z = {0u, 0u, 0u, 0u};
            *(u32x4 *)(lds + act_off(board, cell, 0)) = v0;
            *(u32x4 *)(lds + act_off(board, cell, 1)) = z;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 18 * (n_layers - 1) + (out_heads ? 1 : 0);
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * T12_CHUNK_BYTES, lds, pc % T12_NSLOT, wave, lane);
    if (n_chunks >= 3) wait_vmcnt<8>();
    else wait_vmcnt<0>();

    uint32_t res[T12_CT][2][4][2];
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc[T12_CT][2];
        if (L == 0) run_layer<1>(acc, lds, weights, chunk, n_chunks, c, board, nh, wave, lane);
        else run_layer<4>(acc, lds, weights, chunk, n_chunks, c, board, nh, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has finished reading this layer's input
        asm volatile("" ::: "memory");
        const bool conv2 = (L >= 2) && ((L & 1) == 0);
        const bool keep = (L == 0) || conv2;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = (nh * 2 + nt) * 32 + 8 * q + 4 * h;
                const f32x4 b = *(const f32x4 *)(lds + T12_BIAS_OFF + (L * T12_CH + co) * 4);
#pragma unroll
                for (int tt = 0; tt < T12_CT; tt++) {
                    f32x2 v01 = {acc[tt][nt][4 * q + 0] + b[0], acc[tt][nt][4 * q + 1] + b[1]};
                    f32x2 v23 = {acc[tt][nt][4 * q + 2] + b[2], acc[tt][nt][4 * q + 3] + b[3]};
                    if (conv2) {
                        const uint32_t r0 = res[tt][nt][q][0], r1 = res[tt][nt][q][1];
                        v01 += (f32x2){bf_lo(r0), bf_hi(r0)};
                        v23 += (f32x2){bf_lo(r1), bf_hi(r1)};
                    }
                    const uint32_t p0 = relu_pk(pack_bf16(v01[0], v01[1]));
                    const uint32_t p1 = relu_pk(pack_bf16(v23[0], v23[1]));
                    if (keep) {
                        res[tt][nt][q][0] = p0;
                        res[tt][nt][q][1] = p1;
                    }
                    const int cell = tt * 32 + c;
                    if (cell < T12_CELLS) {
                        u32x2 pk = {p0, p1};
                        *(u32x2 *)(lds + act_off(board, cell, co >> 3) + (co & 4) * 2) = pk;
                    }
                }
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (out_heads) {
        // 1x1 head convs: chunk [ks 8][nt 2][h 2][c 32][j 8]; wave half nh computes head nh (0 policy, 1 value)
        const unsigned char *hw = lds + T12_RING_OFF + (chunk % T12_NSLOT) * T12_CHUNK_BYTES + (h * 32 + c) * 16 + nh * 1024;
        f32x16 hacc[T12_CT];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(hw + ks * 2048));
#pragma unroll
            for (int tt = 0; tt < T12_CT; tt++) {
                const int cell = tt * 32 + c;
                const uint32_t xb = (cell < T12_CELLS ? (uint32_t)((board * T12_CELLS + cell) * T12_ROW_BYTES) : (uint32_t)T12_ZERO_OFF) + h * 16;
                const bf16x8 xf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + xb + ks * 32));
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                hacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, ks == 0 ? z : hacc[tt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // both halves have read the board's rows before they become the staging area
        asm volatile("" ::: "memory");
        // staging [head][channel 32][cell 144] bf16 = 18 KB inside the board's own 38 KB of rows
        unsigned char *stg = lds + board * T12_CELLS * T12_ROW_BYTES;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(lds + T12_BIAS_OFF + (n_layers * T12_CH + nh * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int tt = 0; tt < T12_CT; tt++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float v = fmaxf(hacc[tt][4 * q + i] + b[i], 0.0f);
                    const int ch = 8 * q + 4 * h + i, cell = tt * 32 + c;
                    if (cell < T12_CELLS)
                        *(unsigned short *)(stg + ((nh * 32 + ch) * T12_CELLS + cell) * 2) = (unsigned short)(pack_bf16(v, 0.0f) & 0xFFFFu);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (gb < G) {   // 18432 B per board = 18 x 1 KiB; each of the board's two waves streams 9
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = nh * 9 + i;
                const u32x4 v = *(const u32x4 *)(stg + (j * 64 + lane) * 16);
                *(u32x4 *)(out_heads + (size_t)gb * (2 * 32 * T12_CELLS) + (j * 64 + lane) * 8) = v;
            }
        }
        return;
    }
    if (gb < G) {   // activations [cell][128] bf16: 144 x 16 pieces of 16 B, half per wave
        for (int p = nh * 1152 + lane; p < (nh + 1) * 1152; p += 64) {
            const int cell = p >> 4, ch = p & 15;
            const u32x4 v = *(const u32x4 *)(lds + act_off(board, cell, ch));
            *(u32x4 *)(out + ((size_t)gb * T12_CELLS + cell) * T12_CH + ch * 8) = v;
        }
    }
}

}   // namespace t12

extern "C" int yy_tower12_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                 int G, int n_layers, yy_stream_t s) {
    t12::k_tower12<<<dim3((G + T12_TB - 1) / T12_TB), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, bias, (unsigned short *)out, (unsigned short *)out_heads, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower: launch failed");
    return YY_OK;
}

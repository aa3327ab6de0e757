// yy_tower6.hip -- the LDS-resident tower kernel (see yy_tower.hip for the design) for 6x6 boards, 128 channels.
//
// A 6x6 board has 36 cells, so EIGHT boards share a workgroup: their 288 (board, cell) columns are exactly nine
// 32-column MFMA tiles (a tile may straddle boards; every column keeps its own board for the 3x3 neighbourhood).
// Wave w owns output-channel quarter w for all 288 columns: a 32-cout x 288-column tile = 9 accumulators (144
// registers) + the packed residual (72), one weight fragment and nine activation fragments per k-step.  Every wave
// reads all input channels of every column, hence two workgroup barriers around each layer's epilogue.  LDS: 288 rows
// x 272 B (76.5 KB) + 4-slot x 16 KB weight ring + bias table + zero row = 152 KB; weight chunk format and numerics
// are those of the 8x8 kernel.  Algorithmic FLOPs per board: 2*9*16*128*36 + layers * 2*9*128*128*36 (+ 2*128*64*36).
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

#define T6_R 6
#define T6_CELLS 36
#define T6_CT 9                                             // column tiles of 32: 8 boards x 36 cells = 288 columns
#define T6_TB 8
#define T6_CH 128
#define T6_ROW_BYTES 272
#define T6_ACT_BYTES (T6_TB * T6_CELLS * T6_ROW_BYTES)   // 78336
#define T6_CHUNK_BYTES 16384
#define T6_NSLOT 4
#define T6_RING_OFF T6_ACT_BYTES
#define T6_BIAS_OFF (T6_RING_OFF + T6_NSLOT * T6_CHUNK_BYTES)
#define T6_MAX_LAYERS 23
#define T6_ZERO_OFF (T6_BIAS_OFF + T6_MAX_LAYERS * T6_CH * 4)
#define T6_LDS_BYTES (T6_ZERO_OFF + 256)                     // 155904

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace t6 {

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
__device__ __forceinline__ uint32_t act_off(int col, int chunk) {   // col = board_in_tile * 36 + cell
    return (uint32_t)(col * T6_ROW_BYTES + chunk * 16);
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
                                         (__attribute__((address_space(3))) void *)(lds + T6_RING_OFF + slot * T6_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
struct Frags {
    bf16x8 x[T6_CT], w;
};
__device__ __forceinline__ void load_frags(Frags &f, const unsigned char *lds, int slot, int half, int ks,
                                           const uint32_t (&cbase)[T6_CT], int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    const unsigned char *wslot = lds + T6_RING_OFF + slot * T6_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 4096 + nh * 1024;
#pragma unroll
    for (int tt = 0; tt < T6_CT; tt++)
        f.x[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase[tt] + half * 128 + ks * 32));
    f.w = __builtin_bit_cast(bf16x8, *(const u32x4 *)wslot);
}
template <bool ZERO> __device__ __forceinline__ void mma9(f32x16 (&acc)[T6_CT], const Frags &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < T6_CT; tt++)
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w, f.x[tt], ZERO ? z : acc[tt], 0, 0, 0);
}
__device__ __forceinline__ void interleave_hint() {   // 10 reads of the next k-step inside the 9 MFMAs of this one
#pragma unroll
    for (int j = 0; j < 5; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
}
// this lane's column in tile tt is col = tt*32 + c = board_in_tile*36 + cell; its tap neighbour is column col + dy*6 + dx
__device__ __forceinline__ void tap_geo(int tap, int c, int h, uint32_t (&cbase)[T6_CT]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int tt = 0; tt < T6_CT; tt++) {
        const int col = tt * 32 + c;
        const int cell = col % T6_CELLS;
        const int y = cell / T6_R, x = cell - y * T6_R;
        const bool ok = ((unsigned)(y + dy) < (unsigned)T6_R) && ((unsigned)(x + dx) < (unsigned)T6_R);
        cbase[tt] = (ok ? (uint32_t)((col + dy * T6_R + dx) * T6_ROW_BYTES) : (uint32_t)T6_ZERO_OFF) + (uint32_t)(h * 16);
    }
}

template <int KS>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[T6_CT], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, int c, int nh, int wave, int lane) {
    constexpr int NCH = (KS == 1) ? 9 : 18;
    const int h = lane >> 5;
    uint32_t cb[T6_CT];
    tap_geo(0, c, h, cb);
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
            issue_chunk(weights + (size_t)(chunk + 3) * T6_CHUNK_BYTES, lds, (chunk + 3) % T6_NSLOT, wave, lane);
        if (i == 0) load_frags(cur, lds, chunk % T6_NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        uint32_t ncb[T6_CT];
        const int ni = last ? i : i + 1;
        tap_geo((KS == 1) ? ni : (ni >> 1), c, h, ncb);
        const int nhalf = (KS == 1) ? 0 : (ni & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags(nxt, lds, chunk % T6_NSLOT, half, ks + 1, cb, nh, lane);
            else if (!last) load_frags(nxt, lds, (chunk + 1) % T6_NSLOT, nhalf, 0, ncb, nh, lane);
            if (i == 0 && ks == 0) mma9<true>(acc, cur);
            else mma9<false>(acc, cur);
            if (has_next) {
                interleave_hint();
                cur = nxt;
            }
        }
#pragma unroll
        for (int tt = 0; tt < T6_CT; tt++) cb[tt] = ncb[tt];
    }
}

__global__ void __launch_bounds__(256, 1)
k_tower6(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
         unsigned short *__restrict__ out, unsigned short *__restrict__ out_heads, int G, int n_layers) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[T6_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nh = wave;                                   // this wave's output-channel quarter
    const int g0 = blockIdx.x * T6_TB;                     // first board of the tile
    const int h = lane >> 5, c = lane & 31;
    constexpr int NCOL = T6_TB * T6_CELLS;                 // 288

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * T6_CH; i += 256)
        ((float *)(lds + T6_BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + T6_ZERO_OFF))[threadIdx.x] = 0u;
    for (int col = threadIdx.x; col < NCOL; col += 256) {  // 5 planes -> channels 0..4 of a 16-channel zero-padded input
        const int gb = g0 + col / T6_CELLS, cell = col % T6_CELLS;
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * T6_CELLS + cell] : 0.0f;
        u32x4 v0 = {pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], 0.0f), 0u};
        u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + act_off(col, 0)) = v0;
        *(u32x4 *)(lds + act_off(col, 1)) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 18 * (n_layers - 1) + (out_heads ? 1 : 0);
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
        if (pc < n_chunks) issue_chunk(weights + (size_t)pc * T6_CHUNK_BYTES, lds, pc % T6_NSLOT, wave, lane);
    if (n_chunks >= 3) wait_vmcnt<8>();
    else wait_vmcnt<0>();

    uint32_t res[T6_CT][4][2];
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc[T6_CT];
        if (L == 0) run_layer<1>(acc, lds, weights, chunk, n_chunks, c, nh, wave, lane);
        else run_layer<4>(acc, lds, weights, chunk, n_chunks, c, nh, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has finished reading this layer's input
        asm volatile("" ::: "memory");
        const bool conv2 = (L >= 2) && ((L & 1) == 0);
        const bool keep = (L == 0) || conv2;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = nh * 32 + 8 * q + 4 * h;       // this lane's 4 couts
            const f32x4 b = *(const f32x4 *)(lds + T6_BIAS_OFF + (L * T6_CH + co) * 4);
#pragma unroll
            for (int tt = 0; tt < T6_CT; tt++) {
                f32x2 v01 = {acc[tt][4 * q + 0] + b[0], acc[tt][4 * q + 1] + b[1]};
                f32x2 v23 = {acc[tt][4 * q + 2] + b[2], acc[tt][4 * q + 3] + b[3]};
                if (conv2) {
                    const uint32_t r0 = res[tt][q][0], r1 = res[tt][q][1];
                    v01 += (f32x2){bf_lo(r0), bf_hi(r0)};
                    v23 += (f32x2){bf_lo(r1), bf_hi(r1)};
                }
                const uint32_t p0 = relu_pk(pack_bf16(v01[0], v01[1]));
                const uint32_t p1 = relu_pk(pack_bf16(v23[0], v23[1]));
                if (keep) {
                    res[tt][q][0] = p0;
                    res[tt][q][1] = p1;
                }
                u32x2 pk = {p0, p1};
                *(u32x2 *)(lds + act_off(tt * 32 + c, co >> 3) + (co & 4) * 2) = pk;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (out_heads) {
        // 1x1 head convs: chunk [ks 8][nt 2][h 2][c 32][j 8]; wave w: head (w & 1), column tiles 0-4 (w < 2) or 5-8
        const int head = wave & 1, t0 = (wave >> 1) * 5, nt_cnt = (wave >> 1) ? 4 : 5;
        const unsigned char *hw = lds + T6_RING_OFF + (chunk % T6_NSLOT) * T6_CHUNK_BYTES + (h * 32 + c) * 16 + head * 1024;
        f32x16 hacc[5];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(hw + ks * 2048));
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const int col = min((t0 + t) * 32 + c, NCOL - 1);   // the 5th tile of waves 2,3 is a duplicate, never stored
                const bf16x8 xf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + act_off(col, 0) + h * 16 + ks * 32));
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                hacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, ks == 0 ? z : hacc[t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everyone has read the activations before they become the staging area
        asm volatile("" ::: "memory");
        // staging [board 8][head 2][channel 32][cell 36] bf16 = 36 KB = the global layout of these 8 boards
        unsigned char *stg = lds;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(lds + T6_BIAS_OFF + (n_layers * T6_CH + head * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (t < nt_cnt) {
                    const int col = (t0 + t) * 32 + c, bd = col / T6_CELLS, cell = col % T6_CELLS;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = fmaxf(hacc[t][4 * q + i] + b[i], 0.0f);
                        const int ch = 8 * q + 4 * h + i;
                        *(unsigned short *)(stg + ((bd * 2 + head) * 32 + ch) * (T6_CELLS * 2) + cell * 2) =
                            (unsigned short)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int PER_BOARD = 2 * 32 * T6_CELLS * 2;   // 4608 B
        for (int p = threadIdx.x; p < T6_TB * PER_BOARD / 16; p += 256) {
            if (g0 + (p * 16) / PER_BOARD < G)
                *(u32x4 *)((unsigned char *)out_heads + (size_t)g0 * PER_BOARD + (size_t)p * 16) = *(const u32x4 *)(stg + p * 16);
        }
        return;
    }
    for (int p = threadIdx.x; p < NCOL * 16; p += 256) {   // activations [column][128] bf16
        const int col = p >> 4, ch = p & 15;
        if (g0 + col / T6_CELLS < G)
            *(u32x4 *)(out + ((size_t)g0 * T6_CELLS + col) * T6_CH + ch * 8) = *(const u32x4 *)(lds + act_off(col, ch));
    }
}

}   // namespace t6

extern "C" int yy_tower6_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                 int G, int n_layers, yy_stream_t s) {
    t6::k_tower6<<<dim3((G + T6_TB - 1) / T6_TB), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, bias, (unsigned short *)out, (unsigned short *)out_heads, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower: launch failed");
    return YY_OK;
}

// yy_towerq.hip -- the LDS-resident tower kernel (see yy_tower.hip for the design) in its "output-channel quarter"
// form: wave w owns output channels [32w, 32w+32) for ALL columns of the workgroup.  Template <R, TB>: TB boards of
// R x R cells share a workgroup; their TB*R*R (board, cell) columns are CT = TB*R*R/32 MFMA column tiles (a tile may
// straddle boards; every column keeps its own board for the 3x3 neighbourhood).  Per k-step a wave reads CT activation
// fragments + ONE weight fragment for CT MFMAs.  Every wave reads all input channels of every column, hence two
// workgroup barriers around each layer's epilogue.  Weight chunk format, numerics and results are those of the 8x8
// kernel (same accumulation order, same epilogue -> bit-identical outputs).
//
// Instantiations:
//   <6, 8>  6x6 boards: 8 boards = 288 columns = 9 tiles; 9 accumulators (144 registers) + packed residual (72);
//           LDS 288 rows x 272 B (76.5 KB) + 4-slot x 16 KB ring + bias + zero row = 152 KB.  The 6x6 evaluator.
//   <12, 2> 12x12 boards: two boards = 288 columns = 9 full tiles, the same shape as <6, 8>.  The 12x12 evaluator.
//   <8, 1>  8x8 boards, ONE board per workgroup (2 tiles): the low-latency form for small batches (arena matches,
//   <8, 2>  single-board MCTS.search, a few hundred concurrent games), where yy_tower.hip's 4-boards-per-workgroup
//           grid leaves most CUs idle and a step costs one full workgroup latency whatever G is.  A quarter of the MFMA
//           work per wave, more LDS reads per MFMA (1.5 / 1.25 instead of 0.75) -- the right trade only while the
//           chip is not full; yy_tower.hip stays the kernel for G >= 1024.
// Algorithmic FLOPs per board: 2*9*16*128*R*R + layers * 2*9*128*128*R*R (+ 2*128*64*R*R).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define TQ_CH 128
#define TQ_ROW_BYTES 272
#define TQ_CHUNK_BYTES 16384
#define TQ_NSLOT 4
#define TQ_MAX_LAYERS 23

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace tq {

template <int R_, int TB_> struct Geo {
    static constexpr int R = R_, TB = TB_, CELLS = R_ * R_, NCOL = TB_ * R_ * R_, CT = NCOL / 32;
    static_assert(NCOL % 32 == 0, "columns must fill whole MFMA tiles");
    static constexpr int ACT_BYTES = NCOL * TQ_ROW_BYTES;
    static constexpr int RING_OFF = ACT_BYTES;
    static constexpr int BIAS_OFF = RING_OFF + TQ_NSLOT * TQ_CHUNK_BYTES;
    static constexpr int ZERO_OFF = BIAS_OFF + TQ_MAX_LAYERS * TQ_CH * 4;
    static constexpr int LDS_BYTES = ZERO_OFF + 256;
};


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
__device__ __forceinline__ uint32_t act_off(int col, int chunk) {   // col = board_in_workgroup * CELLS + cell
    return (uint32_t)(col * TQ_ROW_BYTES + chunk * 16);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
template <class GEO>
__device__ __forceinline__ void issue_chunk(const unsigned char *wchunk, unsigned char *lds, int slot, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int piece = (r * 4 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + piece + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + GEO::RING_OFF + slot * TQ_CHUNK_BYTES + piece),
                                         16, 0, 0);
    }
}
template <int CT> struct Frags {
    bf16x8 x[CT], w;
};
template <class GEO>
__device__ __forceinline__ void load_frags(Frags<GEO::CT> &f, const unsigned char *lds, int slot, int half, int ks,
                                           const uint32_t (&cbase)[GEO::CT], int nh, int lane) {
    const int h = lane >> 5, c = lane & 31;
    const unsigned char *wslot = lds + GEO::RING_OFF + slot * TQ_CHUNK_BYTES + (h * 32 + c) * 16 + ks * 4096 + nh * 1024;
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++)
        f.x[tt] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + cbase[tt] + half * 128 + ks * 32));
    f.w = __builtin_bit_cast(bf16x8, *(const u32x4 *)wslot);
}
template <int CT, bool ZERO> __device__ __forceinline__ void mma_ct(f32x16 (&acc)[CT], const Frags<CT> &f) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < CT; tt++)
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w, f.x[tt], ZERO ? z : acc[tt], 0, 0, 0);
}
// the CT+1 reads of the next k-step inside the CT MFMAs of this one: (1 MFMA + 2 reads) groups, then the MFMAs left
template <int CT> __device__ __forceinline__ void interleave_hint() {
    constexpr int NR = CT + 1, PAIRS = NR / 2, ODD = NR & 1, REST = CT - PAIRS - ODD;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if (ODD) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
}
// Per-lane geometry, computed once per kernel: this lane's column in tile tt is col = tt*32 + c = board*CELLS + cell, its
// tap neighbour is column col + dy*R + dx.  rowbase = LDS offset of the column's own row (+ h*16), okmask = 9-bit mask of the
// taps whose neighbour is on the board.  The per-chunk tap geometry is then an and / compare / add / select per tile.
template <class GEO> struct LaneGeo {
    uint32_t rowbase[GEO::CT], okmask[GEO::CT], zbase;
};
template <class GEO>
__device__ __forceinline__ void make_lane_geo(LaneGeo<GEO> &g, int c, int h) {
    g.zbase = (uint32_t)GEO::ZERO_OFF + (uint32_t)(h * 16);
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) {
        const int col = tt * 32 + c;
        const int cell = col % GEO::CELLS;
        const int y = cell / GEO::R, x = cell - y * GEO::R;
        g.rowbase[tt] = (uint32_t)(col * TQ_ROW_BYTES) + (uint32_t)(h * 16);
        uint32_t m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
            if (((unsigned)sy < (unsigned)GEO::R) && ((unsigned)sx < (unsigned)GEO::R)) m |= 1u << tap;
        }
        g.okmask[tt] = m;
    }
}
template <class GEO>
__device__ __forceinline__ void tap_geo(int tap, const LaneGeo<GEO> &g, uint32_t (&cbase)[GEO::CT]) {
    const int shift = ((tap / 3 - 1) * GEO::R + (tap % 3 - 1)) * TQ_ROW_BYTES;   // wave-uniform
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int tt = 0; tt < GEO::CT; tt++) cbase[tt] = (g.okmask[tt] & bit) ? g.rowbase[tt] + (uint32_t)shift : g.zbase;
}

template <class GEO, int KS>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[GEO::CT], unsigned char *lds, const unsigned char *weights, int &chunk,
                                          int n_chunks, const LaneGeo<GEO> &geo, int nh, int wave, int lane) {
    constexpr int NCH = (KS == 1) ? 9 : 18, CT = GEO::CT;
    uint32_t cb[CT];
    tap_geo<GEO>(0, geo, cb);
    Frags<CT> cur;
    for (int i = 0; i < NCH; i++, chunk++) {
        const int half = (KS == 1) ? 0 : (i & 1);
        if (chunk + 1 < n_chunks) {
            if (n_chunks - 2 - chunk >= 1) wait_vmcnt<4>();   // chunk+2 may stay in flight
            else wait_vmcnt<0>();
        }
        // The chunk layout [ks 4][nt 4][1 KB] and issue_chunk's piece order make wave w load exactly the pieces (nt == w) it
        // reads itself: the weight ring is wave-private, the counted vmcnt above is all a chunk needs -- no workgroup barrier.
        // Barriers remain where the waves exchange activations: at the start of a layer and around its epilogue.
        if (i == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // the previous layer's epilogue (or the prologue) is visible to every wave
            asm volatile("" ::: "memory");
        }
        if (chunk + 1 < n_chunks && chunk + 3 < n_chunks)
            issue_chunk<GEO>(weights + (size_t)(chunk + 3) * TQ_CHUNK_BYTES, lds, (chunk + 3) % TQ_NSLOT, wave, lane);
        if (i == 0) load_frags<GEO>(cur, lds, chunk % TQ_NSLOT, 0, 0, cb, nh, lane);
        const bool last = (i == NCH - 1);
        uint32_t ncb[CT];
        const int ni = last ? i : i + 1;
        tap_geo<GEO>((KS == 1) ? ni : (ni >> 1), geo, ncb);
        const int nhalf = (KS == 1) ? 0 : (ni & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frags<CT> nxt;
            const bool has_next = (ks + 1 < KS) || !last;
            if (ks + 1 < KS) load_frags<GEO>(nxt, lds, chunk % TQ_NSLOT, half, ks + 1, cb, nh, lane);
            else if (!last) load_frags<GEO>(nxt, lds, (chunk + 1) % TQ_NSLOT, nhalf, 0, ncb, nh, lane);
            if (i == 0 && ks == 0) mma_ct<CT, true>(acc, cur);
            else mma_ct<CT, false>(acc, cur);
            if (has_next) {
                interleave_hint<CT>();
                cur = nxt;
            }
        }
#pragma unroll
        for (int tt = 0; tt < CT; tt++) cb[tt] = ncb[tt];
    }
}

template <int R_, int TB_>
__global__ void __launch_bounds__(256, 1)
k_towerq(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const float *__restrict__ bias,
         unsigned short *__restrict__ out, unsigned short *__restrict__ out_heads, int G, int n_layers) {
    using GEO = Geo<R_, TB_>;
    constexpr int CT = GEO::CT, CELLS = GEO::CELLS, NCOL = GEO::NCOL, TB = GEO::TB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GEO::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nh = wave;                                   // this wave's output-channel quarter
    const int g0 = blockIdx.x * TB;                        // first board of the workgroup
    const int h = lane >> 5, c = lane & 31;

    for (int i = threadIdx.x; i < (n_layers + (out_heads ? 1 : 0)) * TQ_CH; i += 256)
        ((float *)(lds + GEO::BIAS_OFF))[i] = bias[i];
    if (threadIdx.x < 64) ((uint32_t *)(lds + GEO::ZERO_OFF))[threadIdx.x] = 0u;
    for (int col = threadIdx.x; col < NCOL; col += 256) {  // 5 planes -> channels 0..4 of a 16-channel zero-padded input
        const int gb = g0 + col / CELLS, cell = col % CELLS;
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = (gb < G) ? planes[((size_t)gb * 5 + k) * CELLS + cell] : 0.0f;
        u32x4 v0 = {pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], 0.0f), 0u};
        u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + act_off(col, 0)) = v0;
        *(u32x4 *)(lds + act_off(col, 1)) = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int n_chunks = 9 + 18 * (n_layers - 1) + (out_heads ? 1 : 0);
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
        if (pc < n_chunks) issue_chunk<GEO>(weights + (size_t)pc * TQ_CHUNK_BYTES, lds, pc % TQ_NSLOT, wave, lane);
    if (n_chunks >= 3) wait_vmcnt<8>();
    else wait_vmcnt<0>();

    LaneGeo<GEO> geo;
    make_lane_geo<GEO>(geo, c, h);
    uint32_t res[CT][4][2];
    int chunk = 0;
    for (int L = 0; L < n_layers; L++) {
        f32x16 acc[CT];
        if (L == 0) run_layer<GEO, 1>(acc, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        else run_layer<GEO, 4>(acc, lds, weights, chunk, n_chunks, geo, nh, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has finished reading this layer's input
        asm volatile("" ::: "memory");
        const bool conv2 = (L >= 2) && ((L & 1) == 0);
        const bool keep = (L == 0) || conv2;
        // no MFMA runs in the epilogue, so stalls are paid in full: the lane's four bias vectors are fetched back to back
        // (one LDS latency), the pair conversion is one v_cvt_pk_bf16_f32, the residual branch is resolved once per layer
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bq[q] = *(const f32x4 *)(lds + GEO::BIAS_OFF + (L * TQ_CH + nh * 32 + 8 * q + 4 * h) * 4);
        auto epilogue = [&](auto with_res) {
            constexpr bool RES = decltype(with_res)::value;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = nh * 32 + 8 * q + 4 * h;       // this lane's 4 couts
                const f32x4 b = bq[q];
#pragma unroll
                for (int tt = 0; tt < CT; tt++) {
                    f32x2 v01 = {acc[tt][4 * q + 0] + b[0], acc[tt][4 * q + 1] + b[1]};
                    f32x2 v23 = {acc[tt][4 * q + 2] + b[2], acc[tt][4 * q + 3] + b[3]};
                    if (RES) {
                        const uint32_t r0 = res[tt][q][0], r1 = res[tt][q][1];
                        v01 += (f32x2){bf_lo(r0), bf_hi(r0)};
                        v23 += (f32x2){bf_lo(r1), bf_hi(r1)};
                    }
                    const uint32_t p0 = relu_pk(__builtin_bit_cast(uint32_t, __builtin_convertvector(v01, bf16x2)));
                    const uint32_t p1 = relu_pk(__builtin_bit_cast(uint32_t, __builtin_convertvector(v23, bf16x2)));
                    if (keep) {
                        res[tt][q][0] = p0;
                        res[tt][q][1] = p1;
                    }
                    u32x2 pk = {p0, p1};
                    *(u32x2 *)(lds + act_off(tt * 32 + c, co >> 3) + (co & 4) * 2) = pk;
                }
            }
        };
        if (conv2) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (out_heads) {
        // 1x1 head convs: chunk [ks 8][nt 2][h 2][c 32][j 8]; wave w: head (w & 1), the first HT column tiles (w < 2)
        // or the remaining CT - HT
        constexpr int HT = (CT + 1) / 2;
        const int head = wave & 1, t0 = (wave >> 1) * HT, nt_cnt = (wave >> 1) ? CT - HT : HT;
        const unsigned char *hw = lds + GEO::RING_OFF + (chunk % TQ_NSLOT) * TQ_CHUNK_BYTES + (h * 32 + c) * 16 + head * 1024;
        f32x16 hacc[HT];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(hw + ks * 2048));
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const int col = min((t0 + t) * 32 + c, NCOL - 1);   // a surplus tile of waves 2,3 is a duplicate, never stored
                const bf16x8 xf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(lds + act_off(col, 0) + h * 16 + ks * 32));
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                hacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, ks == 0 ? z : hacc[t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everyone has read the activations before they become the staging area
        asm volatile("" ::: "memory");
        // staging [board TB][head 2][channel 32][cell CELLS] bf16 = the global layout of these boards
        unsigned char *stg = lds;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(lds + GEO::BIAS_OFF + (n_layers * TQ_CH + head * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int t = 0; t < HT; t++) {
                if (t < nt_cnt) {
                    const int col = (t0 + t) * 32 + c, bd = col / CELLS, cell = col % CELLS;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = fmaxf(hacc[t][4 * q + i] + b[i], 0.0f);
                        const int ch = 8 * q + 4 * h + i;
                        *(unsigned short *)(stg + ((bd * 2 + head) * 32 + ch) * (CELLS * 2) + cell * 2) =
                            (unsigned short)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int PER_BOARD = 2 * 32 * CELLS * 2;      // 4608 B at 6x6, 8192 B at 8x8
        for (int p = threadIdx.x; p < TB * PER_BOARD / 16; p += 256) {
            if (g0 + (p * 16) / PER_BOARD < G)
                *(u32x4 *)((unsigned char *)out_heads + (size_t)g0 * PER_BOARD + (size_t)p * 16) = *(const u32x4 *)(stg + p * 16);
        }
        return;
    }
    for (int p = threadIdx.x; p < NCOL * 16; p += 256) {   // activations [column][128] bf16
        const int col = p >> 4, ch = p & 15;
        if (g0 + col / CELLS < G)
            *(u32x4 *)(out + ((size_t)g0 * CELLS + col) * TQ_CH + ch * 8) = *(const u32x4 *)(lds + act_off(col, ch));
    }
}

}   // namespace tq

template <int R_, int TB_>
static int launch_q(const float *planes, const void *weights, const float *bias, void *out, void *out_heads, int G,
                    int n_layers, yy_stream_t s) {
    tq::k_towerq<R_, TB_><<<dim3((G + TB_ - 1) / TB_), dim3(256), 0, (hipStream_t)s>>>(
        planes, (const unsigned char *)weights, bias, (unsigned short *)out, (unsigned short *)out_heads, G, n_layers);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower: launch failed");
    return YY_OK;
}

extern "C" int yy_tower6_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                int G, int n_layers, yy_stream_t s) {
    return launch_q<6, 8>(planes, weights, bias, out, out_heads, G, n_layers, s);
}

// 12x12 boards: two boards = 288 columns = nine full tiles
extern "C" int yy_tower12q_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                  int G, int n_layers, yy_stream_t s) {
    return launch_q<12, 2>(planes, weights, bias, out, out_heads, G, n_layers, s);
}

// 8x8 boards, tb = 1 or 2 boards per workgroup (small batches; yy_tower.hip picks)
extern "C" int yy_tower8q_launch(const float *planes, const void *weights, const float *bias, void *out, void *out_heads,
                                 int G, int n_layers, int tb, yy_stream_t s) {
    if (tb == 1) return launch_q<8, 1>(planes, weights, bias, out, out_heads, G, n_layers, s);
    return launch_q<8, 2>(planes, weights, bias, out, out_heads, G, n_layers, s);
}

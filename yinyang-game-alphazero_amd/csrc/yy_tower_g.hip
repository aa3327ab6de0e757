// yy_tower_g.hip -- the split-f16 (float32-accurate) evaluator tower, GENERAL form: any R x C board with at most 144 cells,
// 32 / 64 / 96 / 128 channels, any number of residual blocks up to 10 (train_alphazero.py:35-36 takes any --rows / --cols,
// ai/neural_network.py:39 any num_channels / num_res_blocks).  Evaluator mode "f16x3" (network.BatchedEvaluator).
//
// Reference computation: YinYangNeuralNetwork.forward, stem + residual blocks + the two 1x1 head convolutions
// (src/yin_yang/ai/neural_network.py:16-33, 94-119), float32 on the CPU.  Numerics as in yy_tower_h3.hip: every activation
// x and weight w is a PAIR of float16 numbers hi = f16(x), lo = f16(x - hi) (22 significant bits), weights stored times 2^kw
// and activations times 2^ka so that the lo parts stay normal; per output element one f32 accumulator takes w_hi*x_hi and a
// second one w_lo*x_hi + w_hi*x_lo; epilogue fma(acc1 + acc2, 2^-kw, bias) (+ f32 residual), ReLU, split again.
//
// What is different from yy_tower_h3r.hip (32x32x16 MFMA, 32-column tiles, square boards of 6 / 8 / 12 only):
//   * v_mfma_f32_16x16x32_f16: the output tile of a wave is 32 output channels (two 16-row M blocks) x NB column blocks of
//     16 (board, cell) columns.  Columns come in blocks of 16 instead of 32, so 6x6 (4 boards = 144 columns) and 12x12 (144)
//     are exactly 9 blocks -- no padding tile -- and any other board wastes less than one block per workgroup.  The chip holds
//     a higher clock on this MFMA shape at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).
//   * the (board, cell) geometry is computed at run time from R, C: non-square boards, any number of boards per workgroup.
//   * one k-step of 32 input channels per weight chunk (one MFMA K), so a chunk is [wave][M block 2][part 2][lane 64][8 f16]
//     = 4 KB per wave, loaded global -> register in MFMA operand order and kept in a ring of D chunks (as in yy_tower_h3r).
//   * LDS rows of 2*CH bytes with a stride of 288 B (CH = 96, 128) / 160 B (CH = 32, 64): stride/16 = 2 mod 8 makes the
//     16-column x 4-k-group ds_read_b128 pattern of this MFMA's B operand conflict-free; an off-board tap reads the zero rows
//     at the offset (its own row address mod 256): the bank the lane would have used anyway, so padding reads do not collide
//     with the valid lanes either (the single shared zero slot of yy_tower_h3r.hip cost 22 % extra LDS cycles).
//   * activation fragments are refreshed IN PLACE, block by block, right after their last use (the next chunk's fragment of
//     column block nb is read while the MFMAs of block nb + 1 run): 8*NB registers instead of 16*NB.
//   * bias rows come from scalar loads (no LDS table): NB = 9 with the f32 residual parked in LDS needs 154 of the 160 KB.
// Accumulation order per output element is the same for every (NB, boards-per-workgroup) form: identical bits whichever form
// evaluates a board (tests/test_gpu_network.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/yy_engine.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

extern "C" int yy_tower_set_err(int code, const char *msg);

namespace tg {

template <int NW_, int NB_, int D_> struct Geo {
    static constexpr int NW = NW_, NB = NB_, D = D_, CH = 32 * NW_, NCOL = 16 * NB_;
    static constexpr int RS = (NW_ >= 3) ? 288 : 160;              // row stride: data 2*CH bytes, stride/16 = 2 (mod 8)
    static constexpr int ZERO_OFF = NCOL * RS;                     // two zero rows' worth (512 B) behind the columns of a part
    static constexpr int PART_BYTES = ZERO_OFF + 512;
    static constexpr int RES_OFF = 2 * PART_BYTES;                 // f32 residual: [wave][M block 2][column block NB] x 1 KB, lane-private 16-B slots
    static constexpr int LDS_BYTES = RES_OFF + NW_ * 2 * NB_ * 1024;
    static constexpr int CHUNK_BYTES = NW_ * 4096;
    static_assert(9 % D_ == 0, "ring position = tap % D");
    static_assert(ZERO_OFF % 256 == 0 && PART_BYTES % 256 == 0, "zero rows / parts keep the bank phase");
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

// this wave's fragments of one weight chunk: [M block 2]{hi, lo}, 16 registers
struct WChunk {
    f16x8 h[2], l[2];
};
// The four 1 KB fragment loads of a chunk: wbase = weights + chunk * CHUNK_BYTES (uniform), voff = wave * 4096 + lane * 16.
// Plain loads that the compiler counts itself: it waits for a ring entry with a counted vmcnt (the 4 * (D - 1) younger loads
// stay in flight) and a sched_group_barrier keeps the refill right behind the chunk that freed the entry.  (yy_tower_h3r.hip
// issues these loads from inline asm with a hand-counted vmcnt: under register pressure the compiler then copies or spills
// the destination registers BEFORE the data has landed -- the failure this form cannot have.)
__device__ __forceinline__ void load_w(WChunk &w, const unsigned char *wbase, uint32_t voff) {
    const u32x4 *p = (const u32x4 *)(wbase + voff);
    w.h[0] = __builtin_bit_cast(f16x8, p[0]);
    w.l[0] = __builtin_bit_cast(f16x8, p[64]);
    w.h[1] = __builtin_bit_cast(f16x8, p[128]);
    w.l[1] = __builtin_bit_cast(f16x8, p[192]);
}

// per-lane geometry: column nb*16 + n (n = lane & 15), board = column / cells, cell = column % cells; the lane's k group
// (lane >> 4: input channels [8*kg, 8*kg+8) of a 32-channel chunk) is folded into the row base
template <class GEO> struct LaneGeo {
    uint32_t rowbase[GEO::NB], okmask[GEO::NB];
};
template <class GEO> __device__ __forceinline__ void make_lane_geo(LaneGeo<GEO> &g, int lane, int R, int C, int n_valid_cols) {
    const int n = lane & 15, kg = lane >> 4, cells = R * C;
#pragma unroll
    for (int nb = 0; nb < GEO::NB; nb++) {
        const int col = nb * 16 + n;
        const int cell = col % cells;
        const int y = cell / C, x = cell - y * C;
        g.rowbase[nb] = (uint32_t)(col * GEO::RS + kg * 16);
        uint32_t m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
            if (((unsigned)sy < (unsigned)R) && ((unsigned)sx < (unsigned)C)) m |= 1u << tap;
        }
        g.okmask[nb] = (col < n_valid_cols) ? m : 0u;
    }
}
// LDS byte address (hi part, channel-group offset excluded) column block nb reads for TAP: its shifted row, or -- off the
// board -- the zero rows at (that address mod 256): same bank, no conflict with the lanes that read real rows
template <class GEO, int TAP>
__device__ __forceinline__ uint32_t tap_addr(int nb, int crs, const LaneGeo<GEO> &g) {
    const int shift = (TAP / 3 - 1) * crs + (TAP % 3 - 1) * GEO::RS;   // wave-uniform; crs = C * RS
    const uint32_t a = g.rowbase[nb] + (uint32_t)shift;
    return (g.okmask[nb] & (1u << TAP)) ? a : ((a & 255u) | (uint32_t)GEO::ZERO_OFF);
}

// Activation fragments of one chunk ({hi, lo} x 4 registers per column block), refreshed IN PLACE: right after the MFMAs of block
// nb its slot takes the NEXT chunk's fragment of block nb, i.e. a fragment is requested a whole chunk (6 * NB MFMAs) before its use
// and 8 * NB registers hold activations.  (A shorter rotating window of 3-4 blocks -- 24-32 registers -- was the first form of this
// kernel; with one scheduling region per chunk the full window measures 1.8 % faster at NB = 8 and still fits at NB = 9.)
template <int NB> struct XWin {
    static constexpr int P = NB;
    f16x8 h[P], l[P];
};
template <class GEO> __device__ __forceinline__ void load_x1(XWin<GEO::NB> &f, int slot, const unsigned char *lds, uint32_t cb) {
    f.h[slot] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cb));
    f.l[slot] = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + cb + GEO::PART_BYTES));
}
// acc1 += w_hi * x_hi ;  acc2 += w_lo * x_hi + w_hi * x_lo for the two M blocks of one column block (6 MFMAs).  Two
// accumulators because every MFMA rounds its accumulator once: the large sum is rounded once per k-step (as in an f32 dot
// product) and the 2^-11-times-smaller corrections round among themselves.
template <bool ZERO>
__device__ __forceinline__ void mma6(f32x4 (&acc1)[2], f32x4 (&acc2)[2], const WChunk &w, const f16x8 xh, const f16x8 xl) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.l[mb], xh, ZERO ? z : acc2[mb], 0, 0, 0);
        acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[mb], xh, ZERO ? z : acc1[mb], 0, 0, 0);
        acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[mb], xl, a, 0, 0, 0);
    }
}

// One chunk = tap J of input-channel group kq: 6 * NB MFMAs.  lds_cur / lds_next = LDS base + the channel-group offset of this
// chunk / of the chunk after it (the layer's last chunk prefetches in-bounds bytes nobody uses).
template <class GEO, int J, bool ZERO>
__device__ __forceinline__ void run_chunk(f32x4 (&acc1)[GEO::NB][2], f32x4 (&acc2)[GEO::NB][2], const WChunk &w, XWin<GEO::NB> &X,
                                          const unsigned char *lds_cur, const unsigned char *lds_next, int crs, const LaneGeo<GEO> &geo) {
    constexpr int NB = GEO::NB, P = XWin<NB>::P;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        mma6<ZERO>(acc1[nb], acc2[nb], w, X.h[nb % P], X.l[nb % P]);
        if (nb + P < NB) load_x1<GEO>(X, nb % P, lds_cur, tap_addr<GEO, J>(nb + P, crs, geo));
        else load_x1<GEO>(X, nb % P, lds_next, tap_addr<GEO, (J + 1) % 9>(nb + P - NB, crs, geo));
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

// One layer: KQ input-channel groups of 32 (stem: 1, the 5 planes zero-padded to 32 channels) x 9 taps, ring position =
// tap % D (compile-time).  After its last use a ring entry is refilled with the chunk D further down the stream (the next
// group / the next layer).  Weight stream order: [layer][kq][tap].
template <class GEO, bool STEM>
__device__ __forceinline__ void run_layer(f32x4 (&acc1)[GEO::NB][2], f32x4 (&acc2)[GEO::NB][2], WChunk (&W)[GEO::D], const unsigned char *lds,
                                          const unsigned char *weights, uint32_t voff, int &chunk, int n_tower, int crs,
                                          const LaneGeo<GEO> &geo) {
    constexpr int KQ = STEM ? 1 : GEO::NW, NB = GEO::NB, D = GEO::D, P = XWin<NB>::P;
    XWin<NB> X;
#pragma unroll
    for (int nb = 0; nb < P; nb++) load_x1<GEO>(X, nb, lds, tap_addr<GEO, 0>(nb, crs, geo));
    auto step = [&](auto jt, auto zt, const unsigned char *lds_cur, const unsigned char *lds_next) {
        constexpr int J = decltype(jt)::value, S = J % D;
        run_chunk<GEO, J, decltype(zt)::value>(acc1, acc2, W[S], X, lds_cur, lds_next, crs, geo);
        // refill the ring entry just consumed (the tail of the stream re-reads the last chunk: uniform vmcnt)
        load_w(W[S], weights + (size_t)min(chunk + D, n_tower - 1) * GEO::CHUNK_BYTES, voff);
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);     // the four fragment loads stay here, behind this chunk's MFMAs
        // one scheduling region per chunk: the sched_group_barrier solver is superlinear in region size -- with the nine chunks of a
        // channel group in one region this file took 8 minutes to compile, with this line 20 seconds; same kernel time
        __builtin_amdgcn_sched_barrier(0);
        chunk++;
    };
    auto taps = [&](auto zt, const unsigned char *cur, const unsigned char *nxt) {
        step(std::integral_constant<int, 0>{}, zt, cur, cur);              // zt: the layer's first chunk starts the sums
        step(std::integral_constant<int, 1>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 2>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 3>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 4>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 5>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 6>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 7>{}, std::false_type{}, cur, cur);
        step(std::integral_constant<int, 8>{}, std::false_type{}, cur, nxt);
    };
    taps(std::true_type{}, lds, lds + 64);                                 // channel group 0, peeled: no zero / non-zero branch
#pragma unroll 1
    for (int kq = 1; kq < KQ; kq++) taps(std::false_type{}, lds + kq * 64, lds + (kq + 1) * 64);
}

template <int NW_, int NB_, int D_>
__global__ void __launch_bounds__(64 * NW_, 1)
k_tower_g(const float *__restrict__ planes, const unsigned char *__restrict__ weights, const unsigned char *__restrict__ head_w,
          const float *__restrict__ bias, const float *__restrict__ head_bias, float *__restrict__ out, float *__restrict__ out_heads,
          const int *__restrict__ rows, const int *__restrict__ n_rows, int G, int R, int C, int TB, int n_layers, float in_scale,
          float acc_scale, float head_scale, float out_scale, int gate_lo, int gate_hi) {
    using GEO = Geo<NW_, NB_, D_>;
    constexpr int NW = GEO::NW, NB = GEO::NB, D = GEO::D, CH = GEO::CH, RS = GEO::RS, NT = 64 * NW_;
    __shared__ __attribute__((aligned(256))) unsigned char lds[GEO::LDS_BYTES];
    const int n_live = n_rows ? min(*n_rows, G) : G;
    if (n_live <= gate_lo || n_live > gate_hi) return;     // the launch only runs for row counts in (gate_lo, gate_hi]
    const int g0 = blockIdx.x * TB;                        // first dense row of the workgroup
    if (g0 >= n_live) return;                              // whole workgroup, before any barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cells = R * C, ncol = TB * cells;            // ncol <= 16 * NB (checked on the host)
    const int n_tower = 9 + 9 * NW * (n_layers - 1), crs = C * RS;

    // the weight stream starts first: D chunks in flight before anything else is touched
    const uint32_t voff = (uint32_t)(wave * 4096 + lane * 16);
    WChunk W[D];
#pragma unroll
    for (int j = 0; j < D; j++) load_w(W[j], weights + (size_t)min(j, n_tower - 1) * GEO::CHUNK_BYTES, voff);

    for (int t = threadIdx.x; t < 128; t += NT)   // the zero rows of both parts (512 B each)
        ((u32x2 *)(lds + GEO::ZERO_OFF + (t >> 6) * GEO::PART_BYTES))[t & 63] = (u32x2){0u, 0u};
    for (int col = threadIdx.x; col < GEO::NCOL; col += NT) {   // 5 planes -> channels 0..4 of a 32-channel zero-padded input
        const int gb = g0 + col / cells, cell = col % cells;
        const bool live = col < ncol && gb < n_live;
        const int src = live ? (rows ? rows[gb] : gb) : 0;
        float p[6];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = live ? planes[((size_t)src * 5 + k) * cells + cell] * in_scale : 0.0f;
        p[5] = 0.0f;
        uint32_t hi[3], lo[3];
#pragma unroll
        for (int k = 0; k < 3; k++) split_pair((f32x2){p[2 * k], p[2 * k + 1]}, hi[k], lo[k]);
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4 *)(lds + col * RS) = (u32x4){hi[0], hi[1], hi[2], 0u};
        *(u32x4 *)(lds + GEO::PART_BYTES + col * RS) = (u32x4){lo[0], lo[1], lo[2], 0u};
#pragma unroll
        for (int q = 1; q < 4; q++) {
            *(u32x4 *)(lds + col * RS + 16 * q) = z;
            *(u32x4 *)(lds + GEO::PART_BYTES + col * RS + 16 * q) = z;
        }
    }

    LaneGeo<GEO> geo;
    make_lane_geo<GEO>(geo, lane, R, C, ncol);
    const int n16 = lane & 15, kg = lane >> 4;
    unsigned char *res_lds = lds + GEO::RES_OFF + (wave * 2 * NB) * 1024 + lane * 16;   // slot (mb * NB + nb) * 1 KB, lane-private
    int chunk = 0;
    // (acc1 + acc2) * 2^-kw + bias (+ residual) + ReLU in f32, split again, back to LDS (activations, bias rows and residual
    // live in the 2^ka-scaled domain).  A lane holds output channels co0 + 0..3 (co0 = 32*wave + 16*mb + 4*kg) of column
    // nb*16 + n16.  KEEP: the output is a block input x, kept for the skip.
    auto epilogue = [&](const int L, f32x4 (&acc1)[NB][2], f32x4 (&acc2)[NB][2], auto conv2_tag, auto keep_tag) {
        constexpr bool CONV2 = decltype(conv2_tag)::value, KEEP = decltype(keep_tag)::value;
        const float *bl = bias + L * CH + wave * 32;          // uniform address: scalar loads, pinned in SGPRs before the selects
        float sb[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            sb[k] = bl[k];
            asm volatile("" : "+s"(sb[k]));
        }
        f32x4 bq[2];
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                bq[mb][i] = kg == 0 ? sb[mb * 16 + i] : kg == 1 ? sb[mb * 16 + 4 + i] : kg == 2 ? sb[mb * 16 + 8 + i] : sb[mb * 16 + 12 + i];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const uint32_t rowoff = (uint32_t)((nb * 16 + n16) * RS);
#pragma unroll
            for (int mb = 0; mb < 2; mb++) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = __builtin_fmaf(acc1[nb][mb][i] + acc2[nb][mb][i], acc_scale, bq[mb][i]);
                if (CONV2) v += *(const f32x4 *)(res_lds + (mb * NB + nb) * 1024);
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.0f);
                if (KEEP) *(f32x4 *)(res_lds + (mb * NB + nb) * 1024) = v;
                uint32_t h01, l01, h23, l23;
                split_pair((f32x2){v[0], v[1]}, h01, l01);
                split_pair((f32x2){v[2], v[3]}, h23, l23);
                const uint32_t co2 = (uint32_t)((wave * 32 + mb * 16 + kg * 4) * 2);
                *(u32x2 *)(lds + rowoff + co2) = (u32x2){h01, h23};
                *(u32x2 *)(lds + GEO::PART_BYTES + rowoff + co2) = (u32x2){l01, l23};
            }
        }
    };
    {   // the stem (its own code: 9 one-k-step chunks), outside the loop over the CH -> CH layers
        f32x4 acc1[NB][2], acc2[NB][2];
        __syncthreads();                                   // the prologue's LDS writes are visible
        run_layer<GEO, true>(acc1, acc2, W, lds, weights, voff, chunk, n_tower, crs, geo);
        __syncthreads();                                   // every wave has finished reading the input
        epilogue(0, acc1, acc2, std::false_type{}, std::true_type{});
    }
    for (int L = 1; L < n_layers; L++) {
        f32x4 acc1[NB][2], acc2[NB][2];
        __syncthreads();                                   // the previous layer's epilogue is visible
        run_layer<GEO, false>(acc1, acc2, W, lds, weights, voff, chunk, n_tower, crs, geo);
        __syncthreads();                                   // every wave has finished reading this layer's input
        if ((L & 1) == 0) epilogue(L, acc1, acc2, std::true_type{}, std::true_type{});      // second conv of a block: + skip, keep
        else epilogue(L, acc1, acc2, std::false_type{}, std::false_type{});
    }
    __syncthreads();
    if (out_heads) {
        // 1x1 head convs (neural_network.py:113, 118): unit hm = head * 2 + M block (16 of a head's 32 channels), dealt to the
        // waves round-robin; a wave runs its units over every column block.  head_w: [hm 4][kq NW][part 2][lane 64][8 f16].
        // (the per-lane geometry is derived again from a laundered lane id: kept alive from the prologue it would sit in
        // scratch across the whole tower)
        int lane2 = lane;
        asm volatile("" : "+v"(lane2));
        const int n16 = lane2 & 15, kg = lane2 >> 4;
        for (int hm = wave; hm < 4; hm += NW) {
            const int head = hm >> 1, mb = hm & 1;
            f16x8 wh[NW], wl[NW];
#pragma unroll
            for (int kq = 0; kq < NW; kq++) {
                const unsigned char *p = head_w + ((size_t)(hm * NW + kq) * 2) * 1024 + lane2 * 16;
                wh[kq] = __builtin_bit_cast(f16x8, *(const u32x4 *)p);
                wl[kq] = __builtin_bit_cast(f16x8, *(const u32x4 *)(p + 1024));
            }
            const float *hb = head_bias + head * 32 + mb * 16;      // uniform: scalar loads
            float sb[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                sb[k] = hb[k];
                asm volatile("" : "+s"(sb[k]));
            }
            f32x4 b;
#pragma unroll
            for (int i = 0; i < 4; i++) b[i] = kg == 0 ? sb[i] : kg == 1 ? sb[4 + i] : kg == 2 ? sb[8 + i] : sb[12 + i];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const uint32_t xb = (uint32_t)((nb * 16 + n16) * RS + kg * 16);
                f32x4 h1 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kq = 0; kq < NW; kq++) {
                    const f16x8 xh = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb + kq * 64));
                    const f16x8 xl = __builtin_bit_cast(f16x8, *(const u32x4 *)(lds + xb + GEO::PART_BYTES + kq * 64));
                    const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kq], xh, h2, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kq], xh, h1, 0, 0, 0);
                    h2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kq], xl, a, 0, 0, 0);
                }
                // features f32 [row][head][channel 32][cell] (the reference's NCHW flatten order)
                const int col = nb * 16 + n16;
                const int gb = g0 + col / cells, cell = col % cells;
                if (col < ncol && gb < n_live) {
                    float *o = out_heads + (((size_t)gb * 2 + head) * 32 + mb * 16 + kg * 4) * cells + cell;
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i * cells] = fmaxf(__builtin_fmaf(h1[i] + h2[i], head_scale, b[i]), 0.0f);
                }
            }
        }
        return;
    }
    for (int p = threadIdx.x; p < ncol * (CH / 4); p += NT) {   // activations [column][CH] f32 = (hi + lo) * 2^-ka
        const int col = p / (CH / 4), ch4 = p % (CH / 4);
        if (g0 + col / cells < n_live) {
            const u32x2 ph = *(const u32x2 *)(lds + col * RS + ch4 * 8);
            const u32x2 pl = *(const u32x2 *)(lds + GEO::PART_BYTES + col * RS + ch4 * 8);
            const f32x2 v01 = join_pair(ph.x, pl.x), v23 = join_pair(ph.y, pl.y);
            *(f32x4 *)(out + ((size_t)g0 * cells + col) * CH + ch4 * 4) = (f32x4){v01.x, v01.y, v23.x, v23.y} * out_scale;
        }
    }
}

}   // namespace tg

namespace {

struct TgArgs {
    const float *planes;
    const void *weights, *head_w;
    const float *bias, *head_bias;
    float *out, *out_heads;
    const int *rows, *n_rows;
    int G, R, C, TB, n_layers;
    float sc[4];
    int gate_lo, gate_hi;
    hipStream_t s;
};

template <int NW, int NB, int D> int launch_tg(const TgArgs &a) {
    tg::k_tower_g<NW, NB, D><<<dim3((a.G + a.TB - 1) / a.TB), dim3(64 * NW), 0, a.s>>>(
        a.planes, (const unsigned char *)a.weights, (const unsigned char *)a.head_w, a.bias, a.head_bias, a.out, a.out_heads, a.rows,
        a.n_rows, a.G, a.R, a.C, a.TB, a.n_layers, a.sc[0], a.sc[1], a.sc[2], a.sc[3], a.gate_lo, a.gate_hi);
    if (hipGetLastError() != hipSuccess) return yy_tower_set_err(YY_E_HIP, "yy_nn_tower_g: launch failed");
    return YY_OK;
}

// instantiated forms: 128 channels with 4..9 column blocks; narrower networks with 9 (and 4 for small batches)
int dispatch_tg(int nw, int nb, const TgArgs &a) {
    if (nw == 4) {
        switch (nb) {
        case 4: return launch_tg<4, 4, 9>(a);
        case 5: return launch_tg<4, 5, 9>(a);
        case 6: return launch_tg<4, 6, 9>(a);
        case 7: return launch_tg<4, 7, 9>(a);
        case 8: return launch_tg<4, 8, 9>(a);
        case 9: return launch_tg<4, 9, 9>(a);
        }
    } else if (nb == 4 || nb == 9) {
        if (nw == 1) return nb == 4 ? launch_tg<1, 4, 9>(a) : launch_tg<1, 9, 3>(a);
        if (nw == 2) return nb == 4 ? launch_tg<2, 4, 9>(a) : launch_tg<2, 9, 3>(a);
        if (nw == 3) return nb == 4 ? launch_tg<3, 4, 9>(a) : launch_tg<3, 9, 3>(a);
    }
    return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_g: no kernel form for this (channels, column blocks)");
}

}   // namespace

// Column blocks (of 16 columns) the kernel forms exist for, per channel count: yy_nn_tower_g_forms(channels, out[8]) -> count.
extern "C" int yy_nn_tower_g_forms(int channels, int *nb_out) {
    if (channels == 128) {
        for (int i = 0; i < 6; i++) nb_out[i] = 4 + i;
        return 6;
    }
    if (channels == 32 || channels == 64 || channels == 96) {
        nb_out[0] = 4;
        nb_out[1] = 9;
        return 2;
    }
    return 0;
}

// weights: f16 chunks [9 + 9*(channels/32)*(n_layers-1)][channels/32][2][2][64][8] (network.pack_tower_g) times 2^weight_exp;
// head_w f16 [4][channels/32][2][64][8] times 2^head_exp and head_bias f32 [64] (pack_heads_g), or both NULL with out_heads NULL;
// bias f32 [n_layers, channels] times 2^act_exp; planes f32 [G,5,R,C]; out f32 [G,R,C,channels] or out_heads f32 [G,2,32,R*C].
// nb = column blocks per workgroup (a form listed by yy_nn_tower_g_forms), boards = boards per workgroup (boards*R*C <= 16*nb).
// rows / n_rows (device, or both NULL): evaluate planes[rows[i]] for i < *n_rows into dense row i; the launch only runs when
// gate_lo < live rows <= gate_hi (pass -1, INT_MAX for "always").
extern "C" int yy_nn_tower_g(const float *planes, const void *weights, const void *head_w, const float *bias, const float *head_bias,
                             float *out, float *out_heads, const int32_t *rows, const int32_t *n_rows, int G, int R, int C,
                             int channels, int n_layers, int weight_exp, int head_exp, int act_exp, int nb, int boards, int gate_lo,
                             int gate_hi, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || (!out && !out_heads) || (out_heads && (!head_w || !head_bias)) || G < 0 || (rows && !n_rows))
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_g: bad argument");
    if (R < 1 || C < 1 || R * C > 144 || channels < 32 || channels > 128 || (channels & 31) || n_layers < 1 || n_layers > 21 ||
        (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED,
                                "yy_nn_tower_g: needs boards of at most 144 cells, 32/64/96/128 channels, at most 10 residual blocks");
    if (boards < 1 || nb < 1 || boards * R * C > 16 * nb) return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_g: boards * cells > 16 * nb");
    TgArgs a = {planes, weights, head_w, bias, head_bias, out, out_heads, rows, n_rows, G, R, C, boards, n_layers,
                {ldexpf(1.0f, act_exp), ldexpf(1.0f, -weight_exp), ldexpf(1.0f, -(head_exp + act_exp)), ldexpf(1.0f, -act_exp)},
                gate_lo, gate_hi, (hipStream_t)s};
    return dispatch_tg(channels / 32, nb, a);
}

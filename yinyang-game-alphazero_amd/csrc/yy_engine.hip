// yy_engine.hip -- HIP kernels + C ABI (include/yy_engine.h) of the Yin-Yang self-play hot path
// for MI355X (gfx950, wave64).  Written for CDNA4 only.
//
// Execution model
//   * tree kernels: ONE GAME PER WAVEFRONT (block = 64 threads, blockIdx.x = game).  Lane l owns
//     board cell(s) l, l+64, l+128 and child edge(s) l, l+64, ... of the node being scanned, so a
//     node's children are read with one coalesced 16-B-per-lane load, the PUCT arg-max is a
//     wave reduction, the legal mask -> child list compaction is a ballot/mbcnt prefix sum, and
//     the 5 input planes are written as coalesced rows.  Bitboards are wave-uniform (SGPRs).
//   * stateless rules kernels: ONE GAME PER LANE on the same bitboard code, boards staged through
//     LDS so the int8 [G,R,C] tensors are read/written coalesced.
//   * blockIdx -> game is fixed, and blocks are dealt round-robin over the 8 XCDs, so a game's
//     arena stays in the same XCD's L2 from launch to launch (speed only, never correctness).
//
// HBM layout (all per context, game-major):
//   edges  uint4 [G][edge_cap]   {prior f32, visits i32, value_sum f32, child(24b)|action<<24}
//   nodes  uint4 [G][node_cap]   {first_edge, k | flags<<16 | player<<24, terminal value f32, -}
//   nboard u64   [G][node_cap][2*NW]   (copied mode only) black words then white words
//   gboard u64   [G][2*NW]       root board (copied) / THE shared board (aliased)
//   path   i32   [G][path_cap]   edge indices chosen by the last selection
//   state  GameState [G]         counters, root statistics, the pending leaf record
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "../../include/yy_engine.h"
#include "yy_bitboard.h"

#define YY_VERSION 100

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int set_err(int code, const char *fmt, const char *a = "", const char *b = "") {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}
#define HIP_TRY(x)                                                                    \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) return set_err(YY_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *yy_last_error(void) { return g_err; }
extern "C" int yy_tower_set_err(int code, const char *msg) { return set_err(code, "%s%s", msg); }   // for yy_tower.hip
extern "C" int yy_version(void) { return YY_VERSION; }

static int check_geo(int G, int R, int C) {
    if (G <= 0 || R <= 0 || C <= 0) return set_err(YY_E_INVALID, "non-positive size%s%s");
    if (R > YY_MAX_DIM || C > YY_MAX_DIM || R * C > 64 * YY_MAX_NW)
        return set_err(YY_E_UNSUPPORTED, "board larger than 16x16 / 192 cells%s%s");
    return YY_OK;
}

// ------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rfl(int v) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v);
}
__device__ __forceinline__ float rflf(float v) { return __uint_as_float(rfl(__float_as_uint(v))); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// number of set bits of m below this lane's position
__device__ __forceinline__ int mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave64 reductions on the VALU with DPP (row_shr 1/2/4/8 scan inside each 16-lane row, then row_bcast15 /
// row_bcast31 across rows; lane 63 ends up with the total): ~6 dependent VALU ops instead of 6 LDS-crossbar
// shuffles.  Used on the latency-critical PUCT descent.
#define YY_DPP_STEP(OP, ID, v, ctrl, rmask) \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t uadd32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    YY_DPP_STEP(umax32, 0u, v, 0x111, 0xf);
    YY_DPP_STEP(umax32, 0u, v, 0x112, 0xf);
    YY_DPP_STEP(umax32, 0u, v, 0x114, 0xf);
    YY_DPP_STEP(umax32, 0u, v, 0x118, 0xf);
    YY_DPP_STEP(umax32, 0u, v, 0x142, 0xa);
    YY_DPP_STEP(umax32, 0u, v, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x111, 0xf);
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x112, 0xf);
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x114, 0xf);
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x118, 0xf);
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x142, 0xa);
    YY_DPP_STEP(umin32, 0xFFFFFFFFu, v, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_uadd(uint32_t v) {
    YY_DPP_STEP(uadd32, 0u, v, 0x111, 0xf);
    YY_DPP_STEP(uadd32, 0u, v, 0x112, 0xf);
    YY_DPP_STEP(uadd32, 0u, v, 0x114, 0xf);
    YY_DPP_STEP(uadd32, 0u, v, 0x118, 0xf);
    YY_DPP_STEP(uadd32, 0u, v, 0x142, 0xa);
    YY_DPP_STEP(uadd32, 0u, v, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// float -> unsigned key with the same order (all non-NaN keys are > 0)
__device__ __forceinline__ uint32_t f32_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int NW> __device__ __forceinline__ BB<NW> bb_uniform_load(const uint64_t *p) {
    BB<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = rfl64(p[i]);
    return r;
}
template <int NW> __device__ __forceinline__ void bb_store_lane0(uint64_t *p, BB<NW> b) {
    if (lane_id() == 0) {
#pragma unroll
        for (int i = 0; i < NW; i++) p[i] = b.w[i];
    }
}

// wave-cooperative: int8 board (cell per lane) -> uniform bitboards via ballot
template <int NW>
__device__ __forceinline__ void board_to_bb(const int8_t *b, int A, BB<NW> &black, BB<NW> &white) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
        int cell = j * 64 + lane_id();
        int v = (cell < A) ? (int)b[cell] : 0;
        black.w[j] = __ballot(v == 1);
        white.w[j] = __ballot(v == -1);
    }
}
template <int NW>
__device__ __forceinline__ void bb_to_board(int8_t *b, int A, BB<NW> black, BB<NW> white) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
        int cell = j * 64 + lane_id();
        if (cell < A)
            b[cell] = ((black.w[j] >> lane_id()) & 1) ? 1 : (((white.w[j] >> lane_id()) & 1) ? -1 : 0);
    }
}

// row r of `occ` as the low C bits (C <= 16; may straddle a word boundary)
template <int NW> __device__ __forceinline__ uint32_t bb_row_bits(BB<NW> occ, int r, int C) {
    int s = r * C, w = s >> 6, o = s & 63;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (i == w) lo = occ.w[i];
        if (i == w + 1) hi = occ.w[i];
    }
    uint64_t v = lo >> o;
    if (o) v |= hi << (64 - o);
    return (uint32_t)v & ((1u << C) - 1u);
}

// wave-cooperative board_to_input (neural_network.py:156-196): lane = cell, coalesced rows
template <int NW>
__device__ __forceinline__ void write_planes(float *out, const YYGeo &geo, BB<NW> black, BB<NW> white) {
    const int A = geo.A, C = geo.C;
    BB<NW> occ = black | white;
    BB<NW> col0 = bb_load<NW>(geo.col0);
#pragma unroll
    for (int j = 0; j < NW; j++) {
        int cell = j * 64 + lane_id();
        if (cell < A) {
            int r = cell / C, c = cell - r * C;
            bool bk = (black.w[j] >> lane_id()) & 1, wh = (white.w[j] >> lane_id()) & 1;
            int rowk = __popc(bb_row_bits(occ, r, C));
            int colk = bb_popc(occ & bb_shl(col0, c));
            out[cell] = (!bk && !wh) ? 1.0f : 0.0f;
            out[A + cell] = bk ? 1.0f : 0.0f;
            out[2 * A + cell] = wh ? 1.0f : 0.0f;
            out[3 * A + cell] = geo.rowfill[rowk];
            out[4 * A + cell] = geo.colfill[colk];
        }
    }
}

// ------------------------------------------------------------------------- stateless kernels
// One game per lane; the block's boards (64 games x A bytes, contiguous in HBM) are staged through
// LDS so global traffic is coalesced.
#define RULES_BLOCK 64

template <int NW>
__device__ __forceinline__ void lds_board_to_bb(const int8_t *lb, int A, BB<NW> &black, BB<NW> &white) {
    black = bb_zero<NW>();
    white = bb_zero<NW>();
    for (int a = 0; a < A; a++) {
        int v = lb[a];
        if (v == 1) black.w[a >> 6] |= 1ull << (a & 63);
        if (v == -1) white.w[a >> 6] |= 1ull << (a & 63);
    }
}

// stage this block's boards global -> LDS (coalesced bytes)
__device__ __forceinline__ void stage_in(int8_t *lds, const int8_t *boards, int g0, int ng, int A) {
    const int n = ng * A;
    const int8_t *src = boards + (size_t)g0 * A;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
}

template <int NW>
__global__ void __launch_bounds__(RULES_BLOCK) k_valid_mask(const int8_t *boards, const int8_t *players,
                                                            int G, YYGeo geo, uint8_t *out) {
    __shared__ int8_t lds[RULES_BLOCK * 64 * YY_MAX_NW];
    const int A = geo.A, g0 = blockIdx.x * RULES_BLOCK;
    const int ng = min(RULES_BLOCK, G - g0);
    stage_in(lds, boards, g0, ng, A);
    const int g = g0 + threadIdx.x;
    BB<NW> mask = bb_zero<NW>();
    if (threadIdx.x < ng) {
        GeoBB<NW> gb = geo_bb<NW>(geo);
        BB<NW> black, white;
        lds_board_to_bb<NW>(lds + threadIdx.x * A, A, black, white);
        bool pre = bb_pre2x2(black, white, gb);
        mask = (players[g] == 1) ? bb_legal(black, white, pre, geo, gb) : bb_legal(white, black, pre, geo, gb);
    }
    __syncthreads();
    if (threadIdx.x < ng)
        for (int a = 0; a < A; a++) lds[threadIdx.x * A + a] = (int8_t)bb_test(mask, a);
    __syncthreads();
    uint8_t *dst = out + (size_t)g0 * A;
    for (int i = threadIdx.x; i < ng * A; i += blockDim.x) dst[i] = (uint8_t)lds[i];
}

template <int NW>
__global__ void __launch_bounds__(RULES_BLOCK) k_step(int8_t *boards, int8_t *players, const int32_t *actions,
                                                      int G, YYGeo geo, uint8_t *placed) {
    const int g = blockIdx.x * RULES_BLOCK + threadIdx.x;
    if (g >= G) return;
    const int A = geo.A;
    GeoBB<NW> gb = geo_bb<NW>(geo);
    BB<NW> black = bb_zero<NW>(), white = bb_zero<NW>();
    int8_t *b = boards + (size_t)g * A;
    for (int a = 0; a < A; a++) {
        int v = b[a];
        if (v == 1) black.w[a >> 6] |= 1ull << (a & 63);
        if (v == -1) white.w[a >> 6] |= 1ull << (a & 63);
    }
    const int p = players[g], a = actions[g];
    bool ok = false;
    if (a >= 0 && a < A) {  // yin_yang_logic.py:34 bounds check
        bool pre = bb_pre2x2(black, white, gb);
        BB<NW> m = (p == 1) ? bb_legal(black, white, pre, geo, gb) : bb_legal(white, black, pre, geo, gb);
        ok = bb_test(m, a);
    }
    if (ok) b[a] = (p == 1) ? 1 : -1;  // yin_yang_game.py:55-56
    players[g] = (int8_t)-p;           // yin_yang_game.py:58, regardless
    if (placed) placed[g] = ok;
}

template <int NW>
__global__ void __launch_bounds__(RULES_BLOCK) k_game_ended(const int8_t *boards, const int8_t *players, int G,
                                                            YYGeo geo, double *out, int32_t *counts) {
    __shared__ int8_t lds[RULES_BLOCK * 64 * YY_MAX_NW];
    const int A = geo.A, g0 = blockIdx.x * RULES_BLOCK;
    const int ng = min(RULES_BLOCK, G - g0);
    stage_in(lds, boards, g0, ng, A);
    if (threadIdx.x >= ng) return;
    const int g = g0 + threadIdx.x;
    GeoBB<NW> gb = geo_bb<NW>(geo);
    BB<NW> black, white;
    lds_board_to_bb<NW>(lds + threadIdx.x * A, A, black, white);
    bool pre = bb_pre2x2(black, white, gb);
    BB<NW> mb = bb_legal(black, white, pre, geo, gb), mw = bb_legal(white, black, pre, geo, gb);
    int res = bb_result_black(black, white, mb, mw);
    double v = 0.0;
    if (res == 2) v = 0.0001;                                   // yin_yang_game.py:107
    else if (res != 0) v = (players[g] == 1) ? (double)res : (double)-res;
    out[g] = v;
    if (counts) {
        counts[2 * g] = bb_popc(black);
        counts[2 * g + 1] = bb_popc(white);
    }
}

template <int NW>
__global__ void __launch_bounds__(64) k_encode(const int8_t *boards, int G, YYGeo geo, float *out) {
    const int g = blockIdx.x;
    BB<NW> black, white;
    board_to_bb<NW>(boards + (size_t)g * geo.A, geo.A, black, white);
    write_planes<NW>(out + (size_t)g * 5 * geo.A, geo, black, white);
}

template <int NW>
__global__ void __launch_bounds__(64) k_pack(const int8_t *boards, int G, YYGeo geo, uint64_t *black, uint64_t *white) {
    const int g = blockIdx.x;
    BB<NW> b, w;
    board_to_bb<NW>(boards + (size_t)g * geo.A, geo.A, b, w);
    if (lane_id() == 0) {
#pragma unroll
        for (int j = 0; j < NW; j++) {
            black[(size_t)j * G + g] = b.w[j];
            white[(size_t)j * G + g] = w.w[j];
        }
    }
}

template <int NW>
__global__ void __launch_bounds__(64) k_unpack(const uint64_t *black, const uint64_t *white, int G, YYGeo geo,
                                               int8_t *boards) {
    const int g = blockIdx.x;
    BB<NW> b, w;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        b.w[j] = rfl64(black[(size_t)j * G + g]);
        w.w[j] = rfl64(white[(size_t)j * G + g]);
    }
    bb_to_board<NW>(boards + (size_t)g * geo.A, geo.A, b, w);
}

// packed rules: one game per lane, SoA word-major bitboards -> fully coalesced 8-B loads/stores
template <int NW>
__global__ void __launch_bounds__(256) k_mask_terminal_bb(const uint64_t *black, const uint64_t *white, int G,
                                                          YYGeo geo, uint64_t *m1, uint64_t *m2, int8_t *result) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        GeoBB<NW> gb = geo_bb<NW>(geo);
        BB<NW> b, w;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            b.w[j] = black[(size_t)j * G + g];
            w.w[j] = white[(size_t)j * G + g];
        }
        bool pre = bb_pre2x2(b, w, gb);
        BB<NW> mb = bb_legal(b, w, pre, geo, gb), mw = bb_legal(w, b, pre, geo, gb);
#pragma unroll
        for (int j = 0; j < NW; j++) {
            if (m1) m1[(size_t)j * G + g] = mb.w[j];
            if (m2) m2[(size_t)j * G + g] = mw.w[j];
        }
        if (result) result[g] = (int8_t)bb_result_black(b, w, mb, mw);
    }
}

#define DISPATCH_NW(NWv, ...)                                  \
    switch (NWv) {                                             \
        case 1: { constexpr int NW = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int NW = 2; __VA_ARGS__; } break;  \
        default: { constexpr int NW = 3; __VA_ARGS__; } break; \
    }

extern "C" int yy_rules_valid_mask(const int8_t *boards, const int8_t *players, int G, int R, int C, uint32_t flags,
                                   uint8_t *out, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !players || !out) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, flags);
    dim3 grid((G + RULES_BLOCK - 1) / RULES_BLOCK);
    DISPATCH_NW(geo.NW, k_valid_mask<NW><<<grid, dim3(RULES_BLOCK), 0, (hipStream_t)s>>>(boards,
                                            players, G, geo, out));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_rules_step(int8_t *boards, int8_t *players, const int32_t *actions, int G, int R, int C,
                             uint32_t flags, uint8_t *placed, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !players || !actions) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, flags);
    dim3 grid((G + RULES_BLOCK - 1) / RULES_BLOCK);
    DISPATCH_NW(geo.NW, k_step<NW><<<grid, dim3(RULES_BLOCK), 0, (hipStream_t)s>>>(boards, players,
                                            actions, G, geo, placed));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_rules_game_ended(const int8_t *boards, const int8_t *players, int G, int R, int C, uint32_t flags,
                                   double *out, int32_t *counts, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !players || !out) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, flags);
    dim3 grid((G + RULES_BLOCK - 1) / RULES_BLOCK);
    DISPATCH_NW(geo.NW, k_game_ended<NW><<<grid, dim3(RULES_BLOCK), 0, (hipStream_t)s>>>(boards,
                                            players, G, geo, out, counts));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_encode_planes(const int8_t *boards, int G, int R, int C, float *out, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !out) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, 0);
    DISPATCH_NW(geo.NW, k_encode<NW><<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(boards, G, geo, out));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_pack_boards(const int8_t *boards, int G, int R, int C, uint64_t *black, uint64_t *white,
                              yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !black || !white) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, 0);
    DISPATCH_NW(geo.NW,
                k_pack<NW><<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(boards, G, geo, black, white));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_unpack_boards(const uint64_t *black, const uint64_t *white, int G, int R, int C, int8_t *boards,
                                yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!boards || !black || !white) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, 0);
    DISPATCH_NW(geo.NW,
                k_unpack<NW><<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(black, white, G, geo, boards));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_rules_mask_terminal_bb(const uint64_t *black, const uint64_t *white, int G, int R, int C,
                                         uint32_t flags, uint64_t *m1, uint64_t *m2, int8_t *result, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (int e = check_geo(G, R, C)) return e;
    if (!black || !white) return set_err(YY_E_INVALID, "null pointer%s%s");
    YYGeo geo;
    yy_make_geo(&geo, R, C, flags);
    int blocks = (G + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    DISPATCH_NW(geo.NW, k_mask_terminal_bb<NW><<<dim3(blocks), dim3(256), 0, (hipStream_t)s>>>(black,
                                            white, G, geo, m1, m2, result));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

// =============================================================================== MCTS context
enum : uint8_t { K_NONE = 0, K_TERMINAL = 1, K_EXPAND = 2, K_REEXPAND = 3, K_ROOTPASS = 4, K_ROOTINIT = 5, K_REUSE = 6 };
#define CHILD_NONE 0x00FFFFFFu
#define NF_TERMINAL 1u
#define NF_HASVALUE 2u   // childless node whose evaluator value (record .w) may be reused (YY_FLAG_REUSE_PASS_VALUE)
#define TT_PROBES 8

// slot hash of a position: murmur3's 64-bit finalizer after every word -- positions of one search differ in a few bits, and a
// single multiply leaves the low bits (the slot index) clustered (measured: probe runs of 8 at 5 % load)
template <int NW> __device__ __forceinline__ uint64_t bb_hash(const uint64_t *black, const uint64_t *white) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < 2 * NW; i++) {
        h ^= (i < NW) ? black[i] : white[i - NW];
        h ^= h >> 33;
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
        h *= 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 33;
    }
    return h;
}

struct GameState {
    int32_t n_nodes, n_edges;
    int32_t root_N;
    float root_W;
    double root_W_py;       // root.value_sum while it is still a python float (terminal root only)
    int32_t path_len;       // edges on the selected path == depth of the leaf
    int32_t leaf_node;      // node record of the leaf when it already has one, else -1
    float leaf_tv;          // terminal value of a freshly evaluated leaf (f32 of 1 / -1 / 1e-4)
    int8_t root_player;
    uint8_t active, err, root_w_is_py;
    uint8_t leaf_kind;
    int8_t leaf_player;
    uint8_t leaf_terminal;
    uint8_t err_ever;       // sticky: set with err, survives yy_mcts_begin, cleared only by yy_mcts_status
    int32_t leaf_src;       // >= 0: slot of the evaluation cache that holds this leaf's position (no evaluator row), else -1
    int32_t leaf_ec_slot;   // leaf_src < 0: the cache slot this leaf's evaluation goes into
    uint32_t ec_epoch;      // entries of other epochs are replaceable (see k_begin)
    int32_t root_stones;    // stones on the root board: a cached position with no more stones cannot be a leaf again
    uint64_t leaf_board[2 * YY_MAX_NW];
    uint64_t leaf_mask[YY_MAX_NW];
    uint64_t ctr[8];        // evals, levels, children scanned, children created, terminal revisits, nodes, reused pass values, position-table hits
};

struct yy_mcts {
    yy_mcts_config cfg;
    YYGeo geo;
    int64_t node_cap, edge_cap, path_cap;
    uint4 *edges, *nodes;
    uint64_t *nboard, *gboard;
    int32_t *path;
    // evaluation cache (YY_FLAG_REUSE_TRANSPOSITIONS / YY_FLAG_KEEP_EVALUATIONS): per game ec_cap slots, open addressing
    uint32_t *ec_meta;      // [G, ec_cap]  epoch << 8 | stones, 0 = never used
    uint64_t *ec_key;       // [G, ec_cap, 2*NW]  the position
    float *ec_val;          // [G, ec_cap]  the evaluator's value
    float *ec_pol;          // [G, ec_cap, A]  the evaluator's policy row
    int64_t ec_cap;
    // shared book of pre-evaluated positions (yy_mcts_set_book; arrays owned by the caller, read-only here)
    const uint32_t *bk_meta;
    const uint64_t *bk_key;
    const float *bk_val, *bk_pol;
    int64_t bk_cap;
    int bk_stones;
    GameState *state;
    float *sqrt_tab;
    uint64_t *scratch;      // [8] counters + overflow count
    uint64_t bytes;
    int pending;            // 1 = a select is pending an expand_backup
};

struct MctsDev {  // by-value kernel argument
    YYGeo geo;
    int32_t G;
    uint32_t aliased, reuse;
    float cpuct;
    double eps;
    int64_t node_cap, edge_cap, path_cap;
    uint4 *edges, *nodes;
    uint64_t *nboard, *gboard;
    int32_t *path;
    uint32_t *ec_meta;
    uint64_t *ec_key;
    float *ec_val, *ec_pol;
    int32_t ec_cap;
    uint32_t ec_keep;
    const uint32_t *bk_meta;
    const uint64_t *bk_key;
    const float *bk_val, *bk_pol;
    uint32_t bk_mask;
    int32_t bk_stones;
    GameState *state;
    const float *sqrt_tab;
    int32_t sqrt_n;
};

static MctsDev make_dev(const yy_mcts *c) {
    MctsDev d;
    d.geo = c->geo;
    d.G = c->cfg.G;
    d.aliased = (c->cfg.flags & YY_FLAG_ALIASED) ? 1u : 0u;
    d.reuse = (c->cfg.flags & YY_FLAG_REUSE_PASS_VALUE) ? 1u : 0u;
    d.cpuct = c->cfg.cpuct;
    d.eps = 0.0;
    d.node_cap = c->node_cap;
    d.edge_cap = c->edge_cap;
    d.path_cap = c->path_cap;
    d.edges = c->edges;
    d.nodes = c->nodes;
    d.nboard = c->nboard;
    d.gboard = c->gboard;
    d.path = c->path;
    d.ec_meta = c->ec_meta;
    d.ec_key = c->ec_key;
    d.ec_val = c->ec_val;
    d.ec_pol = c->ec_pol;
    d.ec_cap = (int32_t)c->ec_cap;
    d.ec_keep = (c->cfg.flags & YY_FLAG_KEEP_EVALUATIONS) ? 1u : 0u;
    d.bk_meta = c->bk_meta;
    d.bk_key = c->bk_key;
    d.bk_val = c->bk_val;
    d.bk_pol = c->bk_pol;
    d.bk_mask = (uint32_t)(c->bk_cap - 1);
    d.bk_stones = c->bk_stones;
    d.state = c->state;
    d.sqrt_tab = c->sqrt_tab;
    d.sqrt_n = c->cfg.max_sims + 2;
    return d;
}

__device__ __forceinline__ uint32_t node_pack(int k, uint32_t flags, int player) {
    return (uint32_t)k | (flags << 16) | ((uint32_t)(player & 0xFF) << 24);
}
__device__ __forceinline__ int node_k(uint32_t y) { return (int)(y & 0xFFFFu); }
__device__ __forceinline__ uint32_t node_flags(uint32_t y) { return (y >> 16) & 0xFFu; }
__device__ __forceinline__ int node_player(uint32_t y) { return (int)(int8_t)(y >> 24); }

// getGameEnded(board, player) as float32 + the mask of `player` (yin_yang_game.py:80-110)
template <int NW>
__device__ __forceinline__ void leaf_rules(const YYGeo &geo, const GeoBB<NW> &gb, BB<NW> black, BB<NW> white,
                                           int player, BB<NW> &mask, bool &terminal, float &tv) {
    bool pre = bb_pre2x2(black, white, gb);
    BB<NW> mb = bb_legal(black, white, pre, geo, gb), mw = bb_legal(white, black, pre, geo, gb);
    int res = bb_result_black(black, white, mb, mw);
    mask = (player == 1) ? mb : mw;
    terminal = (res != 0);
    tv = (res == 2) ? 0.0001f : ((player == 1) ? (float)res : (float)-res);
}

// ---- root prologue: mcts.py:288-295
template <int NW> __global__ void __launch_bounds__(64) k_begin(MctsDev d, const int8_t *boards,
                                                                const int8_t *players, const uint8_t *active,
                                                                float *planes) {
    const int g = blockIdx.x;
    GameState *st = d.state + g;
    const bool act = active ? (active[g] != 0) : true;
    BB<NW> black, white;
    board_to_bb<NW>(boards + (size_t)g * d.geo.A, d.geo.A, black, white);
    if (lane_id() == 0) {
        st->n_nodes = 1;  // node 0 = root
        st->n_edges = 0;
        st->root_N = 0;
        st->root_W = 0.0f;
        st->root_W_py = 0.0;
        st->root_w_is_py = 1;
        st->path_len = 0;
        st->leaf_node = 0;
        st->root_player = players[g];
        st->active = act;
        st->err = 0;
        st->leaf_kind = act ? K_ROOTINIT : K_NONE;
        d.nodes[(size_t)g * d.node_cap] = make_uint4(0u, node_pack(0, 0, players[g]), 0u, 0u);
    }
    if (d.ec_meta && lane_id() == 0) {
        // Entries of an older epoch are the first to be replaced.  Without YY_FLAG_KEEP_EVALUATIONS every search is its own
        // epoch and only entries of the current one are looked at (reuse inside one search); with it the epoch changes when
        // a game starts over from the empty board, so a game's searches share their evaluations.
        int stones = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) stones += yy_popc64(black.w[i]) + yy_popc64(white.w[i]);
        uint32_t ep = st->ec_epoch;
        if (!d.ec_keep || stones == 0 || ep == 0u) ep = (ep + 1u) & 0xFFFFFFu;
        if (ep == 0u) ep = 1u;
        st->ec_epoch = ep;
        st->root_stones = stones;
    }
    uint64_t *gbd = d.gboard + (size_t)g * 2 * NW;
    bb_store_lane0<NW>(gbd, black);
    bb_store_lane0<NW>(gbd + NW, white);
    if (!d.aliased) {
        uint64_t *nb = d.nboard + (size_t)g * d.node_cap * 2 * NW;
        bb_store_lane0<NW>(nb, black);
        bb_store_lane0<NW>(nb + NW, white);
    }
    if (act) {
        // root.expand's rules (mcts.py:63-71) for the pending K_ROOTINIT expansion
        const GeoBB<NW> gb = geo_bb<NW>(d.geo);
        const int rp = rfl((int)players[g]);
        BB<NW> mask;
        bool term;
        float tv;
        leaf_rules<NW>(d.geo, gb, black, white, rp, mask, term, tv);
        if (lane_id() == 0) {
#pragma unroll
            for (int i = 0; i < NW; i++) {
                st->leaf_board[i] = black.w[i];
                st->leaf_board[NW + i] = white.w[i];
                st->leaf_mask[i] = mask.w[i];
            }
            st->leaf_player = (int8_t)rp;
            st->leaf_terminal = term;
            st->leaf_tv = tv;
        }
        write_planes<NW>(planes + (size_t)g * 5 * d.geo.A, d.geo, black, white);
    }
}

// ---- selection: mcts.py:356-362 (descent), 385-391 (state of the leaf), 63-71 (its rules)
template <int NW>
__device__ __forceinline__ void do_select(const MctsDev &d, const int g, float *planes, uint8_t *needs_eval) {
    GameState *st = d.state + g;
    const int lane = lane_id();
    if (!rfl((int)st->active) || rfl((int)st->err)) {
        if (lane == 0) {
            st->leaf_kind = K_NONE;
            if (needs_eval) needs_eval[g] = 0;
        }
        return;
    }
    const GeoBB<NW> gb = geo_bb<NW>(d.geo);
    const uint4 *nodes = d.nodes + (size_t)g * d.node_cap;
    uint4 *edges = d.edges + (size_t)g * d.edge_cap;
    int32_t *path = d.path + (size_t)g * d.path_cap;

    int node = 0, depth = 0, parent = -1, action = -1, kind;
    int s_carry = rfl(st->root_N);
    uint64_t c_levels = 0, c_scan = 0;
    for (;;) {
        const uint4 hdr = nodes[node];
        const uint32_t hy = rfl(hdr.y);
        const int k = node_k(hy), first = rfl((int)hdr.x);
        if (node_flags(hy) & NF_TERMINAL) { kind = K_TERMINAL; break; }            // mcts.py:360, 365
        if (k == 0) {                                                              // mcts.py:93-95
            // A node without children is evaluated again on every visit.  With copied boards its board never changes,
            // so that evaluation returns the value it returned the first time: YY_FLAG_REUSE_PASS_VALUE takes it from
            // the node record instead of spending an evaluator row on it (same statistics, fewer rows).
            kind = (d.reuse && (node_flags(hy) & NF_HASVALUE)) ? K_REUSE : (node == 0) ? K_ROOTPASS : K_REEXPAND;
            break;
        }
        if (depth >= (int)d.path_cap) { kind = K_NONE; if (lane == 0) st->err = st->err_ever = 1; break; }
        // ---- Node.select_child (mcts.py:97-145), float32 order of SURVEY 8a/a12
        // sum of child visits (mcts.py:112).  Copied boards: every visit of an expanded node after its
        // first descends into exactly one child, so the sum is N(node)-1 (root: completed simulations) and
        // is carried down the descent; aliased boards can give a pass node children later, so sum there.
        int S;
        if (d.aliased) {
            uint32_t part = 0;
            for (int j = lane; j < k; j += 64) part += edges[first + j].y;
            S = (int)wave_uadd(part);
        } else {
            S = s_carry;
        }
        const float sq = d.sqrt_tab[min(S, d.sqrt_n - 1)];   // f32(math.sqrt(sum_visits))
        uint32_t bk = 0, bw = 0, bn = 0;
        int bi = 0x7FFFFFFF;
        for (int j = lane; j < k; j += 64) {
            const uint4 e = edges[first + j];
            const float P = __uint_as_float(e.x), W = __uint_as_float(e.z);
            const int N = (int)e.y;
            const float t1 = __fmul_rn(d.cpuct, P);
            const float t2 = __fmul_rn(t1, sq);
            const float u = __fdiv_rn(t2, (float)(1 + N));
            const float q = (N > 0) ? __fdiv_rn(W, (float)N) : 0.0f;
            const float ucb = __fadd_rn(__fadd_rn(q, u), 0.0f);   // + 0.0f: -0.0 compares equal to +0.0 (mcts.py:133)
            const uint32_t key = f32_key(ucb);
            if (key > bk) { bk = key; bi = j; bw = e.w; bn = e.y; }               // strict >: lowest j of a lane
        }
        const uint32_t mx = wave_umax(bk);
        const bool tied = (bk == mx) && (bi != 0x7FFFFFFF);
        int best;
        if (k <= 64) {   // lane == child index: lowest tied lane = lowest action (mcts.py:133)
            const uint64_t tm = __ballot(tied);
            best = tm ? (int)__ffsll((unsigned long long)tm) - 1 : 0x7FFFFFFF;
        } else {
            const uint32_t mi = wave_umin(tied ? (uint32_t)bi : 0xFFFFFFFFu);
            best = (mx == 0u || mi == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)mi;
        }
        if (mx == 0u) best = 0x7FFFFFFF;
        if (best == 0x7FFFFFFF) { kind = K_NONE; if (lane == 0) st->err = st->err_ever = 1; break; }  // no selectable child
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)bw, best & 63);
        s_carry = (int)__builtin_amdgcn_readlane((int)bn, best & 63) - 1;
        if (lane == 0) path[depth] = first + best;
        depth++;
        c_levels++;
        c_scan += (uint64_t)k;
        parent = node;
        action = (int)(w >> 24);
        const uint32_t child = w & CHILD_NONE;
        if (child == CHILD_NONE) { kind = K_EXPAND; node = -1; break; }
        node = (int)child;
    }

    bool need = false;
    if (kind == K_EXPAND || kind == K_REEXPAND || kind == K_ROOTPASS) {
        BB<NW> black, white;
        int lplayer;
        if (kind == K_ROOTPASS) {                                                   // mcts.py:371-381
            const uint64_t *src = d.aliased ? d.gboard + (size_t)g * 2 * NW : d.nboard + (size_t)g * d.node_cap * 2 * NW;
            black = bb_uniform_load<NW>(src);
            white = bb_uniform_load<NW>(src + NW);
            lplayer = rfl((int)st->root_player);
        } else {                                                                    // mcts.py:385-391
            const uint64_t *src = d.aliased ? d.gboard + (size_t)g * 2 * NW
                                            : d.nboard + ((size_t)g * d.node_cap + parent) * 2 * NW;
            black = bb_uniform_load<NW>(src);
            white = bb_uniform_load<NW>(src + NW);
            const int pplayer = node_player(rfl(nodes[parent].y));
            // getNextState: place iff legal (yin_yang_game.py:52-58).  In copied mode the action is
            // one of the parent's legal moves on this very board, so it always places; in aliased
            // mode the shared board has moved on and the full legality test is required.
            bool ok = true;
            if (d.aliased) {
                bool pre = bb_pre2x2(black, white, gb);
                BB<NW> m = (pplayer == 1) ? bb_legal(black, white, pre, d.geo, gb) : bb_legal(white, black, pre, d.geo, gb);
                ok = bb_test(m, action);
            }
            if (ok) {
                if (pplayer == 1) black = black | bb_bit<NW>(action);
                else white = white | bb_bit<NW>(action);
                if (d.aliased) {
                    uint64_t *dst = d.gboard + (size_t)g * 2 * NW;
                    bb_store_lane0<NW>(dst, black);
                    bb_store_lane0<NW>(dst + NW, white);
                }
            }
            lplayer = -pplayer;
        }
        BB<NW> mask;
        bool term;
        float tv;
        leaf_rules<NW>(d.geo, gb, black, white, lplayer, mask, term, tv);
        // Evaluation cache: the evaluator is a function of the position alone (the planes encode the board, not the side to
        // move), so a position it has already seen -- another move order inside this search, a pass node visited again, or,
        // with YY_FLAG_KEEP_EVALUATIONS, an earlier search of the same game -- needs no evaluator row: the cached policy row
        // and value ARE this leaf's evaluation.  Keys are compared in full; the hash only picks the probe sequence.
        int src = -1, slot = -1;
        uint64_t h = 0;
        if (d.ec_meta || d.bk_meta) h = bb_hash<NW>(black.w, white.w);
        if (d.bk_meta) {
            // the shared book (yy_mcts_set_book): positions of the first plies evaluated once for ALL games before play; read-only
            int stones = 0;
#pragma unroll
            for (int i = 0; i < NW; i++) stones += yy_popc64(black.w[i]) + yy_popc64(white.w[i]);
            if (stones <= d.bk_stones) {
                uint32_t at = (uint32_t)h & d.bk_mask;
                for (int pr = 0; pr < TT_PROBES; pr++, at = (at + 1u) & d.bk_mask) {
                    if (rfl((int)d.bk_meta[at]) == 0) break;
                    const BB<NW> ob = bb_uniform_load<NW>(d.bk_key + (size_t)at * 2 * NW),
                                 ow = bb_uniform_load<NW>(d.bk_key + (size_t)at * 2 * NW + NW);
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < NW; i++) same = same && ob.w[i] == black.w[i] && ow.w[i] == white.w[i];
                    if (same) { src = -2 - (int)at; break; }
                }
            }
        }
        if (d.ec_meta && src == -1) {
            const uint32_t *meta = d.ec_meta + (size_t)g * d.ec_cap;
            const uint64_t *keys = d.ec_key + (size_t)g * d.ec_cap * 2 * NW;
            const uint32_t mask_c = (uint32_t)d.ec_cap - 1u;
            const uint32_t ep = (uint32_t)rfl((int)st->ec_epoch);
            const int rstones = rfl(st->root_stones);
            uint32_t at = (uint32_t)h & mask_c;
            const uint32_t at0 = at;
            for (int pr = 0; pr < TT_PROBES; pr++, at = (at + 1u) & mask_c) {
                const uint32_t m = (uint32_t)rfl((int)meta[at]);
                if (m == 0u) { if (slot < 0) slot = (int)at; break; }        // never used: the position is not in the table
                const bool cur = (m >> 8) == ep;
                if (cur || d.ec_keep) {
                    const BB<NW> ob = bb_uniform_load<NW>(keys + (size_t)at * 2 * NW),
                                 ow = bb_uniform_load<NW>(keys + (size_t)at * 2 * NW + NW);
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < NW; i++) same = same && ob.w[i] == black.w[i] && ow.w[i] == white.w[i];
                    if (same) { src = (int)at; break; }
                }
                // replaceable: another epoch, or a position that cannot come back (a leaf has more stones than the root)
                if (slot < 0 && (!cur || (int)(m & 0xFFu) <= rstones)) slot = (int)at;
            }
            if (slot < 0) slot = (int)at0;                                     // every probed entry is live: replace the first
        }
        if (src == -1) write_planes<NW>(planes + (size_t)g * 5 * d.geo.A, d.geo, black, white);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NW; i++) {
                st->leaf_board[i] = black.w[i];
                st->leaf_board[NW + i] = white.w[i];
                st->leaf_mask[i] = mask.w[i];
            }
            st->leaf_player = (int8_t)lplayer;
            st->leaf_terminal = term;
            st->leaf_tv = tv;
            st->leaf_src = src;
            st->leaf_ec_slot = slot;
        }
        need = src == -1;
        if (!need && lane == 0) st->ctr[7] += 1;
    }
    if (lane == 0) {
        st->leaf_kind = (uint8_t)kind;
        st->leaf_node = node;
        st->path_len = depth;
        st->ctr[1] += c_levels;
        st->ctr[2] += c_scan;
        if (need) st->ctr[0] += 1;
        if (kind == K_TERMINAL) st->ctr[4] += 1;
        if (kind == K_REUSE) st->ctr[6] += 1;
        if (needs_eval) needs_eval[g] = need;
    }
}

// ---- expansion + backup: mcts.py:50-91 (expand), 147-156 + 406-412 (update along the path)
template <int NW>
__device__ __forceinline__ void do_expand_backup(const MctsDev &d, const int g, const float *policy,
                                                 const float *value, const double *noise) {
    GameState *st = d.state + g;
    const int lane = lane_id();
    const int kind = rfl((int)st->leaf_kind);
    if (kind == K_NONE) return;
    uint4 *nodes = d.nodes + (size_t)g * d.node_cap;
    uint4 *edges = d.edges + (size_t)g * d.edge_cap;
    const int32_t *path = d.path + (size_t)g * d.path_cap;
    const int depth = rfl(st->path_len);
    int node = rfl(st->leaf_node);
    float v;
    bool v_is_py = false;   // value is a python number (terminal value), not np.float32
    if (kind == K_TERMINAL) {
        v = rflf(__uint_as_float(nodes[node].z));                                   // mcts.py:366
        v_is_py = true;
    } else if (kind == K_REUSE) {
        v = rflf(__uint_as_float(nodes[node].w));                                   // the np.float32 the evaluator returned for this node
    } else {
        const int src = ((d.ec_meta || d.bk_meta) && kind != K_ROOTINIT) ? rfl(st->leaf_src) : -1;
        const bool copy = src != -1;             // evaluation taken from the game's cache (slot src >= 0) or the book (slot -2 - src)
        const float *cpol = src >= 0 ? d.ec_pol + ((size_t)g * d.ec_cap + src) * d.geo.A
                                     : copy ? d.bk_pol + (size_t)(-2 - src) * d.geo.A : nullptr;
        v = (kind == K_ROOTINIT) ? 0.0f
            : src >= 0 ? rflf(d.ec_val[(size_t)g * d.ec_cap + src]) : copy ? rflf(d.bk_val[-2 - src]) : rflf(value[g]);
        if (v != v) {   // a NaN from the evaluator must not enter the statistics: the game stops searching, the error is sticky
            if (lane == 0) { st->err = st->err_ever = 1; st->leaf_kind = K_NONE; }
            return;
        }
        const int A = d.geo.A;
        const int lplayer = rfl((int)st->leaf_player);
        const bool term = rfl((int)st->leaf_terminal) != 0;
        int n_nodes = rfl(st->n_nodes), n_edges = rfl(st->n_edges);
        if (kind == K_EXPAND) {
            if (n_nodes >= (int)d.node_cap) { if (lane == 0) { st->err = st->err_ever = 1; st->leaf_kind = K_NONE; } return; }
            node = n_nodes++;
            if (lane == 0) {
                uint4 *pe = edges + path[depth - 1];
                pe->w = (pe->w & 0xFF000000u) | (uint32_t)node;
                st->ctr[5] += 1;
            }
        }
        BB<NW> mask;
#pragma unroll
        for (int i = 0; i < NW; i++) mask.w[i] = rfl64(st->leaf_mask[i]);
        if (!d.aliased) {
            uint64_t *nb = d.nboard + ((size_t)g * d.node_cap + node) * 2 * NW;
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NW; i++) nb[i] = st->leaf_board[i];
            }
        }
        if (term) {                                                                 // mcts.py:63-68
            if (lane == 0)
                nodes[node] = make_uint4(0u, node_pack(0, NF_TERMINAL, lplayer), __float_as_uint(st->leaf_tv), __float_as_uint(v));
        } else {                                                                    // mcts.py:71-89
            const int k = bb_popc(mask);
            if (n_edges + k > (int)d.edge_cap) { if (lane == 0) { st->err = st->err_ever = 1; st->leaf_kind = K_NONE; } return; }
            const float keep = (float)(1.0 - d.eps);
            // a game whose noise row is all zero over its legal moves drew no noise (a Dirichlet draw
            // sums to 1): it keeps the raw priors, like add_exploration_noise=False (mcts.py:298)
            bool mix = false;
            if (noise) {
                uint64_t any = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) {
                    const int cell = j * 64 + lane;
                    const bool nz = ((mask.w[j] >> lane) & 1) && noise[(size_t)g * A + cell] != 0.0;
                    any |= __ballot(nz);
                }
                mix = any != 0;
            }
            int base = n_edges;
            bool bad = false;   // a NaN prior (python: `score > best` is never true for it, select_child returns None and the game raises)
#pragma unroll
            for (int j = 0; j < NW; j++) {
                const int cell = j * 64 + lane;
                if ((mask.w[j] >> lane) & 1) {
                    float p = copy ? cpol[cell] : policy[(size_t)g * A + cell];
                    bad |= (p != p);
                    if (mix) {                                                      // mcts.py:310-312
                        const float kp = __fmul_rn(keep, p);
                        p = (float)__dadd_rn((double)kp, __dmul_rn(d.eps, noise[(size_t)g * A + cell]));
                    }
                    edges[base + mbcnt(mask.w[j])] =
                        make_uint4(__float_as_uint(p), 0u, 0u, CHILD_NONE | ((uint32_t)cell << 24));
                }
                base += yy_popc64(mask.w[j]);
            }
            if (__ballot(bad)) {
                if (lane == 0) { st->err = st->err_ever = 1; st->leaf_kind = K_NONE; }
                return;
            }
            if (lane == 0) {
                // a pass node keeps its value (not from the root call of mcts.py:288: that value is discarded and
                // yy_mcts_expand_root is not given it -- a pass root keeps the value of its first simulation)
                const bool hold = d.reuse && k == 0 && kind != K_ROOTINIT;
                nodes[node] = make_uint4((uint32_t)n_edges, node_pack(k, hold ? NF_HASVALUE : 0u, lplayer), 0u,
                                         __float_as_uint(v));                       // .w = the evaluator's value of this position
                st->ctr[3] += (uint64_t)k;
            }
            n_edges += k;
        }
        if (lane == 0) {
            st->n_nodes = n_nodes;
            st->n_edges = n_edges;
        }
        if (d.ec_meta && !copy && kind != K_ROOTINIT) {                            // a fresh evaluation goes into the cache
            const int slot = rfl(st->leaf_ec_slot);
            const size_t e = (size_t)g * d.ec_cap + slot;
#pragma unroll
            for (int j = 0; j < NW; j++) {
                const int cell = j * 64 + lane;
                if (cell < A) d.ec_pol[e * A + cell] = policy[(size_t)g * A + cell];
            }
            if (lane == 0) {
                int stones = 0;
#pragma unroll
                for (int i = 0; i < 2 * NW; i++) {
                    d.ec_key[e * 2 * NW + i] = st->leaf_board[i];
                    stones += yy_popc64(st->leaf_board[i]);
                }
                d.ec_val[e] = v;
                d.ec_meta[e] = (st->ec_epoch << 8) | (uint32_t)(stones & 0xFF);
            }
        }
    }
    if (lane == 0) st->leaf_kind = K_NONE;
    if (kind == K_ROOTINIT) return;                                                 // no backup
    // ---- backup (mcts.py:406-412): edge i leads to the node at depth i+1; the leaf is at `depth`;
    // players alternate every ply so the sign is the parity of the distance to the leaf.
    for (int i = lane; i < depth; i += 64) {
        uint4 *e = edges + path[i];
        const int dist = depth - (i + 1);
        const float sv = (dist & 1) ? -v : v;
        uint4 r = *e;
        r.y = (uint32_t)((int)r.y + 1);
        r.z = __float_as_uint(__fadd_rn(__uint_as_float(r.z), sv));
        *e = r;
    }
    if (lane == 0) {
        const float sv = (depth & 1) ? -v : v;
        st->root_N += 1;
        if (st->root_w_is_py && v_is_py && depth == 0) {
            // terminal root: python float + python number stays a python float (f64)
            st->root_W_py += (double)((sv == 0.0001f) ? 0.0001 : (sv == -0.0001f ? -0.0001 : (double)sv));
        } else {
            const float base = st->root_w_is_py ? (float)st->root_W_py : st->root_W;
            st->root_W = __fadd_rn(base, sv);
            st->root_w_is_py = 0;
        }
    }
}

template <int NW> __global__ void __launch_bounds__(64) k_mcts(MctsDev d, int do_backup, int do_sel,
                                                               const float *policy, const float *value,
                                                               const double *noise, float *planes,
                                                               uint8_t *needs_eval) {
    const int g = blockIdx.x;
    if (do_backup) do_expand_backup<NW>(d, g, policy, value, noise);
    if (do_backup && do_sel) __syncthreads();   // edges/nodes written above are re-read below by other lanes
    if (do_sel) do_select<NW>(d, g, planes, needs_eval);
}

template <int NW> __global__ void __launch_bounds__(64) k_root_counts(MctsDev d, int32_t *counts, float *cw, float *cp) {
    const int g = blockIdx.x, A = d.geo.A;
    const uint4 hdr = d.nodes[(size_t)g * d.node_cap];
    const int k = node_k(rfl(hdr.y)), first = rfl((int)hdr.x);
    const bool act = rfl((int)d.state[g].active) != 0;
    for (int a = lane_id(); a < A; a += 64) {
        counts[(size_t)g * A + a] = 0;
        if (cw) cw[(size_t)g * A + a] = 0.0f;
        if (cp) cp[(size_t)g * A + a] = 0.0f;
    }
    __syncthreads();
    if (!act) return;
    const uint4 *edges = d.edges + (size_t)g * d.edge_cap;
    for (int j = lane_id(); j < k; j += 64) {
        const uint4 e = edges[first + j];
        const int a = (int)(e.w >> 24);
        counts[(size_t)g * A + a] = (int)e.y;
        if (cw) cw[(size_t)g * A + a] = __uint_as_float(e.z);
        if (cp) cp[(size_t)g * A + a] = __uint_as_float(e.x);
    }
}

// Node.get_children_distribution (mcts.py:183-215) for T == 1 and T == 0
template <int NW> __global__ void __launch_bounds__(64) k_root_policy(MctsDev d, int tzero, double *pi) {
    const int g = blockIdx.x, A = d.geo.A;
    const uint4 hdr = d.nodes[(size_t)g * d.node_cap];
    const int k = node_k(rfl(hdr.y)), first = rfl((int)hdr.x);
    const uint4 *edges = d.edges + (size_t)g * d.edge_cap;
    const bool act = rfl((int)d.state[g].active) != 0;
    int sum = 0, mx = 0;
    if (act)
        for (int j = lane_id(); j < k; j += 64) {
            int n = (int)edges[first + j].y;
            sum += n;
            mx = max(mx, n);
        }
    sum = wave_sum(sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    double fill;
    int nbest = 0;
    if (tzero) {
        // counts == max over ALL A actions (zeros included when max == 0), mcts.py:200-203
        if (mx == 0) nbest = A;
        else {
            for (int j = lane_id(); j < k; j += 64) nbest += ((int)edges[first + j].y == mx);
            nbest = wave_sum(nbest);
        }
        fill = (mx == 0) ? 1.0 / (double)A : 0.0;
    } else {
        fill = (sum > 0) ? 0.0 : 1.0 / (double)A;                                  // mcts.py:209-213
    }
    for (int a = lane_id(); a < A; a += 64) pi[(size_t)g * A + a] = act ? fill : 0.0;
    __syncthreads();
    if (!act) return;
    if (tzero ? (mx > 0) : (sum > 0))
        for (int j = lane_id(); j < k; j += 64) {
            const uint4 e = edges[first + j];
            const int a = (int)(e.w >> 24), n = (int)e.y;
            pi[(size_t)g * A + a] = tzero ? ((n == mx) ? 1.0 / (double)nbest : 0.0) : (double)n / (double)sum;
        }
}

__global__ void k_root_stats(MctsDev d, int32_t *visits, double *wsum) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    const GameState *st = d.state + g;
    if (visits) visits[g] = st->root_N;
    if (wsum) wsum[g] = st->root_w_is_py ? st->root_W_py : (double)st->root_W;
}

template <int NW> __global__ void __launch_bounds__(64) k_get_boards(MctsDev d, int8_t *boards) {
    const int g = blockIdx.x;
    const uint64_t *src = d.gboard + (size_t)g * 2 * NW;
    BB<NW> b = bb_uniform_load<NW>(src), w = bb_uniform_load<NW>(src + NW);
    bb_to_board<NW>(boards + (size_t)g * d.geo.A, d.geo.A, b, w);
}

__global__ void k_status(MctsDev d, uint64_t *out /*[9]: 8 counters, overflow games*/) {
    uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < d.G; g += gridDim.x * blockDim.x) {
        GameState *st = d.state + g;
        for (int i = 0; i < 8; i++) acc[i] += st->ctr[i];
        acc[8] += (st->err || st->err_ever) ? 1 : 0;   // a failure in ANY search since the last status call
        st->err_ever = 0;
    }
    for (int i = 0; i < 9; i++)
        if (acc[i]) atomicAdd((unsigned long long *)&out[i], (unsigned long long)acc[i]);
}

__global__ void k_reset_counters(MctsDev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    for (int i = 0; i < 8; i++) d.state[g].ctr[i] = 0;
}

// ---------------------------------------------------------------------------------- host API
extern "C" int yy_mcts_create(const yy_mcts_config *cfg, yy_mcts **out) {
    if (!cfg || !out) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (int e = check_geo(cfg->G, cfg->R, cfg->C)) return e;
    if (cfg->max_sims < 1) return set_err(YY_E_INVALID, "max_sims < 1%s%s");
    if ((cfg->flags & YY_FLAG_REUSE_PASS_VALUE) && (cfg->flags & YY_FLAG_ALIASED))
        return set_err(YY_E_INVALID, "YY_FLAG_REUSE_PASS_VALUE needs copied boards: with the aliased board a node's position changes between visits%s%s");
    yy_mcts *c = new yy_mcts();
    memset(c, 0, sizeof *c);
    c->cfg = *cfg;
    yy_make_geo(&c->geo, cfg->R, cfg->C, cfg->flags & YY_FLAG_ROWCOL);
    const int A = c->geo.A, NW = c->geo.NW;
    c->node_cap = cfg->nodes_per_game > 0 ? cfg->nodes_per_game : (int64_t)cfg->max_sims + 2;
    c->edge_cap = cfg->edges_per_game > 0 ? cfg->edges_per_game : ((int64_t)cfg->max_sims + 2) * A;
    if (c->node_cap >= (int64_t)CHILD_NONE || c->edge_cap > 0x7FFFFFFFll)
        { delete c; return set_err(YY_E_UNSUPPORTED, "arena per game too large%s%s"); }
    // copied mode: every level places a stone, so depth <= A+1; aliased mode: depth <= sims
    c->path_cap = (cfg->flags & YY_FLAG_ALIASED) ? (int64_t)cfg->max_sims + 2 : (int64_t)A + 2;
    const size_t G = (size_t)cfg->G;
    const bool copied = !(cfg->flags & YY_FLAG_ALIASED);
    const bool ec_on = (cfg->flags & (YY_FLAG_REUSE_TRANSPOSITIONS | YY_FLAG_KEEP_EVALUATIONS)) != 0;
    // slots per game: 4x the node arena (the positions of several searches of a game stay useful), fewer when the
    // policy rows of all games would exceed 48 GiB, never fewer than 2x the arena of one search
    c->ec_cap = 64;
    while (c->ec_cap < 4 * c->node_cap) c->ec_cap *= 2;
    while (c->ec_cap > 2 * c->node_cap && (double)G * (double)c->ec_cap * A * 4.0 > 48.0 * 1073741824.0) c->ec_cap /= 2;
    struct { void **p; size_t n; } allocs[] = {
        {(void **)&c->edges, G * c->edge_cap * sizeof(uint4)},
        {(void **)&c->nodes, G * c->node_cap * sizeof(uint4)},
        {(void **)&c->nboard, copied ? G * c->node_cap * 2 * NW * sizeof(uint64_t) : 8},
        {(void **)&c->gboard, G * 2 * NW * sizeof(uint64_t)},
        {(void **)&c->path, G * c->path_cap * sizeof(int32_t)},
        {(void **)&c->state, G * sizeof(GameState)},
        {(void **)&c->sqrt_tab, (size_t)(cfg->max_sims + 2) * sizeof(float)},
        {(void **)&c->scratch, 9 * sizeof(uint64_t)},
        {(void **)&c->ec_meta, ec_on ? G * (size_t)c->ec_cap * sizeof(uint32_t) : 0},
        {(void **)&c->ec_key, ec_on ? G * (size_t)c->ec_cap * 2 * NW * sizeof(uint64_t) : 0},
        {(void **)&c->ec_val, ec_on ? G * (size_t)c->ec_cap * sizeof(float) : 0},
        {(void **)&c->ec_pol, ec_on ? G * (size_t)c->ec_cap * A * sizeof(float) : 0},
    };
    for (auto &a : allocs) {
        if (a.n == 0) continue;
        const hipError_t me = hipMalloc(a.p, a.n);
        if (me != hipSuccess) {
            (void)hipGetLastError();
            char what[96];
            snprintf(what, sizeof what, " (%zu bytes after %llu allocated)", a.n, (unsigned long long)c->bytes);
            *a.p = nullptr;
            yy_mcts_destroy(c);
            return set_err(YY_E_NOMEM, "hipMalloc failed: %s%s", hipGetErrorString(me), what);
        }
        c->bytes += a.n;
    }
    // f32(math.sqrt(S)) table built with the host's correctly rounded double sqrt (mcts.py:130)
    float *tab = new float[cfg->max_sims + 2];
    for (int s = 0; s < cfg->max_sims + 2; s++) tab[s] = (float)sqrt((double)s);
    hipError_t e = hipMemset(c->state, 0, G * sizeof(GameState));
    if (e == hipSuccess) e = hipMemset(c->nodes, 0, G * c->node_cap * sizeof(uint4));
    if (e == hipSuccess && c->ec_meta) e = hipMemset(c->ec_meta, 0, G * (size_t)c->ec_cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(c->sqrt_tab, tab, (size_t)(cfg->max_sims + 2) * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    delete[] tab;
    if (e != hipSuccess) { yy_mcts_destroy(c); return set_err(YY_E_HIP, "arena initialisation: %s%s", hipGetErrorString(e)); }
    *out = c;
    return YY_OK;
}

extern "C" int yy_mcts_destroy(yy_mcts *c) {
    if (!c) return YY_OK;
    void *ps[] = {c->edges, c->nodes, c->nboard, c->gboard, c->path, c->state, c->sqrt_tab, c->scratch, c->ec_meta, c->ec_key, c->ec_val, c->ec_pol};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    delete c;
    return YY_OK;
}

extern "C" int yy_mcts_memory_bytes(const yy_mcts *c, uint64_t *out) {
    if (!c || !out) return set_err(YY_E_INVALID, "null pointer%s%s");
    *out = c->bytes;
    return YY_OK;
}

extern "C" int yy_mcts_begin(yy_mcts *c, const int8_t *boards, const int8_t *players, const uint8_t *active,
                             float *planes, yy_stream_t s) {
    if (!c || !boards || !players || !planes) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    DISPATCH_NW(c->geo.NW, k_begin<NW><<<dim3(c->cfg.G), dim3(64), 0, (hipStream_t)s>>>(d, boards,
                                               players, active, planes));
    HIP_TRY(hipGetLastError());
    c->pending = 2;  // root expansion pending
    return YY_OK;
}

static int launch_mcts(yy_mcts *c, int backup, int sel, const float *policy, const float *value, const double *noise,
                       double eps, float *planes, uint8_t *needs_eval, yy_stream_t s) {
    MctsDev d = make_dev(c);
    d.eps = eps;
    DISPATCH_NW(c->geo.NW, k_mcts<NW><<<dim3(c->cfg.G), dim3(64), 0, (hipStream_t)s>>>(d, backup, sel,
                                               policy, value, noise, planes, needs_eval));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_expand_root(yy_mcts *c, const float *policy, const double *noise, double eps, yy_stream_t s) {
    if (!c || !policy) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (c->pending != 2) return set_err(YY_E_STATE, "yy_mcts_expand_root without yy_mcts_begin%s%s");
    int e = launch_mcts(c, 1, 0, policy, nullptr, noise, eps, nullptr, nullptr, s);
    if (e == YY_OK) c->pending = 0;
    return e;
}

extern "C" int yy_mcts_select(yy_mcts *c, float *planes, uint8_t *needs_eval, yy_stream_t s) {
    if (!c || !planes) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (c->pending != 0) return set_err(YY_E_STATE, "yy_mcts_select with an expansion pending%s%s");
    int e = launch_mcts(c, 0, 1, nullptr, nullptr, nullptr, 0.0, planes, needs_eval, s);
    if (e == YY_OK) c->pending = 1;
    return e;
}

extern "C" int yy_mcts_expand_backup(yy_mcts *c, const float *policy, const float *value, yy_stream_t s) {
    if (!c || !policy || !value) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (c->pending != 1) return set_err(YY_E_STATE, "yy_mcts_expand_backup without yy_mcts_select%s%s");
    int e = launch_mcts(c, 1, 0, policy, value, nullptr, 0.0, nullptr, nullptr, s);
    if (e == YY_OK) c->pending = 0;
    return e;
}

extern "C" int yy_mcts_step(yy_mcts *c, const float *policy, const float *value, float *planes, uint8_t *needs_eval,
                            yy_stream_t s) {
    if (!c || !policy || !value || !planes) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (c->pending != 1) return set_err(YY_E_STATE, "yy_mcts_step without a pending select%s%s");
    return launch_mcts(c, 1, 1, policy, value, nullptr, 0.0, planes, needs_eval, s);
}

extern "C" int yy_mcts_root_counts(yy_mcts *c, int32_t *counts, float *cw, float *cp, yy_stream_t s) {
    if (!c || !counts) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    DISPATCH_NW(c->geo.NW,
                k_root_counts<NW><<<dim3(c->cfg.G), dim3(64), 0, (hipStream_t)s>>>(d, counts, cw, cp));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_root_policy(yy_mcts *c, int tzero, double *pi, yy_stream_t s) {
    if (!c || !pi) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    DISPATCH_NW(c->geo.NW,
                k_root_policy<NW><<<dim3(c->cfg.G), dim3(64), 0, (hipStream_t)s>>>(d, tzero, pi));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_root_stats(yy_mcts *c, int32_t *visits, double *wsum, yy_stream_t s) {
    if (!c) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    hipLaunchKernelGGL(k_root_stats, dim3((c->cfg.G + 255) / 256), dim3(256), 0, (hipStream_t)s, d, visits, wsum);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_get_boards(yy_mcts *c, int8_t *boards, yy_stream_t s) {
    if (!c || !boards) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    DISPATCH_NW(c->geo.NW, k_get_boards<NW><<<dim3(c->cfg.G), dim3(64), 0, (hipStream_t)s>>>(d, boards));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_status(yy_mcts *c, int32_t *n_overflow, uint64_t *counters) {
    if (!c) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(c->scratch, 0, 9 * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_status, dim3(64), dim3(256), 0, 0, d, c->scratch);
    HIP_TRY(hipGetLastError());
    uint64_t h[9];
    HIP_TRY(hipMemcpy(h, c->scratch, sizeof h, hipMemcpyDeviceToHost));
    if (counters) {
        for (int i = 0; i < 8; i++) counters[i] = h[i];
    }
    if (n_overflow) *n_overflow = (int32_t)h[8];
    if (h[8]) return set_err(YY_E_ARENA, "tree arena overflow or non-finite evaluator output in at least one game since the last status call%s%s");
    return YY_OK;
}

// ---- shared book of pre-evaluated positions
template <int NW> __global__ void k_book_insert(const uint64_t *__restrict__ keys, int n, uint32_t *meta, uint64_t *tkeys,
                                                uint32_t mask, int32_t *slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *k = keys + (size_t)i * 2 * NW;
    uint32_t at = (uint32_t)bb_hash<NW>(k, k + NW) & mask;
    int got = -1;
    for (int pr = 0; pr < TT_PROBES; pr++, at = (at + 1u) & mask)
        if (atomicCAS(&meta[at], 0u, 1u) == 0u) { got = (int)at; break; }      // keys are distinct: a taken slot is another position
    if (got >= 0)
        for (int j = 0; j < 2 * NW; j++) tkeys[(size_t)got * 2 * NW + j] = k[j];
    slot_of[i] = got;      // -1: no free slot within the probe window (the position stays out of the book)
}

extern "C" int yy_book_insert(const uint64_t *keys, int n, int R, int C, uint32_t *meta, uint64_t *table_keys, int64_t cap,
                              int32_t *slot_of, yy_stream_t s) {
    if (n == 0) return YY_OK;
    if (!keys || !meta || !table_keys || !slot_of || n < 0 || cap < 64 || (cap & (cap - 1)) || cap > (1ll << 31))
        return set_err(YY_E_INVALID, "yy_book_insert: bad argument (cap must be a power of two)%s%s");
    if (int e = check_geo(1, R, C)) return e;
    const int nw = (R * C + 63) / 64;
    DISPATCH_NW(nw, k_book_insert<NW><<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s>>>(keys, n, meta, table_keys,
                                                                                               (uint32_t)(cap - 1), slot_of));
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_mcts_set_book(yy_mcts *c, const uint32_t *meta, const uint64_t *keys, const float *val, const float *pol,
                                int64_t cap, int max_stones) {
    if (!c) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (!meta) { c->bk_meta = nullptr; c->bk_key = nullptr; c->bk_val = c->bk_pol = nullptr; c->bk_cap = 0; c->bk_stones = 0; return YY_OK; }
    if (!keys || !val || !pol || cap < 64 || (cap & (cap - 1)) || cap > (1ll << 31) || max_stones < 1)
        return set_err(YY_E_INVALID, "yy_mcts_set_book: bad argument (cap must be a power of two)%s%s");
    c->bk_meta = meta; c->bk_key = keys; c->bk_val = val; c->bk_pol = pol; c->bk_cap = cap; c->bk_stones = max_stones;
    return YY_OK;
}

extern "C" int yy_mcts_cache_clear(yy_mcts *c, yy_stream_t s) {
    if (!c) return set_err(YY_E_INVALID, "null pointer%s%s");
    if (c->ec_meta) HIP_TRY(hipMemsetAsync(c->ec_meta, 0, (size_t)c->cfg.G * (size_t)c->ec_cap * sizeof(uint32_t), (hipStream_t)s));
    return YY_OK;
}

extern "C" int yy_mcts_reset_counters(yy_mcts *c, yy_stream_t s) {
    if (!c) return set_err(YY_E_INVALID, "null pointer%s%s");
    MctsDev d = make_dev(c);
    hipLaunchKernelGGL(k_reset_counters, dim3((c->cfg.G + 255) / 256), dim3(256), 0, (hipStream_t)s, d);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

// =============================================================================== evaluator epilogue
// Fused bias + residual + ReLU over a channels-last bf16 activation tensor, in place:
//     x[r, c] = relu( x[r, c] + bias[c] (+ residual[r, c]) )
// This replaces the 3-4 separate elementwise passes PyTorch/MIOpen run after every convolution of
// the policy/value tower (bias add, residual add, clamp) by ONE pass: 16-B loads/stores per lane,
// f32 arithmetic, one bf16 rounding.  HBM-bound: (2 or 3) * rows * C * 2 bytes per launch.
typedef __attribute__((ext_vector_type(8))) unsigned short us8;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even, NaN stays NaN
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

template <bool HAS_RES, bool RELU, bool FIXED>
__global__ void __launch_bounds__(256) k_bias_act(us8 *x, const float *bias, const us8 *res, size_t n_vec, int cvec) {
    // cvec = C / 8 vectors per row.  FIXED: the grid stride is a multiple of cvec, so a lane keeps the
    // same 8 channels for its whole grid-stride walk and the bias slice lives in registers.
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float b[8];
    if (FIXED) {
        const float4 *bp = reinterpret_cast<const float4 *>(bias + (tid % (size_t)cvec) * 8);
        const float4 b0 = bp[0], b1 = bp[1];
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
        b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    for (size_t i = tid; i < n_vec; i += stride) {
        if (!FIXED) {
            const int cb = (int)(i % (size_t)cvec) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = bias[cb + j];
        }
        const us8 v = x[i];
        us8 r;
        if (HAS_RES) r = res[i];
        us8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float f = bf2f(v[j]) + b[j];
            if (HAS_RES) f += bf2f(r[j]);
            if (RELU) f = fmaxf(f, 0.0f);
            o[j] = f2bf(f);
        }
        x[i] = o;
    }
}

template <bool FIXED>
static void launch_bias_act(us8 *xv, const float *bias, const us8 *rv, size_t n_vec, int cvec, int relu, unsigned blocks,
                            hipStream_t st) {
    if (rv) {
        if (relu) k_bias_act<true, true, FIXED><<<dim3(blocks), dim3(256), 0, st>>>(xv, bias, rv, n_vec, cvec);
        else k_bias_act<true, false, FIXED><<<dim3(blocks), dim3(256), 0, st>>>(xv, bias, rv, n_vec, cvec);
    } else {
        if (relu) k_bias_act<false, true, FIXED><<<dim3(blocks), dim3(256), 0, st>>>(xv, bias, rv, n_vec, cvec);
        else k_bias_act<false, false, FIXED><<<dim3(blocks), dim3(256), 0, st>>>(xv, bias, rv, n_vec, cvec);
    }
}

extern "C" int yy_nn_bias_act_bf16(void *x, const float *bias, const void *residual, int64_t rows, int C, int relu,
                                   yy_stream_t s) {
    if (rows == 0) return YY_OK;
    if (!x || !bias || rows < 0 || C <= 0) return set_err(YY_E_INVALID, "bad argument%s%s");
    if (C % 8) return set_err(YY_E_UNSUPPORTED, "channels must be a multiple of 8%s%s");
    const int cvec = C / 8;
    const size_t n_vec = (size_t)rows * (size_t)cvec;
    size_t blocks = (n_vec + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;          // ~8 blocks per CU, grid-stride the rest
    const bool fixed = (256 % cvec) == 0;            // grid stride (blocks*256) is then a multiple of cvec
    if (fixed) launch_bias_act<true>((us8 *)x, bias, (const us8 *)residual, n_vec, cvec, relu, (unsigned)blocks, (hipStream_t)s);
    else launch_bias_act<false>((us8 *)x, bias, (const us8 *)residual, n_vec, cvec, relu, (unsigned)blocks, (hipStream_t)s);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

// Head finish (neural_network.py:115, 120-121 + predict's softmax :152): row g of h holds the A policy logits
// followed by the H hidden activations of value_fc1 (bias already added by the GEMM, bf16).  One wave per
// row: policy = softmax(logits) in f32; value = tanh(sum_j relu(hidden_j) * w2_j + b2).
__global__ void __launch_bounds__(64) k_head_finish(const unsigned short *__restrict__ h, int A, int H,
                                                    const float *__restrict__ w2, const float *__restrict__ b2,
                                                    float *__restrict__ policy, float *__restrict__ value) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const unsigned short *row = h + (size_t)g * (A + H);
    float mx = -INFINITY;
    for (int a = lane; a < A; a += 64) mx = fmaxf(mx, bf2f(row[a]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (int a = lane; a < A; a += 64) sum += expf(bf2f(row[a]) - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    for (int a = lane; a < A; a += 64) policy[(size_t)g * A + a] = expf(bf2f(row[a]) - mx) / sum;
    float acc = 0.0f;
    for (int j = lane; j < H; j += 64) acc += fmaxf(bf2f(row[A + j]), 0.0f) * w2[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) value[g] = tanhf(acc + b2[0]);
}

// float32 head finish (evaluator modes "f16x3" / float32 towers): logits f32 [Gd, A] and value_fc1 outputs f32 [Gd, H]
// (bias added, no ReLU yet) of DENSE row i -> policy[g] = softmax(logits[i]), value[g] = tanh(relu(hidden[i]) . w2 + b2)
// with g = rows ? rows[i] : i; blocks i >= *n_rows exit (the rows a compacted launch did not evaluate).
__global__ void __launch_bounds__(64) k_head_finish_f32(const float *__restrict__ logits, const float *__restrict__ hidden,
                                                        int A, int H, const float *__restrict__ w2,
                                                        const float *__restrict__ b2, const int32_t *__restrict__ rows,
                                                        const int32_t *__restrict__ n_rows, float *__restrict__ policy,
                                                        float *__restrict__ value) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (n_rows && i >= *n_rows) return;
    const int g = rows ? rows[i] : i;
    const float *row = logits + (size_t)i * A;
    float mx = -INFINITY;
    for (int a = lane; a < A; a += 64) mx = fmaxf(mx, row[a]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (int a = lane; a < A; a += 64) sum += expf(row[a] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    for (int a = lane; a < A; a += 64) policy[(size_t)g * A + a] = expf(row[a] - mx) / sum;
    const float *hr = hidden + (size_t)i * H;
    float acc = 0.0f;
    for (int j = lane; j < H; j += 64) acc += fmaxf(hr[j], 0.0f) * w2[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) value[g] = tanhf(acc + b2[0]);
}

extern "C" int yy_nn_head_finish_f32(const float *logits, const float *hidden, int G, int A, int H, const float *w2,
                                     const float *b2, const int32_t *rows, const int32_t *n_rows, float *policy, float *value,
                                     yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!logits || !hidden || !w2 || !b2 || !policy || !value || G < 0 || A <= 0 || H <= 0 || (rows && !n_rows))
        return set_err(YY_E_INVALID, "bad argument%s%s");
    k_head_finish_f32<<<dim3(G), dim3(64), 0, (hipStream_t)s>>>(logits, hidden, A, H, w2, b2, rows, n_rows, policy, value);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

// Leaf-batch compaction: rows[0 .. *n) = the indices g with flags[g] != 0, ascending; one 1024-thread workgroup.
__global__ void __launch_bounds__(1024) k_compact_rows(const uint8_t *__restrict__ flags, int G, int32_t *__restrict__ rows,
                                                       int32_t *__restrict__ n) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (G + 1023) / 1024, lo = min(t * per, G), hi = min(lo + per, G);
    int cnt = 0;
    for (int g = lo; g < hi; g++) cnt += flags[g] != 0;
    int incl = cnt;                                             // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    int pos = base + incl - cnt;
    for (int g = lo; g < hi; g++)
        if (flags[g] != 0) rows[pos++] = g;
    if (t == 1023) *n = base + incl;
}

extern "C" int yy_compact_rows(const uint8_t *flags, int G, int32_t *rows, int32_t *n, yy_stream_t s) {
    if (!flags || !rows || !n || G < 0) return set_err(YY_E_INVALID, "bad argument%s%s");
    k_compact_rows<<<dim3(1), dim3(1024), 0, (hipStream_t)s>>>(flags, G, rows, n);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

extern "C" int yy_nn_head_finish_bf16(const void *h, int G, int A, int H, const float *w2, const float *b2, float *policy,
                                      float *value, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!h || !w2 || !b2 || !policy || !value || G < 0 || A <= 0 || H <= 0) return set_err(YY_E_INVALID, "bad argument%s%s");
    k_head_finish<<<dim3(G), dim3(64), 0, (hipStream_t)s>>>((const unsigned short *)h, A, H, w2, b2, policy, value);
    HIP_TRY(hipGetLastError());
    return YY_OK;
}

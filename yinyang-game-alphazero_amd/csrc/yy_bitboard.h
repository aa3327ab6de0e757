// yy_bitboard.h -- Yin-Yang rules on multiword bitboards (device + host).
//
// A board is two bitboards (black, white); bit a = cell a = x*C+y (the action index of
// src/yin_yang/yin_yang_game.py:180-186), packed into NW = ceil(R*C/64) 64-bit words.  An 8x8
// board is exactly one word == one wave64 ballot mask.
//
// The legal-move mask is the decomposition of YinYangLogic.is_valid_move
// (src/yin_yang/yin_yang_logic.py:31-56) proven in SURVEY.md 9.1: a cell c is legal for colour p
// iff   (1) c is empty,
//       (2) no monochrome 2x2 of either colour exists already (the check at :96-109 is global
//           and colour-agnostic, so a pre-existing block makes every move illegal),
//       (3) c is 4-adjacent to EVERY connected component of p (or p has no stones)  [:58-94],
//       (4) none of the <=4 windows containing c has its three other cells == p     [:96-109].
// The same functions run with wave-uniform operands (tree kernels: one game per wave, values in
// SGPRs) and with per-lane operands (stateless kernels: one game per lane).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define YY_HD __host__ __device__ __forceinline__
#else
#define YY_HD inline
#endif

#define YY_MAX_NW 3
#define YY_MAX_DIM 16

// Board geometry + precomputed masks; passed by value to kernels (lands in SGPRs).
struct YYGeo {
    int32_t R, C, A, NW;
    uint64_t full[YY_MAX_NW];   // bits < A
    uint64_t notc0[YY_MAX_NW];  // cells with column > 0
    uint64_t notcl[YY_MAX_NW];  // cells with column < C-1
    uint64_t col0[YY_MAX_NW];   // cells with column == 0
    float rowfill[YY_MAX_DIM + 1];  // (float)((double)k / (double)C)  neural_network.py:186-188
    float colfill[YY_MAX_DIM + 1];  // (float)((double)k / (double)R)  neural_network.py:191-194
    uint32_t flags;
};

template <int NW>
struct BB {
    uint64_t w[NW];
};

template <int NW> YY_HD BB<NW> bb_zero() {
    BB<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = 0;
    return r;
}
template <int NW> YY_HD BB<NW> bb_load(const uint64_t *m) {
    BB<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = m[i];
    return r;
}
template <int NW> YY_HD BB<NW> operator&(BB<NW> a, BB<NW> b) {
#pragma unroll
    for (int i = 0; i < NW; i++) a.w[i] &= b.w[i];
    return a;
}
template <int NW> YY_HD BB<NW> operator|(BB<NW> a, BB<NW> b) {
#pragma unroll
    for (int i = 0; i < NW; i++) a.w[i] |= b.w[i];
    return a;
}
template <int NW> YY_HD BB<NW> bb_andn(BB<NW> a, BB<NW> b) {  // a & ~b
#pragma unroll
    for (int i = 0; i < NW; i++) a.w[i] &= ~b.w[i];
    return a;
}
template <int NW> YY_HD bool bb_any(BB<NW> a) {
    uint64_t o = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) o |= a.w[i];
    return o != 0;
}
template <int NW> YY_HD bool bb_eq(BB<NW> a, BB<NW> b) {
    uint64_t o = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) o |= a.w[i] ^ b.w[i];
    return o == 0;
}
YY_HD int yy_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
template <int NW> YY_HD int bb_popc(BB<NW> a) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) n += yy_popc64(a.w[i]);
    return n;
}
template <int NW> YY_HD bool bb_test(BB<NW> a, int bit) {
    bool r = false;
#pragma unroll
    for (int i = 0; i < NW; i++)
        if ((bit >> 6) == i) r = (a.w[i] >> (bit & 63)) & 1;
    return r;
}
template <int NW> YY_HD BB<NW> bb_bit(int bit) {
    BB<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = ((bit >> 6) == i) ? (1ull << (bit & 63)) : 0ull;
    return r;
}
// lowest set bit isolated (zero if empty)
template <int NW> YY_HD BB<NW> bb_lowest(BB<NW> a) {
    BB<NW> r;
    bool found = false;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint64_t l = a.w[i] & (0 - a.w[i]);
        r.w[i] = found ? 0ull : l;
        found = found || (l != 0);
    }
    return r;
}
// towards higher cell index by n bits, 0 <= n < 64
template <int NW> YY_HD BB<NW> bb_shl(BB<NW> a, int n) {
    if (n == 0) return a;
    BB<NW> r;
#pragma unroll
    for (int i = NW - 1; i >= 0; i--) {
        uint64_t v = a.w[i] << n;
        if (i > 0) v |= a.w[i - 1] >> (64 - n);
        r.w[i] = v;
    }
    return r;
}
template <int NW> YY_HD BB<NW> bb_shr(BB<NW> a, int n) {
    if (n == 0) return a;
    BB<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint64_t v = a.w[i] >> n;
        if (i + 1 < NW) v |= a.w[i + 1] << (64 - n);
        r.w[i] = v;
    }
    return r;
}

template <int NW> struct GeoBB {
    BB<NW> full, notc0, notcl;
    int C;
};
template <int NW> YY_HD GeoBB<NW> geo_bb(const YYGeo &g) {
    GeoBB<NW> r;
    r.full = bb_load<NW>(g.full);
    r.notc0 = bb_load<NW>(g.notc0);
    r.notcl = bb_load<NW>(g.notcl);
    r.C = g.C;
    return r;
}

// cells 4-adjacent to x (without x itself)
template <int NW> YY_HD BB<NW> bb_neighbors(BB<NW> x, const GeoBB<NW> &g) {
    BB<NW> e = bb_shl(x, 1) & g.notc0;         // (r, c+1)
    BB<NW> w = bb_shr(x, 1) & g.notcl;         // (r, c-1)
    BB<NW> s = bb_shl(x, g.C) & g.full;        // (r+1, c)
    BB<NW> n = bb_shr(x, g.C);                 // (r-1, c)
    return (e | w) | (s | n);
}

// connected component of `seed` inside S (yin_yang_logic.py:73-91 as iterated masked dilation)
template <int NW> YY_HD BB<NW> bb_flood(BB<NW> seed, BB<NW> S, const GeoBB<NW> &g) {
    BB<NW> comp = seed;
    for (int it = 0; it < YY_MAX_NW * 64; it++) {
        BB<NW> nxt = comp | (bb_neighbors(comp, g) & S);
        if (bb_eq(nxt, comp)) break;
        comp = nxt;
    }
    return comp;
}

// anchors (top-left cells) of monochrome 2x2 windows inside X
template <int NW> YY_HD BB<NW> bb_mono2x2(BB<NW> X, const GeoBB<NW> &g) {
    BB<NW> right = bb_shr(X, 1) & g.notcl;
    BB<NW> down = bb_shr(X, g.C);
    BB<NW> diag = bb_shr(X, g.C + 1) & g.notcl;
    return (X & right) & (down & diag);
}

// cells c such that some 2x2 window containing c has its other three cells in S
template <int NW> YY_HD BB<NW> bb_completes2x2(BB<NW> S, const GeoBB<NW> &g) {
    BB<NW> e = bb_shr(S, 1) & g.notcl;             // S at (r, c+1) seen from (r, c)
    BB<NW> w = bb_shl(S, 1) & g.notc0;             // S at (r, c-1)
    BB<NW> s = bb_shr(S, g.C);                     // S at (r+1, c)
    BB<NW> n = bb_shl(S, g.C) & g.full;            // S at (r-1, c)
    BB<NW> se = bb_shr(S, g.C + 1) & g.notcl;      // (r+1, c+1)
    BB<NW> sw = bb_shr(S, g.C - 1) & g.notc0;      // (r+1, c-1)
    BB<NW> ne = bb_shl(S, g.C - 1) & g.notcl & g.full;  // (r-1, c+1)
    BB<NW> nw = bb_shl(S, g.C + 1) & g.notc0 & g.full;  // (r-1, c-1)
    BB<NW> tl = (e & s) & se;
    BB<NW> tr = (w & s) & sw;
    BB<NW> bl = (e & n) & ne;
    BB<NW> br = (w & n) & nw;
    return (tl | tr) | (bl | br);
}

// Optional browser-only rule (yin_yang_game.js:338-384): cells of `empty` whose placement (colour
// own) leaves some FULL row or FULL column single-coloured; all cells if one exists already.
template <int NW> YY_HD BB<NW> bb_rowcol_bad(BB<NW> own, BB<NW> opp, int R, int C, const GeoBB<NW> &g) {
    BB<NW> occ = own | opp;
    BB<NW> empty = bb_andn(g.full, occ);
    BB<NW> bad = bb_zero<NW>();
    bool pre = false;
    for (int r = 0; r < R; r++) {
        BB<NW> rowm = bb_zero<NW>();
        for (int c = 0; c < C; c++) rowm = rowm | bb_bit<NW>(r * C + c);
        int ne = bb_popc(empty & rowm), no = bb_popc(own & rowm), np = bb_popc(opp & rowm);
        if (ne == 0 && (no == 0 || np == 0)) pre = true;
        if (ne == 1 && np == 0) bad = bad | (empty & rowm);
    }
    for (int c = 0; c < C; c++) {
        BB<NW> colm = bb_zero<NW>();
        for (int r = 0; r < R; r++) colm = colm | bb_bit<NW>(r * C + c);
        int ne = bb_popc(empty & colm), no = bb_popc(own & colm), np = bb_popc(opp & colm);
        if (ne == 0 && (no == 0 || np == 0)) pre = true;
        if (ne == 1 && np == 0) bad = bad | (empty & colm);
    }
    return pre ? g.full : bad;
}

// Legal mask of colour `own` (getValidMoves, yin_yang_game.py:60-78).
// pre2x2: result of bb_any(mono2x2(own)|mono2x2(opp)), shared between the two colours.
template <int NW>
YY_HD BB<NW> bb_legal(BB<NW> own, BB<NW> opp, bool pre2x2, const YYGeo &geo, const GeoBB<NW> &g) {
    BB<NW> empty = bb_andn(g.full, own | opp);
    if (pre2x2) return bb_zero<NW>();
    BB<NW> cand = empty;
    BB<NW> rest = own;
    // at most 4 components can all touch one cell; a 5th makes the mask empty
    for (int k = 0; k < 5; k++) {
        if (!bb_any(rest)) break;
        if (k == 4 || !bb_any(cand)) {
            cand = bb_zero<NW>();
            break;
        }
        BB<NW> comp = bb_flood(bb_lowest(rest), rest, g);
        cand = cand & bb_neighbors(comp, g);
        rest = bb_andn(rest, comp);
    }
    cand = bb_andn(cand, bb_completes2x2(own, g));
    if (geo.flags & 1u) cand = bb_andn(cand, bb_rowcol_bad(own, opp, geo.R, geo.C, g));
    return cand;
}

template <int NW> YY_HD bool bb_pre2x2(BB<NW> black, BB<NW> white, const GeoBB<NW> &g) {
    return bb_any(bb_mono2x2(black, g) | bb_mono2x2(white, g));
}

// getGameEnded from BLACK's point of view given both masks (yin_yang_game.py:80-110):
// 0 ongoing, +1 black wins, -1 white wins, 2 draw.
template <int NW>
YY_HD int bb_result_black(BB<NW> black, BB<NW> white, BB<NW> mask_b, BB<NW> mask_w) {
    if (bb_any(mask_b) || bb_any(mask_w)) return 0;
    int nb = bb_popc(black), nw = bb_popc(white);
    return nb > nw ? 1 : (nw > nb ? -1 : 2);
}

inline void yy_make_geo(YYGeo *g, int R, int C, uint32_t flags) {
    g->R = R;
    g->C = C;
    g->A = R * C;
    g->NW = (g->A + 63) / 64;
    g->flags = flags;
    for (int i = 0; i < YY_MAX_NW; i++) g->full[i] = g->notc0[i] = g->notcl[i] = g->col0[i] = 0;
    for (int a = 0; a < g->A; a++) {
        uint64_t bit = 1ull << (a & 63);
        int w = a >> 6, c = a % C;
        g->full[w] |= bit;
        if (c > 0) g->notc0[w] |= bit;
        if (c < C - 1) g->notcl[w] |= bit;
        if (c == 0) g->col0[w] |= bit;
    }
    for (int k = 0; k <= YY_MAX_DIM; k++) {
        g->rowfill[k] = (float)((double)k / (double)C);
        g->colfill[k] = (float)((double)k / (double)R);
    }
}

/*
 * yy_engine.h -- C ABI of libyy_hip.so, the MI355X (gfx950) Yin-Yang self-play hot path.
 *
 * The reference (Arash-san/YinYang-Game-AlphaZero) is pure Python and has no FFI: the drop-in
 * boundary is its Python class API.  This header is the C-ABI a maintainer would bind with ctypes
 * from those classes (INTEGRATION.md shows the stubs); every entry point cites the reference
 * interface it replaces as path:line under the reference root.
 *
 * Conventions
 *   - plain pointers and sizes only; ALL data pointers are DEVICE pointers (HBM) unless the
 *     parameter is documented as host; no torch types.
 *   - every function returns an int status (YY_OK or a negative YY_E_*); yy_last_error() gives a
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no call synchronises unless documented ("sync").
 *   - boards are int8 [G,R,C] row-major, 0 empty / +1 black / -1 white
 *     (src/yin_yang/yin_yang_logic.py:8-18); players are int8 +1 / -1; an action is x*C+y
 *     (src/yin_yang/yin_yang_game.py:180-186).  1 <= R,C <= 16 and R*C <= 192.
 */
#ifndef YY_ENGINE_H
#define YY_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YY_OK 0
#define YY_E_INVALID (-1)     /* null pointer, non-positive size, bad enum */
#define YY_E_UNSUPPORTED (-2) /* board larger than 16x16 / 192 cells */
#define YY_E_NOMEM (-3)       /* hipMalloc failed */
#define YY_E_HIP (-4)         /* HIP runtime error (message in yy_last_error) */
#define YY_E_STATE (-5)       /* call order violation (e.g. expand before select) */
#define YY_E_ARENA (-6)       /* a game's tree arena overflowed (reported by yy_mcts_status) */

/* rule / semantics flags */
#define YY_FLAG_ROWCOL 1u  /* also enforce the browser-only full-row/column rule
                              (src/gui/static/js/yin_yang_game.js:338-384). Default off: the
                              Python rules have no such rule. */
#define YY_FLAG_ALIASED 2u /* MCTS shares ONE board per game through the whole tree and mutates
                              it, exactly like the reference (yin_yang_game.py:52-58 +
                              ai/mcts.py:385-397).  Default (flag clear) = "copied": every
                              expanded node owns its board. */
#define YY_FLAG_REUSE_PASS_VALUE 4u /* copied boards only (yy_mcts_create refuses it with YY_FLAG_ALIASED).  The
                              reference evaluates a non-terminal node without legal moves again on EVERY visit
                              (ai/mcts.py:93-95 is_expanded() is False without children, :371-397 evaluate + expand).
                              With copied boards the node's position never changes, so the evaluator returns the
                              same value each time; with this flag the value of the first evaluation is kept in the
                              node record and a revisit takes it from there: no planes written, needs_eval = 0,
                              counters[6] += 1.  Visit counts, value sums and pi are unchanged (tests/test_gpu_mcts.py
                              ::test_evaluation_reuse_*); counters[0] (evaluator rows) gets smaller.  Default off:
                              the evaluator call sequence is then the reference's, row for row. */

#define YY_FLAG_REUSE_TRANSPOSITIONS 8u /* evaluation cache, one search at a time.  The reference evaluates every leaf
                              (ai/mcts.py:371-397), also when the position was evaluated before: another move order
                              inside the same search, a pass node visited again.  A deterministic evaluator returns the
                              same (policy, value) each time -- its input planes encode the board only, not the side to
                              move (ai/neural_network.py:156-196).  With this flag every game keeps a table in HBM
                              [position -> policy row f32[A], value], open addressing on a 64-bit mix of the bitboards,
                              <= 8 probes, keys compared in full; a leaf whose position is in the table takes its
                              evaluation from there instead of an evaluator row (needs_eval = 0, no planes written,
                              counters[7] += 1).  Only entries written by the current search are used.  Same visit
                              counts, value sums and pi (tests/test_gpu_mcts.py::test_evaluation_reuse_*).  Works with
                              copied and aliased boards (the key is the position, not the node).  Default off: the
                              evaluator's call sequence is then the reference's, row for row. */
#define YY_FLAG_KEEP_EVALUATIONS 16u /* implies the table of YY_FLAG_REUSE_TRANSPOSITIONS and keeps its entries from one
                              search to the next (the next move's search re-derives much of the subtree the game walked
                              into: at 8x8 / 800 simulations about half of the rows that are new within a search were
                              evaluated by an earlier search of the same game).  Entries are replaced oldest-game first,
                              then positions that cannot recur (no more stones than the root).  REQUIRES that every
                              search of the context uses the same evaluator: call yy_mcts_cache_clear when it changes. */

typedef void *yy_stream_t;

const char *yy_last_error(void);
int yy_version(void);

/* ------------------------------------------------------------------ stateless rules kernels */

/* YinYangGame.getValidMoves(board, player)  (yin_yang_game.py:60-78 over
 * yin_yang_logic.py:31-128).  out_mask: uint8 [G,A], 1 = legal.  players: int8 [G]. */
int yy_rules_valid_mask(const int8_t *boards, const int8_t *players, int G, int R, int C,
                        uint32_t flags, uint8_t *out_mask, yy_stream_t stream);

/* YinYangGame.getNextState(board, player, action)  (yin_yang_game.py:39-58,
 * yin_yang_logic.py:24-29): place iff legal IN PLACE, then players[g] = -players[g] regardless.
 * placed: uint8 [G] (may be NULL). */
int yy_rules_step(int8_t *boards, int8_t *players, const int32_t *actions, int G, int R, int C,
                  uint32_t flags, uint8_t *placed, yy_stream_t stream);

/* YinYangGame.getGameEnded(board, player)  (yin_yang_game.py:80-110): out float64 [G] holding
 * exactly the reference's values 0, +1, -1, 0.0001.  counts: int32 [G,2] = count_pieces()
 * (yin_yang_logic.py:130-134), may be NULL. */
int yy_rules_game_ended(const int8_t *boards, const int8_t *players, int G, int R, int C,
                        uint32_t flags, double *out, int32_t *counts, yy_stream_t stream);

/* YinYangNeuralNetwork.board_to_input(board)  (ai/neural_network.py:156-196):
 * out float32 [G,5,R,C] = [empty, black, white, row fill, column fill]. */
int yy_encode_planes(const int8_t *boards, int G, int R, int C, float *out, yy_stream_t stream);

/* Native packed form of the same rules: one game per lane on bitboards.
 * black/white: uint64 [NW][G] (word-major SoA, NW = ceil(R*C/64), bit a = cell a);
 * mask_p1/mask_m1: uint64 [NW][G] legal masks for +1 / -1; result: int8 [G] game-ended code from
 * BLACK's (+1) point of view: 0 ongoing, +1, -1, 2 = draw.  Any output may be NULL. */
int yy_rules_mask_terminal_bb(const uint64_t *black, const uint64_t *white, int G, int R, int C,
                              uint32_t flags, uint64_t *mask_p1, uint64_t *mask_m1,
                              int8_t *result, yy_stream_t stream);

/* int8 boards <-> packed bitboards (layout as above). */
int yy_pack_boards(const int8_t *boards, int G, int R, int C, uint64_t *black, uint64_t *white,
                   yy_stream_t stream);
int yy_unpack_boards(const uint64_t *black, const uint64_t *white, int G, int R, int C,
                     int8_t *boards, yy_stream_t stream);

/* ------------------------------------------------------------------ batched MCTS context
 * One context = G concurrent games searched in lockstep; replaces Node + MCTS.search/_simulate
 * (ai/mcts.py:28-225, 275-414).  Per lockstep step:
 *     yy_mcts_select        -> leaf planes for the evaluator   (mcts.py:356-362, 385-397)
 *     <policy/value CNN on planes, caller's business>
 *     yy_mcts_expand_backup <- policy, value                   (mcts.py:50-91, 406-412)
 * or the fused yy_mcts_step (= expand_backup of the previous leaves + select of the next). */

typedef struct yy_mcts_config {
    int32_t G;                 /* concurrent games */
    int32_t R, C;              /* board */
    int32_t max_sims;          /* largest num_simulations a search will run (sizes the arenas) */
    float cpuct;               /* mcts.py:231 cpuct */
    uint32_t flags;            /* YY_FLAG_* */
    int64_t edges_per_game;    /* 0 = worst case (max_sims+2)*A */
    int64_t nodes_per_game;    /* 0 = max_sims+2 */
} yy_mcts_config;

typedef struct yy_mcts yy_mcts; /* opaque */

/* sync. Allocates the arenas in HBM on the current device. */
int yy_mcts_create(const yy_mcts_config *cfg, yy_mcts **out);
int yy_mcts_destroy(yy_mcts *ctx);
/* bytes of HBM held by the context (host out) */
int yy_mcts_memory_bytes(const yy_mcts *ctx, uint64_t *out);

/* MCTS.search prologue (mcts.py:288-295): fresh root per game from boards int8 [G,R,C] and
 * root_players int8 [G]; active uint8 [G] (NULL = all active; inactive games are skipped by every
 * later call).  Writes the root planes float32 [G,5,R,C] for evaluator call #0. */
int yy_mcts_begin(yy_mcts *ctx, const int8_t *boards, const int8_t *root_players,
                  const uint8_t *active, float *planes_out, yy_stream_t stream);

/* Root expansion (mcts.py:297-317).  policy float32 [G,A] = softmax output of call #0 (the value
 * is discarded by the reference).  noise: NULL, or float64 [G,A] holding each game's Dirichlet
 * draw scattered to its legal action indices; priors become
 * f32( f64(f32(1-eps) * p) + eps*noise )  exactly as mcts.py:310-312 evaluates under numpy 2.
 * A game whose noise row is all zero keeps its raw priors (add_exploration_noise=False). */
int yy_mcts_expand_root(yy_mcts *ctx, const float *policy, const double *noise, double eps,
                        yy_stream_t stream);

/* Selection + leaf state (mcts.py:356-399 up to the predict call): PUCT descent in float32
 * (mcts.py:97-145), place-if-legal, terminal test and legal mask of the new position, and the 5
 * input planes written to row g of planes_out float32 [G,5,R,C].  needs_eval uint8 [G] = 1 when
 * row g must be evaluated (0: terminal leaf revisited, or inactive). */
int yy_mcts_select(yy_mcts *ctx, float *planes_out, uint8_t *needs_eval, yy_stream_t stream);

/* Expansion + backup (mcts.py:50-91, 147-156, 406-412) for the leaves chosen by the last select.
 * policy float32 [G,A], value float32 [G] (rows with needs_eval == 0 are ignored). */
int yy_mcts_expand_backup(yy_mcts *ctx, const float *policy, const float *value,
                          yy_stream_t stream);

/* Fused: expand_backup(policy, value) then select(planes_out, needs_eval) in ONE launch. */
int yy_mcts_step(yy_mcts *ctx, const float *policy, const float *value, float *planes_out,
                 uint8_t *needs_eval, yy_stream_t stream);

/* Root child visit counts int32 [G,A] (Node.get_children_visit_counts, mcts.py:168-181);
 * optional per-child value_sum float32 [G,A] and prior float32 [G,A] (NULL to skip). */
int yy_mcts_root_counts(yy_mcts *ctx, int32_t *counts, float *child_w, float *child_p,
                        yy_stream_t stream);

/* Node.get_children_distribution(T) for T == 1 (counts/sum, uniform 1/A when all zero) or T == 0
 * (uniform over the arg-max set) in float64 [G,A]  (mcts.py:183-215). */
int yy_mcts_root_policy(yy_mcts *ctx, int temperature_is_zero, double *pi, yy_stream_t stream);

/* root.visits int32 [G] and root.value_sum float64 [G] (mcts.py:39-40). */
int yy_mcts_root_stats(yy_mcts *ctx, int32_t *visits, double *value_sum, yy_stream_t stream);

/* The board each game's root refers to after the search, int8 [G,R,C]: unchanged in copied mode,
 * the mutated caller board in aliased mode. */
int yy_mcts_get_boards(yy_mcts *ctx, int8_t *boards, yy_stream_t stream);

/* sync. Host outputs: number of games that failed in ANY search since the previous status call -- arena overflow,
 * NaN priors or a NaN value from the evaluator; such a game stops searching, and the flag is sticky: yy_mcts_begin
 * does not clear it, only this call does -- and
 * counters[8] = {evaluator rows requested, selection levels walked, children scanned during
 * selection, children created, terminal revisits, nodes created, pass values reused, evaluation-cache hits} accumulated since create
 * or the last yy_mcts_reset_counters.  Returns YY_E_ARENA if any game overflowed. */
int yy_mcts_status(yy_mcts *ctx, int32_t *n_overflow, uint64_t *counters);
int yy_mcts_reset_counters(yy_mcts *ctx, yy_stream_t stream);
/* Shared book of pre-evaluated positions.  Every self-play game starts from the empty board, so the searches of the first plies
 * of ALL games walk the same few hundred thousand positions (8x8: 770 232 positions with <= 8 stones are reachable).  The
 * caller enumerates them, evaluates them once in large batches with the SAME evaluator the searches use, and hands the table
 * to the contexts: a leaf with at most `max_stones` stones is looked up there first (same hash / probe / full key compare as
 * the per-game cache) and, on a hit, needs no evaluator row (counters[7]).  Read-only during play, so no atomics on the
 * lookup side; arrays stay owned by the caller and must outlive the context (or be unset with meta = NULL).
 *   yy_book_insert: keys u64 [n, 2*NW] (black words, white words), distinct -> slot_of i32 [n] (-1 = no free slot within
 *   the probe window); fills meta u32 [cap] (zeroed by the caller, cap a power of two) and table_keys u64 [cap, 2*NW].
 *   The caller then scatters its value f32 [cap] and policy f32 [cap, A] rows by slot_of. */
int yy_book_insert(const uint64_t *keys, int n, int R, int C, uint32_t *meta, uint64_t *table_keys, int64_t cap,
                   int32_t *slot_of, yy_stream_t stream);
int yy_mcts_set_book(yy_mcts *ctx, const uint32_t *meta, const uint64_t *keys, const float *value, const float *policy,
                     int64_t cap, int max_stones);
/* Forget every cached evaluation (YY_FLAG_REUSE_TRANSPOSITIONS / YY_FLAG_KEEP_EVALUATIONS): to be called when the evaluator
 * (the network) changes between two searches of one context.  Async on `stream`. */
int yy_mcts_cache_clear(yy_mcts *ctx, yy_stream_t stream);

/* ------------------------------------------------------------------ evaluator epilogue
 * Batched leaf evaluator fast path (the reference evaluates one board per call:
 * ai/neural_network.py:94-123, 125-154).  In-place fused epilogue of one convolution of the tower
 * on a channels-last bf16 activation tensor x [rows, C] (rows = G*R*Cb cells):
 *     x = relu?( x + bias[c] (+ residual) )         bias float32 [C]; residual bf16 [rows, C] or NULL
 * (eval-mode BatchNorm is folded into the convolution weights and this bias on the host). */
int yy_nn_bias_act_bf16(void *x, const float *bias, const void *residual, int64_t rows, int C,
                        int relu, yy_stream_t stream);

/* The residual tower of the evaluator (stem + residual blocks, ai/neural_network.py:16-33, 105-110)
 * as ONE LDS-resident MFMA kernel: planes float32 [G,5,R,R] -> out bf16 [G,R,R,128] (channels-last
 * activations after the last block).  weights: bf16 chunks in fragment order and bias float32
 * [n_layers,128], both produced on the host from the module with eval-mode BatchNorm folded
 * (network.pack_tower).  n_layers = 1 + 2*res_blocks <= 21.  Boards 6x6, 8x8 or 12x12 (planes
 * [G,5,R,R], out [G,R,R,128]) and 128 channels; anything else returns YY_E_UNSUPPORTED and the
 * caller uses library convolutions + yy_nn_bias_act_bf16.  The launch picks the workgroup shape from G (8x8: one, two
 * or four boards per workgroup, so that small batches still spread over the chip); a board's output bits do not depend on G. */
int yy_nn_tower_bf16(const float *planes, const void *weights, const float *bias, void *out, int G,
                     int R, int C, int channels, int n_layers, yy_stream_t stream);

/* Same kernel, with the policy_conv / value_conv 1x1 head convolutions + BatchNorm + ReLU
 * (neural_network.py:113, 118) fused behind the tower: out_heads bf16 [G,2,32,R*R] = [policy features,
 * value features] in the reference's NCHW flatten order (channel*R*R + cell, :114 / :119), ready for
 * policy_fc / value_fc1.  weights holds one extra 16 KB chunk ([ks 8][nt 2][h 2][c 32][j 8]) and bias
 * one extra row [n_layers] = [policy bias 32 | value bias 32 | 0...]. */
int yy_nn_tower_heads_bf16(const float *planes, const void *weights, const float *bias,
                           void *out_heads, int G, int R, int C, int channels, int n_layers,
                           yy_stream_t stream);

/* The residual tower in EXACT float32 on the f32-input MFMA (v_mfma_f32_32x32x2_f32; a k-ordered fmaf chain,
 * bitwise f32): the fast form of the fp32 parity evaluator.  planes float32 [G,5,8,8] -> out float32 [G,8,8,128]
 * (channels-last activations after the last block); weights float32 chunks in fragment order and bias float32
 * [n_layers,128] from network.pack_tower_f32.  8x8 boards, 128 channels. */
int yy_nn_tower_f32(const float *planes, const void *weights, const float *bias, float *out, int G,
                    int R, int C, int channels, int n_layers, yy_stream_t stream);

/* Head finish (neural_network.py:115, 120-121, 152): h bf16 [G, A+H] = policy logits then value_fc1
 * outputs (bias added); policy float32 [G,A] = softmax(logits); value float32 [G] =
 * tanh(relu(hidden) . w2 + b2) with w2 float32 [H], b2 float32 [1]. */
int yy_nn_head_finish_bf16(const void *h, int G, int A, int H, const float *w2, const float *b2,
                           float *policy, float *value, yy_stream_t stream);

/* The residual tower at FLOAT32 accuracy on the F16 matrix cores ("split-f16"): activations and weights are held as
 * hi = f16(x), lo = f16(x - hi) (22 significant bits); per output element one f32 accumulator takes w_hi*x_hi and a second
 * one w_lo*x_hi + w_hi*x_lo; bias / residual / ReLU in f32.  To keep the lo parts in float16's normal range the weights are
 * stored times 2^weight_exp and the activations (and the bias rows of the tower) live times 2^act_exp -- exact scalings chosen
 * by the packers in network.py.  Agrees with a float64 evaluation to ~5e-7 of scale, like the float32 module itself.
 * Replaces ai/neural_network.py:94-119 (float32 on the CPU in the reference).
 *
 * yy_nn_tower_f16x3_regs = the round-2 form (csrc/yy_tower_h3r.hip: v_mfma_f32_32x32x16_f16, boards 6x6 / 8x8 / 12x12,
 * 128 channels), kept as the A/B partner of yy_nn_tower_g below: weights / head_w in the wave-major order of
 * network.pack_tower_h3r / pack_heads_h3r ([9 + 36*(n_layers-1)][8192] / [2][8192] f16); bias float32 [n_layers (+1), 128];
 * exactly one of out (float32 [G,R,R,128] tower activations) and out_heads (float32 [G,2,32,R*R] = [policy features, value
 * features] in the reference's NCHW flatten order, neural_network.py:114 / :119; needs head_w) is non-NULL; rows / n_rows
 * (device pointers, or both NULL): evaluate planes[rows[i]] for i < *n_rows into dense row i, workgroups past *n_rows exit. */
int yy_nn_tower_f16x3_regs(const float *planes, const void *weights, const void *head_w, const float *bias,
                           float *out, float *out_heads, const int32_t *rows, const int32_t *n_rows, int G,
                           int R, int C, int channels, int n_layers, int weight_exp, int head_exp,
                           int act_exp, yy_stream_t stream);

/* The GENERAL form of the float32-accurate tower (csrc/yy_tower_g.hip; the form the engine uses): any R x C board with at
 * most 144 cells (train_alphazero.py:35-36 takes any --rows / --cols), 32 / 64 / 96 / 128 channels and up to 10 residual
 * blocks (ai/neural_network.py:39), on v_mfma_f32_16x16x32_f16: a workgroup of channels/32 waves evaluates `boards` boards =
 * nb blocks of 16 (board, cell) columns (boards * R * C <= 16 * nb; nb one of yy_nn_tower_g_forms).  Same split-f16 numerics
 * and the same I/O conventions as yy_nn_tower_f16x3_regs: weights = f16 chunks [9 + 9*(channels/32)*(n_layers-1)]
 * [channels*64] in MFMA operand order (network.pack_tower_g) times 2^weight_exp; bias float32 [n_layers, channels] times
 * 2^act_exp; exactly one of out (float32 [G,R,C,channels]) and out_heads (float32 [G,2,32,R*C]: needs head_w = f16
 * [4][channels/32][2][512] times 2^head_exp and head_bias float32 [64], network.pack_heads_g) is non-NULL; rows / n_rows as
 * above.  The launch only computes when gate_lo < live rows <= gate_hi (-1, INT_MAX: always), so that two forms can be
 * enqueued and the one fitting the device-side row count runs.  A row's results do not depend on nb, boards or its position
 * in the batch (same accumulation order per output element in every form).  Replaces ai/neural_network.py:94-119. */
int yy_nn_tower_g(const float *planes, const void *weights, const void *head_w, const float *bias, const float *head_bias,
                  float *out, float *out_heads, const int32_t *rows, const int32_t *n_rows, int G, int R, int C,
                  int channels, int n_layers, int weight_exp, int head_exp, int act_exp, int nb, int boards, int gate_lo,
                  int gate_hi, yy_stream_t stream);
/* the nb values yy_nn_tower_g is built for at this channel count: writes up to 8 ints, returns how many (0: unsupported) */
int yy_nn_tower_g_forms(int channels, int *nb_out);

/* policy_fc and value_fc1 (neural_network.py:115, :120; float32 on the CPU in the reference) as one split-f16 GEMM kernel of
 * our own (csrc/yy_fc_heads.hip), on the dense feature rows a tower launch wrote: feats float32 [G,2,K] (K = 32*R*C) ->
 * logits float32 [G,A] = policy_fc(feats[:,0]), hidden float32 [G,H] = value_fc1(feats[:,1]), bias added.  wpk / bias / jobs
 * from network.pack_fc_heads: jobs int32 [n_jobs][4] = (head, first output, outputs <= 64, first 8 KB weight block), weights f16
 * [n_jobs][ceil(K/128)*4][4][2][64][8] times 2^weight_exp; features are split as x * 2^act_exp = hi + lo.  n_rows (device, or
 * NULL = G): rows past *n_rows are neither read nor written.  Every output element is one fixed k-ascending MFMA chain (no
 * split-K, no batch-size dependent tiling): a row's outputs do not depend on G, *n_rows or its position in the batch, which
 * is what lets the search reuse evaluations across launches of different sizes (YY_FLAG_REUSE_*). */
int yy_nn_fc_heads_f16x3(const float *feats, const void *wpk, const float *bias, const int32_t *jobs, int n_jobs,
                         float *logits, float *hidden, const int32_t *n_rows, int G, int K, int A, int H,
                         int weight_exp, int act_exp, yy_stream_t stream);

/* float32 head finish (neural_network.py:115, 120-121, 152): logits float32 [G,A] (policy_fc output, bias added), hidden
 * float32 [G,H] (value_fc1 output, bias added) of dense row i -> policy[g] = softmax(logits[i]),
 * value[g] = tanh(relu(hidden[i]) . w2 + b2), g = rows ? rows[i] : i, for i < (n_rows ? *n_rows : G). */
int yy_nn_head_finish_f32(const float *logits, const float *hidden, int G, int A, int H, const float *w2,
                          const float *b2, const int32_t *rows, const int32_t *n_rows, float *policy,
                          float *value, yy_stream_t stream);

/* Leaf-batch compaction for the lockstep step: rows[0 .. *n) = ascending indices g < G with flags[g] != 0 (the games whose
 * selected leaf needs an evaluation: yy_mcts_select / yy_mcts_step write these flags; a terminal revisit does not,
 * mcts.py:365-366).  All pointers are device pointers. */
int yy_compact_rows(const uint8_t *flags, int G, int32_t *rows, int32_t *n, yy_stream_t stream);

/* ------------------------------------------------------------------ episode-loop random draws
 * Counter-based (Philox4x32-10) replacements of the two draws the reference takes from numpy's global stream, keyed by
 * (seed, GLOBAL game index, ply, purpose, element) so that a game's transcript does not depend on its slot, on the batch
 * size, on refill order or on the number of ranks.  All pointers are device pointers.
 *
 * yy_selfplay_root_noise: np.random.dirichlet([alpha] * k) over the k legal moves (ai/mcts.py:298-312), for the games with
 * draw[g] != 0: noise float64 [G,A], zero rows elsewhere (yy_mcts_expand_root treats an all-zero row as "no noise"). */
int yy_selfplay_root_noise(uint64_t seed, const int64_t *game_id, const int32_t *ply, const uint8_t *draw,
                           const uint8_t *mask, int G, int A, double alpha, double *noise, yy_stream_t stream);

/* yy_selfplay_sample_actions: the move choice of SelfPlayWorker.play_game (ai/self_play.py:143-160) for the games with
 * searching[g] != 0 (action -1 otherwise): ply < temperature_threshold -> sample from pi * mask renormalised (uniform over
 * the legal moves when that sum is 0); else a uniformly random member of argmax(pi).  pi float64 [G,A], mask uint8 [G,A]. */
int yy_selfplay_sample_actions(uint64_t seed, const int64_t *game_id, const int32_t *ply,
                               const uint8_t *searching, const double *pi, const uint8_t *mask, int G, int A,
                               int temperature_threshold, int32_t *action, yy_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* YY_ENGINE_H */

// yy_tower_h3.hip -- the FLOAT32-ACCURATE evaluator tower on the F16 matrix cores ("split-f16", evaluator mode "f16x3"):
// the numerics shared by the two kernels and the C-ABI entry points that dispatch to them.
//
// Reference computation: YinYangNeuralNetwork.forward, stem + residual blocks + the two 1x1 head convolutions
// (src/yin_yang/ai/neural_network.py:16-33, 94-119) in float32 on the CPU.  gfx950's f32-input MFMA runs at 1/16 of the
// f16 rate, so every activation x and every weight w is held as TWO float16 numbers
//     hi = f16(x),  lo = f16(x - hi)          (x == hi + lo to 22 significant bits; f32 has 24)
// and each product is formed as  w_hi*x_hi  (accumulator 1)  +  w_lo*x_hi + w_hi*x_lo  (accumulator 2) with three
// v_mfma_f32_32x32x16_f16; the dropped w_lo*x_lo term is 2^-22 relative.  (Two accumulators because every MFMA rounds its
// accumulator once: one shared accumulator rounds the large sum three times per k-step and measured 1.7x the error.)  For lo to keep its 11 bits it must stay in float16's NORMAL
// range (|lo| >= 2^-14, i.e. |x| >~ 0.06): weights of a trained or random-init network are ~0.02, so
//   * weights are stored times 2^kw (kw chosen on the host so that max |w| * 2^kw lies in [2^13, 2^14); an exact scaling),
//   * activations -- and with them the bias table and the residual -- live times 2^ka (ka = 3),
// and the epilogue is v' = (acc1 + acc2) * 2^-kw + bias' (+ residual'), ReLU, split again.  (Round 2's first version scaled
// the lo parts alone by 2^11 instead, which costs a multiply per output element in every epilogue.)  The MFMA keeps subnormal float16
// inputs (measured: tools/mfma_f16_denorm.hip), so a tiny activation loses relative, not absolute, accuracy.
// Against a float64 evaluation of the same folded network the activations agree to ~4e-7 of scale -- the float32 module
// itself is at ~3e-7 -- where the split-bf16 kernel (yy_tower_x3.hip: 16 significant bits) is at 9e-6
// (tests/test_gpu_network.py::test_split_f16_tower_kernel_is_float32_grade); searches driven by it return the reference's own
// visit counts on all 64 recorded 800-simulation roots (tests/test_gpu_mcts.py::test_live_gpu_evaluator_search_vs_reference_pi).
//
// Kernels: yy_tower_h3r.hip (weight stream in registers: the shipped form for every board size), yy_tower_h3q.hip (weights
// through wave-private LDS rings: 8x8 batches of <= 256 boards run it with one board per workgroup; for 6x6 / 12x12 it is the
// A/B partner).  Both accumulate every output element in the same order: identical bits.
// Optional row gather: with (rows, n_rows) a launch evaluates planes[rows[i]] for i < *n_rows into dense output row i and
// workgroups past *n_rows exit at once -- the lockstep step compacts the leaves that need an evaluation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/yy_engine.h"

extern "C" int yy_tower_set_err(int code, const char *msg);
extern "C" int yy_tower_h3q_launch(const float *planes, const void *weights, const float *bias, float *out, float *out_heads,
                                   const int *rows, const int *n_rows, int G, int R, int n_layers, const float *scales,
                                   yy_stream_t s);   // yy_tower_h3q.hip

static int launch_h3(const float *planes, const void *weights, const float *bias, float *out, float *out_heads, const int *rows,
                     const int *n_rows, int G, int R, int C, int channels, int n_layers, int weight_exp, int head_exp, int act_exp,
                     yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights || !bias || (!out && !out_heads) || G < 0 || (rows && !n_rows))
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_f16x3: bad argument");
    if (R != C || (R != 6 && R != 8 && R != 12) || channels != 128 || n_layers < 1 || n_layers > 21 || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED,
                                "yy_nn_tower_f16x3: needs 6x6, 8x8 or 12x12 boards, 128 channels, at most 10 residual blocks");
    // {input scale 2^ka, accumulator scale 2^-kw, head accumulator scale 2^-(kh+ka), output scale 2^-ka}
    const float sc[4] = {ldexpf(1.0f, act_exp), ldexpf(1.0f, -weight_exp), ldexpf(1.0f, -(head_exp + act_exp)), ldexpf(1.0f, -act_exp)};
    return yy_tower_h3q_launch(planes, weights, bias, out, out_heads, rows, n_rows, G, R, n_layers, sc, s);
}

// weights: f16 chunks [9 + 36*(n_layers-1)][8192] in fragment order (network.pack_tower_h3), times 2^weight_exp; bias f32
// [n_layers,128] times 2^act_exp; planes f32 [G,5,R,R]; out f32 [G,R,R,128].
extern "C" int yy_nn_tower_f16x3(const float *planes, const void *weights, const float *bias, float *out, int G, int R, int C,
                                 int channels, int n_layers, int weight_exp, int act_exp, yy_stream_t s) {
    return launch_h3(planes, weights, bias, out, nullptr, nullptr, nullptr, G, R, C, channels, n_layers, weight_exp, 0, act_exp, s);
}

// the same + the two 1x1 head convolutions: weights hold two more chunks (times 2^head_exp), bias one more row (unscaled);
// out_heads f32 [G,2,32,R*R]; rows / n_rows (device, or both NULL): evaluate planes[rows[i]] for i < *n_rows into row i.
extern "C" int yy_nn_tower_heads_f16x3(const float *planes, const void *weights, const float *bias, float *out_heads,
                                       const int32_t *rows, const int32_t *n_rows, int G, int R, int C, int channels,
                                       int n_layers, int weight_exp, int head_exp, int act_exp, yy_stream_t s) {
    return launch_h3(planes, weights, bias, nullptr, out_heads, rows, n_rows, G, R, C, channels, n_layers, weight_exp, head_exp,
                     act_exp, s);
}

// Small live batches (8x8).  With leaf-row compaction the number of rows a launch really computes is only known on the device,
// and below ~320 rows the one-board-per-workgroup LDS-ring form is the faster one (232 us against 312 us for <= 128 rows: twice
// as many CUs at work), above it the two-board register-ring form.  Both are launched with a gate on *n_rows -- the form that
// is not wanted exits at once -- and write the same bits, so the choice never shows in the results.
//   weights_lds = network.pack_tower_h3 (+ head chunks), weights_regs / head_w_regs = pack_tower_h3r / pack_heads_h3r.
extern "C" int yy_tower_h3q_launch81_gated(const float *planes, const void *weights, const float *bias, float *out_heads,
                                           const int *rows, const int *n_rows, int G, int n_layers, const float *sc,
                                           int gate_lo, int gate_hi, yy_stream_t s);
extern "C" int yy_tower_h3r_launch8_gated(const float *planes, const void *weights, const void *head_w, const float *bias,
                                          float *out_heads, const int *rows, const int *n_rows, int G, int n_layers,
                                          const float *sc, int gate_lo, int gate_hi, yy_stream_t s);

extern "C" int yy_nn_tower_heads_f16x3_auto(const float *planes, const void *weights_lds, const void *weights_regs,
                                            const void *head_w_regs, const float *bias, float *out_heads, const int32_t *rows,
                                            const int32_t *n_rows, int G, int R, int C, int channels, int n_layers,
                                            int weight_exp, int head_exp, int act_exp, int split, yy_stream_t s) {
    if (G == 0) return YY_OK;
    if (!planes || !weights_lds || !weights_regs || !head_w_regs || !bias || !out_heads || !rows || !n_rows || G < 0 || split < 0)
        return yy_tower_set_err(YY_E_INVALID, "yy_nn_tower_heads_f16x3_auto: bad argument");
    if (R != 8 || C != 8 || channels != 128 || n_layers < 1 || n_layers > 21 || (n_layers & 1) == 0)
        return yy_tower_set_err(YY_E_UNSUPPORTED, "yy_nn_tower_heads_f16x3_auto: 8x8 boards, 128 channels, at most 10 residual blocks");
    const float sc[4] = {ldexpf(1.0f, act_exp), ldexpf(1.0f, -weight_exp), ldexpf(1.0f, -(head_exp + act_exp)), ldexpf(1.0f, -act_exp)};
    if (int e = yy_tower_h3q_launch81_gated(planes, weights_lds, bias, out_heads, rows, n_rows, G, n_layers, sc, -1, split, s)) return e;
    return yy_tower_h3r_launch8_gated(planes, weights_regs, head_w_regs, bias, out_heads, rows, n_rows, G, n_layers, sc, split,
                                      0x7FFFFFFF, s);
}

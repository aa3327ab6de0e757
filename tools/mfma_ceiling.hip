// Bare f16 MFMA loops on random register operands: what this board sustains for dense f16 MFMA issue, per shape, with no
// memory traffic at all (the ceiling the tower kernels are priced against beside the nominal 2.5 PFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_ceiling tools/mfma_ceiling.hip && tools/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE> __global__ void __launch_bounds__(256, 1) k_loop(const f16x8 *in, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = in[(t * 8 + i) & 0xFFFF]; b[i] = in[(t * 8 + 4 + i) & 0xFFFF]; }
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
        out[t] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][7];
        out[t] = s;
    }
}

int main() {
    const int WG = 256 * 8, iters = 20000;          // 8 workgroups of 4 waves per CU queued; one resident per CU at a time
    f16x8 *in; float *out;
    hipMalloc(&in, 65536 * sizeof(f16x8)); hipMalloc(&out, WG * 256 * sizeof(float));
    _Float16 *h = (_Float16 *)malloc(65536 * 16);
    srand(1);
    for (int i = 0; i < 65536 * 8; i++) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++)
        for (int shape = 0; shape < 2; shape++) {
            hipEventRecord(e0);
            if (shape == 0) k_loop<16><<<WG, 256>>>(in, out, iters); else k_loop<32><<<WG, 256>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave and iteration: 16 MFMAs x 16384 FLOP (16x16x32) or 8 MFMAs x 32768 FLOP (32x32x16) = 262144 FLOP
            const double fl = (double)WG * 4 * iters * 262144.0;
            printf("{\"shape\": \"%s\", \"ms\": %.2f, \"tflops\": %.1f, \"frac_of_2500\": %.3f}\n", shape == 0 ? "16x16x32_f16" : "32x32x16_f16", ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
        }
    return 0;
}

// Does v_mfma_f32_32x32x16_f16 keep SUBNORMAL f16 inputs (hipcc's default kernel mode)?  One wave: A[0][0] = 2^-20 (a
// subnormal f16), B[0][*] = 1024 -> D[0][*] must be 2^-10 if subnormals are kept, 0 if they are flushed.  Also checks
// v_cvt_pk_f16_f32 producing a subnormal and rounding to nearest even.   hipcc --offload-arch=gfx950 -o /tmp/t tools/mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    const f32x2 tiny = {9.5367431640625e-07f, 3.0001e-5f};       // 2^-20 and a value that becomes a subnormal f16
    const f16x2 t = __builtin_convertvector(tiny, f16x2);
    if (lane == 0) a[0] = t[0];                                  // A[row 0][k 0]
    if (lane < 32) b[0] = (_Float16)1024.0f;                     // B[k 0][col lane]
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) {
        out[0] = c[0];                  // D[row 0][col 0]
        out[1] = (float)t[0];
        out[2] = (float)t[1];
        const f32x2 rne = {1.00048828125f, 1.00146484375f};       // halfway cases: 1 + 2^-11 -> 1.0 (even), 1 + 3*2^-11 -> 1 + 2^-9
        const f16x2 r = __builtin_convertvector(rne, f16x2);
        out[3] = (float)r[0];
        out[4] = (float)r[1];
    }
}
int main() {
    float *d, h[5];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mfma(2^-20 * 1024) = %.10g (kept: 0.0009765625)\ncvt(2^-20) = %.10g  cvt(3.0001e-5) = %.10g (subnormal f16: 3.001093864e-05)\n"
           "rne(1+2^-11) = %.10g (1)  rne(1+3*2^-11) = %.10g (1.001953125)\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}

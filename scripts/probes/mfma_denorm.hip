// Probe: does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs on gfx950, and does the fp32 accumulator
// keep small addends?  Prints the result of  sum_k a[k]*b[k]  for a = 2^-20 (f16 subnormal), b = 1024.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float av, float bv, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8); float h[2];
    const float tests[][2] = {{9.5367431640625e-07f, 1024.f}, {5.9604644775390625e-08f, 1024.f}, {3.0517578125e-05f, 1.f}, {1.f, 1.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%.10g (as f16 %.10g) b=%g : mfma sum over k=16 -> %.10g   expected %.10g\n", t[0], h[1], t[1], h[0], 16.0 * h[1] * t[1]);
    }
    return 0;
}

// Probe: does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs on gfx950 (A operand, B operand), and does the
// in-kernel split x -> (hi, lo = f16(x - hi)) keep a subnormal lo?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float av, float bv, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
__global__ void split_probe(float x, float* out) {
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = lo; b[i] = (_Float16)1024.f; }
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);       // lo as the B operand
    if (threadIdx.x == 0) { out[0] = (float)hi; out[1] = (float)lo; out[2] = acc[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 16); float h[3];
    const float tests[][2] = {{9.5367431640625e-07f, 1024.f}, {1024.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1024.f},
                              {1024.f, 5.9604644775390625e-08f}, {3.0517578125e-05f, 1.f}, {1.f, 1.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("A=%.10g B=%.10g : mfma sum over k=16 -> %.10g   expected %.10g\n", h[1], h[2], h[0], 16.0 * h[1] * h[2]);
    }
    for (float x : {0.1f, 0.0123f, 1.0003f}) {
        hipLaunchKernelGGL(split_probe, dim3(1), dim3(64), 0, 0, x, d);
        (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("split %.9g -> hi %.9g lo %.9g (x - hi = %.9g); mfma(1024, lo) k=16 -> %.9g expected %.9g\n", x, h[0], h[1], x - h[0], h[2],
               16.0 * 1024.0 * h[1]);
    }
    return 0;
}

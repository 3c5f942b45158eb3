// Probe: the 3-instruction (hi, lo) split  v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16  against the C++ split, incl. f16-subnormal p.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float* in, float* out, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n + 1) return;
    float p0 = in[2 * i], p1 = in[2 * i + 1];
    unsigned hp, lp;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(p0), "v"(p1));
    asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(p0), "v"(hp));
    asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(p1), "v"(hp));
    _Float16 h0 = __builtin_bit_cast(_Float16, (unsigned short)(hp & 0xffff)), h1 = __builtin_bit_cast(_Float16, (unsigned short)(hp >> 16));
    _Float16 l0 = __builtin_bit_cast(_Float16, (unsigned short)(lp & 0xffff)), l1 = __builtin_bit_cast(_Float16, (unsigned short)(lp >> 16));
    _Float16 ch0 = (_Float16)p0, ch1 = (_Float16)p1;
    _Float16 cl0 = (_Float16)(p0 - (float)ch0), cl1 = (_Float16)(p1 - (float)ch1);
    float* o = out + 16 * i;
    o[0] = p0; o[1] = (float)h0; o[2] = (float)l0; o[3] = (float)ch0; o[4] = (float)cl0;
    o[8] = p1; o[9] = (float)h1; o[10] = (float)l1; o[11] = (float)ch1; o[12] = (float)cl1;
}
int main() {
    const float h_in[] = {1.0003f, 0.7312345f, 3.1e-5f, 2.9e-5f, 6.2e-5f, 1.7e-6f, 5.0e-8f, 2.0e-8f, 1234.567f, 0.12345678f, 6.0e-5f, 6.11e-5f};
    const int n = sizeof(h_in) / 4;
    float *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h_in)); (void)hipMalloc(&d_out, 16 * 4 * (n / 2));
    (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, n);
    float h_out[16 * 8]; (void)hipMemcpy(h_out, d_out, 16 * 4 * (n / 2), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const float* o = h_out + 16 * (i / 2) + 8 * (i & 1);
        printf("p=%-14.9g asm: hi=%-14.9g lo=%-14.9g (sum err %.3g) | c++: hi=%-14.9g lo=%-14.9g (sum err %.3g)\n", o[0], o[1], o[2],
               (double)o[1] + o[2] - o[0], o[3], o[4], (double)o[3] + o[4] - o[0]);
    }
    return 0;
}

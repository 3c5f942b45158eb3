// Probe: the SUSTAINED v_mfma_f32_32x32x16_f16 rate of the whole chip as a function of the operand DATA and of the number of busy CUs.
// One 512-thread workgroup per CU (2 waves per SIMD, like the 256-tile GEMM); every wave issues `iters` x 48 MFMAs over 8 accumulator
// chains from 4 + 4 operand registers; nothing else in the loop.  Operands: zeros | N(0,1) halves | (hi, lo) pairs as the split-f16
// kernels feed them (half of the passes multiply a "lo" operand 2^-11 the size).  Reports the shader clock (s_memtime ticks against the
// 100 MHz s_memrealtime), ticks per MFMA per SIMD and the chip-wide TFLOP/s: the matrix pipe's peak is only reachable with quiet data.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(const f16x8* in, float* out, unsigned long long* res, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(wave * 8 + i) * 64 + lane]; b[i] = in[(wave * 8 + 4 + i) * 64 + lane]; }
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 48; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i >> 1) & 3], b[(i >> 3) & 3], acc[i & 7], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { res[2 * blockIdx.x] = t1 - t0; res[2 * blockIdx.x + 1] = r1 - r0; }
}

static float gauss() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

int main() {
    const int iters = 4000, maxb = 256;
    f16x8* in; float* out; unsigned long long* res;
    CHECK(hipMalloc(&in, 64 * 64 * sizeof(f16x8))); CHECK(hipMalloc(&out, maxb * 512 * 4)); CHECK(hipMalloc(&res, maxb * 16));
    const char* names[4] = {"zeros", "N(0,1) halves", "split-f16 pairs (hi ~ N(0,1), lo ~ 2^-11 hi)", "small weights x activations (|w| ~ 0.05 * 256, x ~ 3)"};
    for (int pat = 0; pat < 4; ++pat) {
        std::vector<_Float16> h(64 * 64 * 8);
        for (size_t i = 0; i < h.size(); ++i) {
            const int reg = (int)(i / (64 * 8)) % 8;            // registers 0-3 = a, 4-7 = b of a wave
            float v = 0.f;
            if (pat == 1) v = gauss();
            if (pat == 2) v = (reg & 1) ? gauss() * 4.8828125e-4f : gauss();                 // odd registers carry a "lo" operand
            if (pat == 3) v = reg < 4 ? ((reg & 1) ? gauss() * 0.05f * 256.f * 4.8828125e-4f : gauss() * 0.05f * 256.f)
                                      : ((reg & 1) ? gauss() * 3.f * 4.8828125e-4f : gauss() * 3.f);
            h[i] = (_Float16)v;
        }
        CHECK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int blocks : {8, 64, 128, 256}) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, in, out, res, iters); CHECK(hipDeviceSynchronize()); }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, in, out, res, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> r(2 * blocks); CHECK(hipMemcpy(r.data(), res, r.size() * 8, hipMemcpyDeviceToHost));
            double ticks = 0, real = 0; for (int bIdx = 0; bIdx < blocks; ++bIdx) { ticks += r[2 * bIdx]; real += r[2 * bIdx + 1]; }
            ticks /= blocks; real /= blocks;
            const double n = 48.0 * iters;                       // MFMAs per wave; 2 waves share a SIMD
            const double flops = (double)blocks * 8 * n * 32768.0;
            printf("%-58s %3d CUs: clock %.3f GHz, %6.2f ticks per MFMA per SIMD, kernel %8.1f us -> %7.1f TFLOP/s (%.0f %% of %d CUs x 4096 flop/clk x 2.4 GHz)\n",
                   names[pat], blocks, ticks / (real * 10.0), ticks / (2.0 * n), ms * 1e3, flops / (ms * 1e-3) / 1e12,
                   100.0 * flops / (ms * 1e-3) / (blocks * 4096.0 * 2.4e9), blocks);
        }
    }
    return 0;
}

// Probe: what does s_memtime count on gfx950?  Per wave: N dependent v_add_f32 (4 shader clocks each on a 16-lane SIMD for a
// wave64) and N independent-chain v_mfma_f32_32x32x16_f16 (CH accumulator chains), timed with s_memtime (ticks) and
// s_memrealtime (100 MHz) -> tick rate, ticks per VALU instruction, ticks per MFMA for 1/2/4/8 chains and 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/clock_calibration.hip -o clock_calibration && ./clock_calibration
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const f16x8* in, float* out, unsigned long long* res, int iters) {
    extern __shared__ char dyn[];
    const int lane = threadIdx.x & 63;
    f16x8 a = in[lane], b = in[64 + lane];
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v = (float)lane;
    if (dyn[0] == 77) out[0] = 1.f;
    __syncthreads();
    // VALU: 64 dependent adds per iteration
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(1.0f));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // MFMA: 64 per iteration over CH chains
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % CH], 0, 0, 0);
    }
    unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    float s = v;
#pragma unroll
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 17) { res[0] = t1 - t0; res[1] = r1 - r0; res[2] = t2 - t1; res[3] = r2 - r1; }
}

template <int CH, int WPS>
void run() {
    f16x8* in; float* out; unsigned long long* res;
    const int blocks = 256 * WPS, iters = 2000;
    CHECK(hipMalloc(&in, 128 * sizeof(f16x8))); CHECK(hipMemset(in, 0, 128 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&res, 64));
    const size_t lds = WPS == 1 ? 100 * 1024 : 60 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<CH, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<CH, WPS>), dim3(blocks), dim3(256), lds, 0, in, out, res, iters); CHECK(hipDeviceSynchronize()); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<CH, WPS>), dim3(blocks), dim3(256), lds, 0, in, out, res, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k<CH, WPS>), 256, lds);
    unsigned long long h[4]; CHECK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
    const double n = 64.0 * iters;
    printf("%d wave(s)/SIMD, %d MFMA chain(s): VALU phase %.3f GHz ticks, %5.2f ticks / %5.2f ns per v_add | MFMA phase %.3f GHz ticks, %5.2f ticks / %5.2f ns per MFMA per wave"
           " -> %6.1f TFLOP/s f16 if all %d waves/SIMD run concurrently; kernel %.1f us (both phases; occupancy %d workgroups/CU)\n", WPS, CH, h[0] / (h[1] * 10.0), h[0] / n, h[1] * 10.0 / n, h[2] / (h[3] * 10.0), h[2] / n, h[3] * 10.0 / n,
           32768.0 * 1024 * WPS / (h[3] * 10.0 / n) / 1e3, WPS, ms * 1e3, occ);
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(res));
}

int main() {
    run<1, 1>(); run<2, 1>(); run<4, 1>(); run<8, 1>();
    run<1, 2>(); run<2, 2>(); run<4, 2>(); run<8, 2>();
    return 0;
}

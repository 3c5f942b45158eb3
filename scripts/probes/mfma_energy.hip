// Probe: at the chip's power cap the sustained matrix rate is energy per MFMA, so WHICH instruction form is cheapest?  All 256 CUs,
// one 512-thread workgroup per CU (2 waves per SIMD; variant W1: 256 threads, 1 wave per SIMD), N(0,1) half operands (bf16 for the
// bf16 variant), nothing but MFMAs in the loop, the same number of flops per wave in every variant:
//   V32   v_mfma_f32_32x32x16_f16, C/D in VGPRs, 8 accumulator chains, 4 + 4 operand registers        (scripts/probes/mfma_power.hip)
//   A32   the same with C/D in AGPRs
//   V16   v_mfma_f32_16x16x32_f16 (twice as many instructions, 4-register accumulators), C/D in VGPRs
//   A16   the same with C/D in AGPRs
//   R32   V32 with ONE A and ONE B register set for every MFMA (operand fetch of identical registers)
//   B32   v_mfma_f32_32x32x16_bf16, C/D in VGPRs
//   W1    V32 with one wave per SIMD
// Reports the clock the chip settles at and the TFLOP/s; higher = less energy per flop.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_energy.hip -o mfma_energy && ./mfma_energy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { V32, A32, V16, A16, R32, B32, W1 };

template <int VAR>
__global__ __launch_bounds__(512) void k(const f16x8* in, float* out, unsigned long long* res, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(wave * 8 + i) * 64 + lane]; b[i] = in[(wave * 8 + 4 + i) * 64 + lane]; }
    float s = 0.f;
    __syncthreads();
    unsigned long long t0, r0, t1, r1;
    if constexpr (VAR == V32 || VAR == R32 || VAR == W1 || VAR == A32 || VAR == B32) {
        f32x16 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        if constexpr (VAR == A32) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                const int ia = VAR == R32 ? 0 : (i >> 1) & 3, ib = VAR == R32 ? 0 : (i >> 3) & 3;
                if constexpr (VAR == A32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i & 7]) : "v"(a[ia]), "v"(b[ib]));
                else if constexpr (VAR == B32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a[ia]), "v"(b[ib]));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a[ia]), "v"(b[ib]));
            }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        if constexpr (VAR == A16) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 96; ++i) {
                if constexpr (VAR == A16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i & 7]) : "v"(a[(i >> 1) & 3]), "v"(b[(i >> 3) & 3]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a[(i >> 1) & 3]), "v"(b[(i >> 3) & 3]));
            }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { res[2 * blockIdx.x] = t1 - t0; res[2 * blockIdx.x + 1] = r1 - r0; }
}

static float gauss() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

template <int VAR>
void run(const char* name, const f16x8* in, float* out, unsigned long long* res) {
    const int iters = 4000, blocks = 256, threads = VAR == W1 ? 256 : 512;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, in, out, res, iters); CHECK(hipDeviceSynchronize()); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, in, out, res, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> r(2 * blocks); CHECK(hipMemcpy(r.data(), res, r.size() * 8, hipMemcpyDeviceToHost));
    double ticks = 0, real = 0; for (int b = 0; b < blocks; ++b) { ticks += r[2 * b]; real += r[2 * b + 1]; }
    const double flops = (double)blocks * (threads / 64) * 48.0 * iters * 32768.0;
    printf("%-70s clock %.3f GHz, kernel %8.1f us -> %7.1f TFLOP/s\n", name, ticks / (real * 10.0), ms * 1e3, flops / (ms * 1e-3) / 1e12);
}

int main() {
    f16x8* in; float* out; unsigned long long* res;
    CHECK(hipMalloc(&in, 64 * 64 * sizeof(f16x8))); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&res, 256 * 16));
    std::vector<_Float16> h(64 * 64 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)gauss();
    CHECK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {
        run<V32>("V32  32x32x16 f16, C/D in VGPRs", in, out, res);
        run<A32>("A32  32x32x16 f16, C/D in AGPRs", in, out, res);
        run<V16>("V16  16x16x32 f16, C/D in VGPRs", in, out, res);
        run<A16>("A16  16x16x32 f16, C/D in AGPRs", in, out, res);
        run<R32>("R32  32x32x16 f16, one A / one B register set", in, out, res);
        run<W1>("W1   32x32x16 f16, one wave per SIMD", in, out, res);
    }
    std::vector<__bf16> hb(64 * 64 * 8);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (__bf16)gauss();
    CHECK(hipMemcpy(in, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    run<B32>("B32  32x32x16 bf16, C/D in VGPRs", in, out, res);
    run<B32>("B32  32x32x16 bf16, C/D in VGPRs", in, out, res);
    return 0;
}

// Probe: does the accumulator FILE of an MFMA stream change how much VALU work another wave of the same SIMD gets done beside it?
// scripts/probes/simd_sharing.hip found that a VALU stream next to a saturated v_mfma_f32_32x32x16_f16 stream (accumulators in
// architectural VGPRs) makes almost no progress.  Here the matrix stream exists in three forms:
//   Mv: C/D in VGPRs (what the compiler emits below 256 registers),   Ma: C/D in AGPRs (inline asm, "a" constraints),
//   Ms: v_mfma_f32_16x16x32_f16 with C/D in VGPRs (4 registers per accumulator instead of 16)
// and the vector stream V is the softmax VALU work of one 64-key attention tile (32 v_exp_f32, row sum, 3-instruction split).
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD; waves 0-3 run stream A, waves 4-7 stream B.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_agpr_sharing.hip -o mfma_agpr_sharing && ./mfma_agpr_sharing
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, unsigned& ha, unsigned& la, unsigned& hb, unsigned& lb) {
    asm("s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %2, %6, %7\n\t"
        "v_fma_mixlo_f16 %1, %4, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, 1.0, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(ha), "=&v"(la), "=&v"(hb), "=&v"(lb)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}

// 48 MFMAs 32x32x16 per iteration, two accumulator chains in VGPRs
__device__ __forceinline__ float stream_mv(const f16x8* in, int lane, int iters) {
    f16x8 a[4];
    for (int i = 0; i < 4; ++i) a[i] = in[lane + 64 * i];
    f32x16 o[2];
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m)
#pragma unroll
            for (int d = 0; d < 2; ++d)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(o[d]) : "v"(a[(m + d) & 3]), "v"(a[m & 3]));
    }
    float acc = 0.f;
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) acc += o[d][r];
    return acc;
}

// the same with the accumulators in AGPRs
__device__ __forceinline__ float stream_ma(const f16x8* in, int lane, int iters) {
    f16x8 a[4];
    for (int i = 0; i < 4; ++i) a[i] = in[lane + 64 * i];
    f32x16 o[2];
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    asm volatile("" : "+a"(o[0]), "+a"(o[1]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m)
#pragma unroll
            for (int d = 0; d < 2; ++d)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o[d]) : "v"(a[(m + d) & 3]), "v"(a[m & 3]));
    }
    float acc = 0.f;
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) acc += o[d][r];
    return acc;
}

// 96 MFMAs 16x16x32 per iteration (the same flops), four accumulator chains of 4 VGPRs
__device__ __forceinline__ float stream_ms(const f16x8* in, int lane, int iters) {
    f16x8 a[4];
    for (int i = 0; i < 4; ++i) a[i] = in[lane + 64 * i];
    f32x4 o[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 4; ++r) o[d][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(o[d]) : "v"(a[(m + d) & 3]), "v"(a[m & 3]));
    }
    float acc = 0.f;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 4; ++r) acc += o[d][r];
    return acc;
}

__device__ __forceinline__ float stream_v(int lane, int iters) {
    float s[32];
    for (int r = 0; r < 32; ++r) s[r] = -1.f - 0.01f * r - 1e-3f * lane;
    float l = 0.f;
    unsigned keep = 0;
    for (int it = 0; it < iters; ++it) {
        float psum = 0.f, mt = s[0];
#pragma unroll
        for (int r = 1; r < 32; ++r) mt = fmaxf(mt, s[r]);
#pragma unroll
        for (int r = 0; r < 32; r += 4) {
            const float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
            const float p2 = __builtin_amdgcn_exp2f(s[r + 2]), p3 = __builtin_amdgcn_exp2f(s[r + 3]);
            psum += (p0 + p1) + (p2 + p3);
            unsigned ha, la, hb, lb;
            split4(p0, p1, p2, p3, ha, la, hb, lb);
            keep ^= ha ^ la ^ hb ^ lb;
        }
        l += psum + mt * 1e-9f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s[r] = s[r] * 0.999f - 1e-4f;
    }
    return l + (float)(keep & 1);
}

// plain full-rate VALU (no transcendental, no conversions): 256 dependent-chain-free v_fma_f32 per iteration
__device__ __forceinline__ float stream_f(int lane, int iters) {
    float s[16];
    for (int r = 0; r < 16; ++r) s[r] = 1.f + 0.01f * r + 1e-3f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[r]) : "v"(0.999f), "v"(1e-4f));
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += s[r];
    return acc;
}

// streams: 0 idle, 1 Mv, 2 Ma, 3 Ms, 4 V, 5 F
template <int W>
__device__ __forceinline__ float stream(const f16x8* in, int lane, int iters) {
    if constexpr (W == 1) return stream_mv(in, lane, iters);
    else if constexpr (W == 2) return stream_ma(in, lane, iters);
    else if constexpr (W == 3) return stream_ms(in, lane, iters);
    else if constexpr (W == 4) return stream_v(lane, iters);
    else if constexpr (W == 5) return stream_f(lane, iters);
    else return 0.f;
}

template <int SA, int SB>
__global__ __launch_bounds__(512, 1) void k(const f16x8* in, float* out, unsigned* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (wave < 4) r = stream<SA>(in, lane, iters);
    else r = stream<SB>(in, lane, iters);
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int SA, int SB>
void run(const char* name) {
    f16x8* in; float* out; unsigned* cyc;
    const int blocks = 256, iters = 400;
    CHECK(hipMalloc(&in, 256 * sizeof(f16x8))); CHECK(hipMemset(in, 0x3c, 256 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, blocks * 512 * 4)); CHECK(hipMalloc(&cyc, blocks * 8 * 4));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<SA, SB>), dim3(blocks), dim3(512), 0, 0, in, out, cyc, iters); CHECK(hipDeviceSynchronize()); }
    unsigned h[64 * 8]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    for (int blk = 0; blk < 64; ++blk) for (int w = 0; w < 4; ++w) { a += h[blk * 8 + w]; b += h[blk * 8 + 4 + w]; }
    printf("%-10s waves 0-3: %7.0f cycles / iteration   waves 4-7: %7.0f cycles / iteration\n", name, a / (256.0 * iters), b / (256.0 * iters));
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    printf("per iteration: M* = 48 MFMA 32x32x16 f16 (or 96 16x16x32) = 1536 cycles of matrix pipe; V = softmax VALU of one 64-key tile; F = 256 v_fma_f32\n");
    run<1, 0>("Mv | -");
    run<2, 0>("Ma | -");
    run<3, 0>("Ms | -");
    run<4, 0>("V  | -");
    run<5, 0>("F  | -");
    run<1, 4>("Mv | V");
    run<2, 4>("Ma | V");
    run<3, 4>("Ms | V");
    run<1, 5>("Mv | F");
    run<2, 5>("Ma | F");
    run<3, 5>("Ms | F");
    run<2, 2>("Ma | Ma");
    return 0;
}

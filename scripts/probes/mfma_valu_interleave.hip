// Probe: can ONE wave per SIMD keep the matrix pipe busy while it also runs the softmax VALU stream of a flash-attention
// tile?  Per loop iteration: NM independent-chain v_mfma_f32_32x32x16_f16 (48 per 32-query block of a 64-key tile with
// split-f16 operands) and the softmax/split VALU work of the same amount of data (exp2, row sum, 3-instruction (hi, lo)
// split), with and without __builtin_amdgcn_sched_group_barrier pinning "1 MFMA + NV VALU" groups.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_interleave.hip -o mfma_valu_interleave && ./mfma_valu_interleave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// QB query blocks per wave; MODE 0: MFMAs then VALU (phases), 1: source order mixed + compiler, 2: sched_group_barrier pinned
template <int QB, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const f16x8* in, float* out, unsigned* cyc, int iters) {
    extern __shared__ char dyn[];
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[QB][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[lane + 64 * i]; for (int q = 0; q < QB; ++q) b[q][i] = in[lane + 64 * (4 + i + 4 * q)]; }
    f32x16 s[QB][2], o[QB][2];
    float l[QB];
    f16x8 pf[QB][4], pl[QB][4];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        l[q] = 0.f;
        for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) { s[q][j][r] = -1.f - 0.01f * r; o[q][j][r] = 0.f; }
        for (int j = 0; j < 4; ++j) { pf[q][j] = a[j]; pl[q][j] = a[(j + 1) & 3]; }
    }
    if (dyn[0] == 77) out[0] = 1.f;     // keep the LDS allocation
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x16 sn[QB][2];
        // ---- QK^T(t+1): 24 MFMAs per query block ----
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sn[q][j][r] = -3.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sn[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(c + p + j) & 3], b[q][c], sn[q][j], 0, 0, 0);
        if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
        // ---- softmax(t): exp2, row sum, split -> P(t) ----
        f16x8 nf[QB][4], nl[QB][4];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            float psum = 0.f;
            if (MODE == 3) { for (int j = 0; j < 4; ++j) { nf[q][j] = pl[q][j]; nl[q][j] = pf[q][j]; } continue; }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float p0 = __builtin_amdgcn_exp2f(s[q][j][r]), p1 = __builtin_amdgcn_exp2f(s[q][j][r + 1]);
                    const float p2 = __builtin_amdgcn_exp2f(s[q][j][r + 2]), p3 = __builtin_amdgcn_exp2f(s[q][j][r + 3]);
                    psum += (p0 + p1) + (p2 + p3);
                    unsigned ha, la, hb, lb;
                    split4(p0, p1, p2, p3, ha, la, hb, lb);
                    unsigned* fw = reinterpret_cast<unsigned*>(&nf[q][j * 2 + (r >> 3)]);
                    unsigned* lw = reinterpret_cast<unsigned*>(&nl[q][j * 2 + (r >> 3)]);
                    fw[(r & 7) >> 1] = ha; fw[((r & 7) >> 1) + 1] = hb;
                    lw[(r & 7) >> 1] = la; lw[((r & 7) >> 1) + 1] = lb;
                }
            l[q] += psum;
        }
        if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
        // ---- PV(t-1): 24 MFMAs per query block with P(t-1) ----
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int d = 0; d < 2; ++d) o[q][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(g + d + p) & 3], p == 0 ? pl[q][g] : pf[q][g], o[q][d], 0, 0, 0);
        if (MODE == 2) {
            // 48*QB MFMAs, ~290*QB VALU: one MFMA, then 6 VALU
#pragma unroll
            for (int i = 0; i < 48 * QB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        // rotate
#pragma unroll
        for (int q = 0; q < QB; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { pf[q][j] = nf[q][j]; pl[q][j] = nl[q][j]; }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[q][j][r] = MODE == 3 ? s[q][j][r] + sn[q][j][r] : sn[q][j][r] * 1e-3f - 1.f;     // stand-in for the max bookkeeping
        }
    }
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < QB; ++q) { acc += l[q]; for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) acc += o[q][d][r] + s[q][d][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int QB, int MODE, int WPS>
void run(const char* name) {
    f16x8* in; float* out; unsigned* cyc;
    const int blocks = 256 * WPS, iters = 200;
    CHECK(hipMalloc(&in, 64 * 16 * sizeof(f16x8))); CHECK(hipMemset(in, 0, 64 * 16 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&cyc, blocks * 16));
    const size_t lds = WPS == 1 ? 100 * 1024 : 60 * 1024;        // pins the number of workgroups per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<QB, MODE, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<QB, MODE, WPS>), dim3(blocks), dim3(256), lds, 0, in, out, cyc, iters); CHECK(hipDeviceSynchronize()); }
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k<QB, MODE, WPS>), 256, lds);
    unsigned h[64];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 64; ++i) avg += h[i]; avg /= 64.0 * iters;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<QB, MODE, WPS>), dim3(blocks), dim3(256), lds, 0, in, out, cyc, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s [%d workgroup(s)/CU] %7.0f cycles / tile / wave; MFMA issue alone %5d; per SIMD %.0f%% of the matrix pipe; kernel %.1f us = %.0f TFLOP/s f16\n", name, occ, avg, 48 * QB * 32, 100.0 * 48 * QB * 32 * WPS / avg, ms * 1e3, (double)blocks * 4 * iters * 48 * QB * 32768 / (ms * 1e-3) / 1e12);
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 3, 1>("1 wave/SIMD, 32 q/wave, MFMAs only");
    run<1, 0, 1>("1 wave/SIMD, 32 q/wave, phases (MFMA | VALU | MFMA)");
    run<1, 1, 1>("1 wave/SIMD, 32 q/wave, compiler-scheduled");
    run<1, 2, 1>("1 wave/SIMD, 32 q/wave, sched_group_barrier 1 MFMA + 6 VALU");
    run<2, 0, 1>("1 wave/SIMD, 64 q/wave, phases");
    run<2, 1, 1>("1 wave/SIMD, 64 q/wave, compiler-scheduled");
    run<2, 2, 1>("1 wave/SIMD, 64 q/wave, sched_group_barrier 1 MFMA + 6 VALU");
    // (two workgroups per CU are not reliably co-scheduled by this launch pattern -- scripts/probes/clock_calibration.hip shows the
    //  second set running after the first -- so no 2-waves-per-SIMD rows here; the attention kernel trace covers that case)
    return 0;
}

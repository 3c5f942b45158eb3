// Probe: WHICH vector instructions make progress next to a saturated MFMA stream of another wave on the same SIMD (gfx950)?
// scripts/probes/mfma_agpr_sharing.hip: plain v_fma_f32 runs at ~70 % of its stand-alone rate beside the matrix stream, the softmax
// stream of an attention tile (v_exp_f32, v_cvt_pk_f16_f32, v_fma_mix*, v_max3, adds) at ~4 %.  Here every instruction class runs alone:
// waves 0-3 of a 512-thread workgroup issue 48 x v_mfma_f32_32x32x16_f16 per iteration, waves 4-7 (same SIMDs) N instructions of ONE
// class per iteration, both for the same number of iterations.  Reported: cycles per iteration alone, and the cycles per iteration the
// class needed WHILE the matrix stream was running (derived from the two finishing times).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_classes.hip -o mfma_valu_classes && ./mfma_valu_classes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float stream_m(const f16x8* in, int lane, int iters) {
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

// classes: 16 independent chains, REP x 16 instructions per iteration
enum { C_FMA = 1, C_EXP, C_CVT, C_MIX, C_MAX3, C_PKADD, C_PKFMA, C_MOV, C_ADDU, C_MULLO, C_LSHLADD64, C_RCP, C_PERM, C_DPP, C_CNDMASK, C_EXP16, C_LDEXP, C_CVTI, C_NOPS, C_MOV64, C_PKMUL, C_ADDF, C_PKADD16, C_PKFMA16, C_FMA64, C_DSREAD, C_BPERM, C_ACCW };
template <int CLS>
__device__ __forceinline__ float stream_c(int lane, int iters) {
    constexpr int REP = 8;
    float s[16];
    f32x2 p[16];
    for (int r = 0; r < 16; ++r) { s[r] = 0.5f + 0.01f * r + 1e-3f * lane; p[r] = f32x2{s[r], s[r] + 0.25f}; }
    unsigned long long q[16];
    for (int r = 0; r < 16; ++r) q[r] = lane + r;
    const float c0 = 0.999f, c1 = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < REP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (CLS == C_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[r]) : "v"(c0), "v"(c1));
                else if constexpr (CLS == C_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(s[r]));
                else if constexpr (CLS == C_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(s[r]));
                else if constexpr (CLS == C_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[r]));
                else if constexpr (CLS == C_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_MIX) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[r]) : "v"(c0), "v"(c1));
                else if constexpr (CLS == C_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 15]));
                else if constexpr (CLS == C_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 15]));
                else if constexpr (CLS == C_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[r]) : "v"(q[(r + 1) & 15]));
                else if constexpr (CLS == C_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(s[r]) : "v"(c0), "v"(c1));
                else if constexpr (CLS == C_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[r]));
                else if constexpr (CLS == C_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(s[r]) : "v"(1));
                else if constexpr (CLS == C_CVTI) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(s[r]));
                else if constexpr (CLS == C_NOPS) asm volatile("s_nop 3" : "+v"(s[r]));
                else if constexpr (CLS == C_MOV64) asm volatile("v_mov_b64 %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 15]));
                else if constexpr (CLS == C_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 15]));
                else if constexpr (CLS == C_ADDF) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_PKADD16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(s[r]) : "v"(c0));
                else if constexpr (CLS == C_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q[r]) : "v"(q[(r + 1) & 15]));
                else if constexpr (CLS == C_DSREAD) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(float4*)&p[r & 14]) : "v"(lane * 16) : "memory");
                else if constexpr (CLS == C_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(s[r]) : "v"(lane * 4) : "memory");
                else if constexpr (CLS == C_ACCW) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(s[r]) : "a0");
            }
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += s[r] + p[r][0] + p[r][1] + (float)(unsigned)q[r];
    return acc;
}

template <int CLS, int WITH_M>
__global__ __launch_bounds__(512, 1) void k(const f16x8* in, float* out, unsigned* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (wave < 4) { if (WITH_M) r = stream_m(in, lane, iters); }
    else r = stream_c<CLS>(lane, iters);
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int CLS, int WITH_M>
void one(double& m, double& c) {
    f16x8* in; float* out; unsigned* cyc;
    const int blocks = 256, iters = 400;
    CHECK(hipMalloc(&in, 256 * sizeof(f16x8))); CHECK(hipMemset(in, 0x3c, 256 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, blocks * 512 * 4)); CHECK(hipMalloc(&cyc, blocks * 8 * 4));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<CLS, WITH_M>), dim3(blocks), dim3(512), 0, 0, in, out, cyc, iters); CHECK(hipDeviceSynchronize()); }
    unsigned h[64 * 8]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    for (int blk = 0; blk < 64; ++blk) for (int w = 0; w < 4; ++w) { a += h[blk * 8 + w]; b += h[blk * 8 + 4 + w]; }
    m = a / (256.0 * iters); c = b / (256.0 * iters);
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

template <int CLS>
void run(const char* name) {
    double m0, c0, m1, c1;
    one<CLS, 0>(m0, c0);
    one<CLS, 1>(m1, c1);
    // beside M: the class finished x iterations while M ran (m1 cycles per iteration of M, both run the same iteration count):
    //   c1 = m1 + (1 - x) c0  (per iteration of the whole run)  ->  x = 1 - (c1 - m1) / c0;   cycles per class iteration beside M = m1 / x
    const double x = c1 > m1 ? 1.0 - (c1 - m1) / c0 : 1.0;
    printf("%-22s alone %6.0f cycles / 128 instr (%.2f per instr)   with M: M %6.0f, class %6.0f -> %5.1f %% of its iterations done beside M = %.2f cycles per instr beside M (rate %.2f of alone)\n",
           name, c0, c0 / 128.0, m1, c1, 100.0 * x, x > 0.01 ? m1 / x / 128.0 : -1.0, x > 0.01 ? c0 / (m1 / x) : 0.0);
}

int main() {
    run<C_FMA>("v_fma_f32");
    run<C_EXP>("v_exp_f32");
    run<C_EXP16>("v_exp_f16");
    run<C_RCP>("v_rcp_f32");
    run<C_CVT>("v_cvt_pk_f16_f32");
    run<C_MIX>("v_fma_mixlo_f16");
    run<C_MAX3>("v_max3_f32");
    run<C_PKADD>("v_pk_add_f32");
    run<C_PKFMA>("v_pk_fma_f32");
    run<C_MOV>("v_mov_b32");
    run<C_ADDU>("v_add_u32");
    run<C_MULLO>("v_mul_lo_u32");
    run<C_LSHLADD64>("v_lshl_add_u64");
    run<C_PERM>("v_perm_b32");
    run<C_DPP>("v_mov_b32_dpp");
    run<C_CNDMASK>("v_cndmask_b32");
    run<C_LDEXP>("v_ldexp_f32");
    run<C_CVTI>("v_cvt_i32_f32");
    run<C_NOPS>("s_nop 3");
    run<C_MOV64>("v_mov_b64");
    run<C_PKMUL>("v_pk_mul_f32");
    run<C_ADDF>("v_add_f32");
    run<C_PKADD16>("v_pk_add_f16");
    run<C_PKFMA16>("v_pk_fma_f16");
    run<C_FMA64>("v_fma_f64");
    run<C_DSREAD>("ds_read_b128 + wait");
    run<C_BPERM>("ds_bpermute_b32 + wait");
    run<C_ACCW>("v_accvgpr_write_b32");
    return 0;
}

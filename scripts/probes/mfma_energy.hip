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
//   M8    (round 5) v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (N(0,1) values) and unit E8M0 block scales: 4x the flops
//         of a 32x32x16 f16 MFMA per instruction -- does the block-scaled fp8 path sustain >= 1.7x the f16 rate under the power cap?
//   I8    v_mfma_i32_32x32x32_i8 (2x the flops per instruction), operands uniform in [-127, 127]
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

enum { V32, A32, V16, A16, R32, B32, W1, M8, I8 };
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(512) void k(const f16x8* in, float* out, unsigned long long* res, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(wave * 8 + i) * 64 + lane]; b[i] = in[(wave * 8 + 4 + i) * 64 + lane]; }
    float s = 0.f;
    __syncthreads();
    unsigned long long t0, r0, t1, r1;
    if constexpr (VAR == M8) {
        // operands: 8 dwords = 32 fp8 bytes per lane (the same random bytes reinterpreted: the host filled `in` with e4m3 encodings for this variant)
        i32x8 a8[2], b8[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a8[i] = i32x8{__builtin_bit_cast(i32x4, a[2 * i])[0], __builtin_bit_cast(i32x4, a[2 * i])[1], __builtin_bit_cast(i32x4, a[2 * i])[2], __builtin_bit_cast(i32x4, a[2 * i])[3],
                          __builtin_bit_cast(i32x4, a[2 * i + 1])[0], __builtin_bit_cast(i32x4, a[2 * i + 1])[1], __builtin_bit_cast(i32x4, a[2 * i + 1])[2], __builtin_bit_cast(i32x4, a[2 * i + 1])[3]};
            b8[i] = i32x8{__builtin_bit_cast(i32x4, b[2 * i])[0], __builtin_bit_cast(i32x4, b[2 * i])[1], __builtin_bit_cast(i32x4, b[2 * i])[2], __builtin_bit_cast(i32x4, b[2 * i])[3],
                          __builtin_bit_cast(i32x4, b[2 * i + 1])[0], __builtin_bit_cast(i32x4, b[2 * i + 1])[1], __builtin_bit_cast(i32x4, b[2 * i + 1])[2], __builtin_bit_cast(i32x4, b[2 * i + 1])[3]};
        }
        f32x16 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i)      // 12 x 4 = the flops of 48 32x32x16 f16 MFMAs
                acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc[i & 7], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else if constexpr (VAR == I8) {
        i32x4 a8[4], b8[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a8[i] = __builtin_bit_cast(i32x4, a[i]); b8[i] = __builtin_bit_cast(i32x4, b[i]); }
        i32x16 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 24; ++i)      // 24 x 2 = the flops of 48 32x32x16 f16 MFMAs
                acc[i & 7] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a8[(i >> 1) & 3], b8[(i >> 3) & 3], acc[i & 7], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += (float)acc[c][r];
    } else if constexpr (VAR == V32 || VAR == R32 || VAR == W1 || VAR == A32 || VAR == B32) {
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
    // fp8 e4m3 encodings of N(0,1) values (OCP e4m3fn: bias 7, 3 mantissa bits), unit block scales
    std::vector<unsigned char> h8(64 * 64 * 16);
    for (size_t i = 0; i < h8.size(); ++i) {
        float v = gauss(); const unsigned sgn = v < 0 ? 0x80u : 0u; v = fabsf(v);
        int e; float m = frexpf(v, &e);                 // v = m 2^e, m in [0.5, 1)
        unsigned code = 0;
        if (v >= 0.015625f) {                           // normal range 2^-6 ..
            int E = e - 1 + 7; int M = (int)lrintf((m * 2.f - 1.f) * 8.f); if (M == 8) { M = 0; ++E; }
            if (E > 15) { E = 15; M = 6; }
            code = (unsigned)(E << 3 | M);
        } else code = (unsigned)lrintf(v * 512.f);      // subnormals: multiples of 2^-9
        h8[i] = (unsigned char)(sgn | code);
    }
    CHECK(hipMemcpy(in, h8.data(), h8.size(), hipMemcpyHostToDevice));
    run<M8>("M8   32x32x64 MX fp8 e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4), unit scales", in, out, res);
    run<M8>("M8   32x32x64 MX fp8 e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4), unit scales", in, out, res);
    for (size_t i = 0; i < h8.size(); ++i) h8[i] = (unsigned char)(rand() & 0xFF);
    CHECK(hipMemcpy(in, h8.data(), h8.size(), hipMemcpyHostToDevice));
    run<I8>("I8   32x32x32 i8 (v_mfma_i32_32x32x32_i8), uniform bytes", in, out, res);
    run<I8>("I8   32x32x32 i8 (v_mfma_i32_32x32x32_i8), uniform bytes", in, out, res);
    run<V32>("V32  32x32x16 f16 again (operands now random bytes)", in, out, res);
    return 0;
}

// Probe: LDS read cost of the attention fragment patterns on gfx950 (unpadded 128-byte rows, 16-byte chunk XOR swizzles).
//   K: ds_read_b128, lane (l31 = key row, hi = k-group) reads logical chunk 2c+hi of row l31        (ideal 8 cycles: 1 KB / 128 B/clk)
//   V: ds_read_b64_tr_b16, lane i of a 16-lane group reads the 8-byte piece (key i>>2, dv 4(i&3)..)  (ideal 4 cycles)
// for several swizzles f(row) applied to the 16-byte chunk index, 1 wave alone and 8 waves of a workgroup together.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_read_patterns.hip -o lds_read_patterns && ./lds_read_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swz(int mode, int row) {
    switch (mode) {
        case 0: return 0;
        case 1: return (row >> 1) & 7;
        case 2: return row & 7;
        case 3: return ((row >> 1) & 1) << 2;
        case 4: return ((row >> 1) & 3) << 1;
        case 5: return (row & 3) << 1;
        case 6: return (row >> 2) & 7;
        default: return 0;
    }
}

template <int KIND>   // 0: K pattern b128, 1: V pattern tr_b64, 2: V pattern with plain ds_read_b64 (same addresses)
__global__ __launch_bounds__(512) void k(unsigned* cyc, float* sink, int mode, int reps) {
    __shared__ __attribute__((aligned(1024))) char smem[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (wave & 1) * 32768;
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned addr[8];
    if (KIND == 0) {
        for (int c = 0; c < 4; ++c) {              // 4 chunks x 2 key blocks
            addr[c] = lds0 + l31 * 128 + (((2 * c + hi) ^ swz(mode, l31)) * 16);
            addr[4 + c] = addr[c] + 32 * 128;
        }
    } else {
        for (int j = 0; j < 8; ++j) {              // (group g = j>>1: keys 16g..., d = j&1)
            const int key = 4 * hi + ((lane & 15) >> 2) + 16 * (j >> 1);
            const int chunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1) + 4 * (j & 1);
            addr[j] = lds0 + key * 128 + ((chunk ^ swz(mode, key)) * 16) + 8 * (lane & 1);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr[j]));
                asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(v));
                acc += v;
            } else if (KIND == 1) {
                f32x2 v, w;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr[j]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(w) : "v"(addr[j]));
                asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(v), "+v"(w));
                acc[0] += v[0]; acc[1] += w[1];
            } else {
                f32x2 v, w;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[j]));
                asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(w) : "v"(addr[j]));
                asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(v), "+v"(w));
                acc[0] += v[0]; acc[1] += w[1];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int mode, int waves) {
    unsigned* cyc; float* sink;
    CHECK(hipMalloc(&cyc, 64 * 4)); CHECK(hipMalloc(&sink, 512 * 4));
    const int reps = 200;
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, cyc, sink, mode, reps); CHECK(hipDeviceSynchronize()); }
    unsigned h[8]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0; for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    const int per = KIND == 0 ? 8 : 16;       // LDS instructions per rep per wave
    printf("%-34s swizzle %d, %d wave(s): %6.1f cycles per instruction per wave; CU-wide %5.1f cycles per instruction\n", name, mode, waves,
           mx / (reps * per), mx / (reps * per * waves));
    CHECK(hipFree(cyc)); CHECK(hipFree(sink));
}

int main() {
    for (int waves : {1, 8}) {
        for (int mode : {0, 1, 2, 6}) run<0>("K  ds_read_b128", mode, waves);
        for (int mode : {0, 1, 2, 3, 4, 5}) run<1>("V  ds_read_b64_tr_b16", mode, waves);
        for (int mode : {0, 3}) run<2>("V  ds_read_b64 (same addresses)", mode, waves);
    }
    return 0;
}

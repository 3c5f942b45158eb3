// Probe: how long does the GPU take to START the workgroups of one launch, as a function of the workgroup's footprint?
// (The 256x256-tile GEMM trace shows its 256 blocks of 512 threads / 248 VGPRs / 128-160 KB LDS entering over ~12 us.)
// Every block stamps s_memrealtime (100 MHz) at entry, then spins for `hold_us` so that no block exits before the last
// one entered.  Variants: threads per block, dynamic LDS bytes, VGPR allocation (forced by clobbering a high register).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/dispatch_ramp.hip -o dispatch_ramp && ./dispatch_ramp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int HIGHREG>
__global__ void ramp_kernel(unsigned* entry, unsigned* hw, unsigned hold_ticks) {
    extern __shared__ char dyn[];
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    if (HIGHREG >= 250) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    else if (HIGHREG >= 120) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    if (threadIdx.x == 0) {
        entry[blockIdx.x] = t0;
        hw[blockIdx.x] = __builtin_amdgcn_s_getreg(0xF804) | (__builtin_amdgcn_s_getreg(0xF814) << 28);
        dyn[0] = 1;
    }
    while ((unsigned)__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}

template <int HIGHREG>
static void run(const char* name, int blocks, int threads, int lds) {
    unsigned *entry, *hw;
    CHECK(hipMalloc(&entry, blocks * 4)); CHECK(hipMalloc(&hw, blocks * 4));
    CHECK(hipFuncSetAttribute((const void*)ramp_kernel<HIGHREG>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    std::vector<unsigned> h(blocks);
    double spans[5];
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(ramp_kernel<HIGHREG>, dim3(blocks), dim3(threads), lds, 0, entry, hw, 4000u /* 40 us */);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), entry, blocks * 4, hipMemcpyDeviceToHost));
        unsigned mn = h[0];
        for (unsigned v : h) if ((int)(v - mn) < 0) mn = v;
        std::vector<unsigned> d(blocks);
        for (int i = 0; i < blocks; ++i) d[i] = h[i] - mn;
        std::sort(d.begin(), d.end());
        spans[rep] = d[blocks - 1] * 0.01;
        if (rep == 4) printf("%-44s blocks %4d: entry p50 %6.2f us  p90 %6.2f us  last %6.2f us   (5 runs last: %.2f %.2f %.2f %.2f %.2f)\n", name, blocks,
                             d[blocks / 2] * 0.01, d[blocks * 9 / 10] * 0.01, d[blocks - 1] * 0.01, spans[0], spans[1], spans[2], spans[3], spans[4]);
    }
    CHECK(hipFree(entry)); CHECK(hipFree(hw));
}

int main() {
    run<0>("256 thr,   0 KB LDS,  few VGPR", 256, 256, 0);
    run<0>("256 thr,   0 KB LDS,  few VGPR", 1024, 256, 0);
    run<0>("512 thr,   0 KB LDS,  few VGPR", 256, 512, 0);
    run<0>("512 thr,  64 KB LDS,  few VGPR", 256, 512, 65536);
    run<0>("512 thr, 128 KB LDS,  few VGPR", 256, 512, 131072);
    run<0>("512 thr, 160 KB LDS,  few VGPR", 256, 512, 163840);
    run<120>("512 thr, 128 KB LDS, 121 VGPR", 256, 512, 131072);
    run<250>("512 thr, 128 KB LDS, 251 VGPR", 256, 512, 131072);
    run<250>("512 thr,   0 KB LDS, 251 VGPR", 256, 512, 0);
    run<250>("256 thr,  64 KB LDS, 251 VGPR", 512, 256, 65536);
    run<250>("256 thr,  80 KB LDS, 251 VGPR", 512, 256, 81920);
    run<250>("512 thr, 160 KB LDS, 251 VGPR", 256, 512, 163840);
    run<250>("512 thr, 160 KB LDS, 251 VGPR (2 per CU)", 512, 512, 163840);
    return 0;
}

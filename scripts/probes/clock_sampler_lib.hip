// Not a stand-alone probe: a tiny shared library (openglue_amd/lib/libprobe_clock.so, built by scripts/build_probes.sh) with ONE kernel that
// samples the shader clock while something else runs: a single wave spins for `n` windows of `gap` ticks of the constant 100 MHz
// counter (s_memrealtime) and stores the s_memtime (shader clock) ticks each window took.  Launched on its own stream beside the
// kernel under test (scripts/clock_under_load.py) it reports the clock the chip actually grants that kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_sampler_kernel(unsigned long long* out, int n, int gap) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
        unsigned long long r1;
        do { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); } while (r1 - r0 < (unsigned long long)gap);
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[2 * i] = r1 - r0;
        out[2 * i + 1] = c1 - c0;
    }
}

extern "C" int clock_sampler_launch(void* out, int n, int gap, void* stream) {
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n, gap);
    return (int)hipGetLastError();
}

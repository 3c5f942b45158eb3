// Probe: what bounds the epilogue stores of the 256x256 GEMM tile?  The trace (scripts/trace_gemm.py) shows 12.8k cycles to
// issue 32 x global_store_dwordx4 per wave (8 waves, 256 KB per tile) = ~20 B/clk/CU, the same with 32 or 256 CUs active.
// Every block writes 256 KB with 8 waves x 32 dwordx4 stores in different address patterns:
//   A  8 rows x 128 B per instruction, rows 2 KB apart   (the GEMM's hl32 epilogue: [token][4D halves], 32-channel groups)
//   B  1 KB contiguous per instruction                   (a token's 256 channels (hi|lo) written at once)
//   C  8 rows x 128 B, rows 4 KB apart
//   D  as A but the eight waves write disjoint 32 KB regions row by row (wave-contiguous)
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(512) void k(uint4* out, unsigned* cyc, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = (char*)out + (size_t)blockIdx.x * (256 * 2048);          // a 256-token x 2 KB region per block
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    __syncthreads();
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            size_t off;
            if (PAT == 0) {            // A: instruction i of wave w: tokens (w/4)*128 + (i/8)*32 + (i%4)*8 + lane/8, channel group (w%4)*2 + (i/4)%2
                const int tok = (wave >> 2) * 128 + (i >> 3) * 32 + (i & 3) * 8 + (lane >> 3);
                const int grp = (wave & 3) * 2 + ((i >> 2) & 1);
                off = (size_t)tok * 2048 + grp * 128 + (lane & 7) * 16;
            } else if (PAT == 1) {     // B: 1 KB contiguous: token = w*32 + i, bytes 1024*half... each token row 2 KB: two instrs per token
                const int tok = wave * 32 + i;
                off = (size_t)tok * 2048 + lane * 16;              // first KB of the token row (this block writes half of every row)
            } else if (PAT == 2) {     // C: rows 4 KB apart (a 2x wider matrix)
                const int tok = (wave >> 2) * 128 + (i >> 3) * 32 + (i & 3) * 8 + (lane >> 3);
                const int grp = (wave & 3) * 2 + ((i >> 2) & 1);
                off = ((size_t)tok * 4096 + grp * 128 + (lane & 7) * 16) % (256 * 2048);
            } else {                   // D: wave-contiguous 32 KB regions, 8 rows x 128 B with rows 128 B apart = 1 KB contiguous again but per wave region
                off = (size_t)wave * 32768 + i * 1024 + lane * 16;
            }
            *reinterpret_cast<uint4*>(base + off) = v;
        }
    }
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t2 = (unsigned)__builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

template <int PAT>
void run(const char* name, int blocks) {
    uint4* out; unsigned* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 2048 * 2)); CHECK(hipMalloc(&cyc, blocks * 8 * 2 * 4));
    std::vector<unsigned> h(blocks * 16);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(512), 0, 0, out, cyc, 1); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> a, b;
    for (int i = 0; i < blocks * 8; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-58s blocks %3d: issue median %6u max %6u cycles | incl. drain median %6u max %6u  -> %.1f B/clk/CU\n", name, blocks, a[a.size() / 2],
           a.back(), b[b.size() / 2], b.back(), 262144.0 / b[b.size() / 2]);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    for (int blocks : {32, 256}) {
        run<0>("A: 8 rows x 128 B per store, rows 2 KB apart (GEMM now)", blocks);
        run<1>("B: 1 KB contiguous per store (token-major)", blocks);
        run<2>("C: 8 rows x 128 B per store, rows 4 KB apart", blocks);
        run<3>("D: 1 KB contiguous, each wave its own 32 KB region", blocks);
    }
    return 0;
}

// Probe: data mapping of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 values = element index; lane l supplies byte
// address addr[l] (host-chosen patterns); prints which 4 LDS elements each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    u16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    unsigned* d_addr; unsigned short* d_out;
    (void)hipMalloc(&d_addr, 256); (void)hipMalloc(&d_out, 512);
    unsigned h_addr[64]; unsigned short h_out[256];
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 8;                                         // lane l -> chunk l (4 elements each)
            else if (pat == 1) h_addr[l] = ((l & 15) / 4) * 256 + (l & 3) * 8 + (l >> 4) * 32;   // 16-lane group: 4 rows (pitch 128 el) x 4 chunks
            else h_addr[l] = (l & 15) * 2 * 0 + (l >> 4) * 128 + ((l & 15) >> 2) * 32 + (l & 3) * 8;  // [4][16] row-major blocks, 64 el apart
        }
        (void)hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        (void)hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr_el %4u -> %4u %4u %4u %4u", l, h_addr[l] / 2, h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}

// Probe (round 6): what the block-scaled fp8 P.V cross products of the attention kernel rest on, checked on the hardware.
//   (1) ds_read_b64_tr_b8: which LDS bytes a lane receives (lane l supplies the byte address of an 8-byte chunk);
//   (2) v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands: the assumed operand layout -- lane l holds row / column l & 31 and the 32 k-values
//       32 (l >> 5) + 4 w + e in byte e of dword w -- and the E8M0 scale operand: byte `opsel` of the lane's scale register applies to the lane's
//       own (row, k-block of 32);
//   (3) v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32: rounding, saturation and the direction of the scale.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void tr8_probe(const unsigned* addr, unsigned char* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned char)((i & 63) | ((i >> 7) << 6));      // byte = column (6 bits) | row bits 0..1 of a 128-byte-pitch matrix
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr[threadIdx.x]));
    for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = (unsigned char)(v >> (8 * j));
}
__global__ void tr8_probe_idx(const unsigned* addr, unsigned short* out) {      // the same with 16-bit "where did it come from" resolution: two passes, low / high byte of the LDS byte index
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned long long lo, hi;
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned char)(i & 255);
    __syncthreads();
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(base + addr[threadIdx.x]));
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned char)(i >> 8);
    __syncthreads();
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi) : "v"(base + addr[threadIdx.x]));
    for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = (unsigned short)(((lo >> (8 * j)) & 255) | (((hi >> (8 * j)) & 255) << 8));
}

__global__ void mx_probe(const unsigned* a8, const unsigned* b8, const unsigned* sa, const unsigned* sb, float* out) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int w = 0; w < 8; ++w) { a[w] = (int)a8[l * 8 + w]; b[w] = (int)b8[l * 8 + w]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
    // opsel = 1: byte 1 of the scale registers
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, (int)sa[l], 1, (int)sb[l]);
    for (int r = 0; r < 16; ++r) out[1024 + ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void cvt_probe(const float* in, float scale, unsigned* out, int n) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    const float x0 = in[2 * i], x1 = in[2 * i + 1];
    unsigned p = 0xAAAAAAAAu, q = 0xAAAAAAAAu, p2 = 0xAAAAAAAAu;
    asm volatile("v_cvt_pk_fp8_f32 %0, %3, %4\n\t"
                 "v_cvt_scalef32_pk_fp8_f32 %1, %3, %4, %5\n\t"
                 "v_cvt_pk_fp8_f32 %2, %3, %4 op_sel:[0,0,1]"
                 : "+v"(p), "+v"(q), "+v"(p2) : "v"(x0), "v"(x1), "v"(scale));
    out[3 * i] = p; out[3 * i + 1] = q; out[3 * i + 2] = p2;
}

static float e4m3_to_float(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) v = NAN;
    else if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}
static unsigned char float_to_e4m3(float x) {          // exactly representable inputs only
    for (int b = 0; b < 256; ++b) { const float v = e4m3_to_float((unsigned char)b); if (v == x && !(b == 0x80)) return (unsigned char)b; }
    return 0x7F;
}

int main() {
    // ---- (1) ----
    unsigned h_addr[64];
    unsigned* d_addr; unsigned short* d_o16;
    (void)hipMalloc(&d_addr, 256); (void)hipMalloc(&d_o16, 64 * 8 * 2);
    unsigned short h_o16[512];
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 8;                                                                  // lane l -> chunk l, contiguous
            else if (pat == 1) h_addr[l] = ((l & 15) >> 1) * 128 + (l & 1) * 8 + (l >> 4) * 16;               // 16-lane group: [8 rows, pitch 128 B][16 B], groups 16 B apart
            else h_addr[l] = ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 32;                             // 16-lane group: [4 rows][32 B]
        }
        (void)hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr8_probe_idx, dim3(1), dim3(64), 0, 0, d_addr, d_o16);
        (void)hipMemcpy(h_o16, d_o16, sizeof(h_o16), hipMemcpyDeviceToHost);
        printf("ds_read_b64_tr_b8 pattern %d (LDS byte index each lane receives; pitch 128: row = idx / 128, col = idx %% 128)\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %4u ->", l, h_addr[l]);
            for (int j = 0; j < 8; ++j) printf(" %4u", h_o16[l * 8 + j]);
            printf("\n");
        }
    }
    // ---- (2) ----
    {
        static float A[32][64], B[64][32], ref[32][32], ref2[32][32];
        unsigned ha[64 * 8], hb[64 * 8], hsa[64], hsb[64];
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = (float)((i + 2 * k) % 7 - 3);
        for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (float)((3 * k + j) % 5 - 2) * 0.5f;
        for (int l = 0; l < 64; ++l)
            for (int w = 0; w < 8; ++w) {
                unsigned va = 0, vb = 0;
                for (int e = 0; e < 4; ++e) {
                    const int k = 32 * (l >> 5) + 4 * w + e;
                    va |= (unsigned)float_to_e4m3(A[l & 31][k]) << (8 * e);
                    vb |= (unsigned)float_to_e4m3(B[k][l & 31]) << (8 * e);
                }
                ha[l * 8 + w] = va; hb[l * 8 + w] = vb;
            }
        // scales: A: byte 0 = 2^-(row & 3) for k-block 0, 2^(1 + (row & 1)) for k-block 1; byte 1 = 2^-11 everywhere.  B: byte 0 = 2^(col & 1), byte 1 = 2^0
        for (int l = 0; l < 64; ++l) {
            const int r = l & 31, kb = l >> 5;
            const int ea = kb == 0 ? -(r & 3) : 1 + (r & 1), eb = r & 1;
            hsa[l] = (unsigned)(127 + ea) | ((127u - 11u) << 8);
            hsb[l] = (unsigned)(127 + eb) | (127u << 8);
        }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, s2 = 0;
            for (int k = 0; k < 64; ++k) {
                const int kb = k >> 5;
                const int ea = kb == 0 ? -(i & 3) : 1 + (i & 1), eb = j & 1;
                s += (double)A[i][k] * B[k][j] * ldexp(1.0, ea + eb);
                s2 += (double)A[i][k] * B[k][j] * ldexp(1.0, -11);
            }
            ref[i][j] = (float)s; ref2[i][j] = (float)s2;
        }
        unsigned *da, *db, *dsa, *dsb; float* dout;
        (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dout, 2048 * 4);
        (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mx_probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
        static float hout[2048];
        (void)hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        int bad = 0, bad2 = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { bad += hout[i * 32 + j] != ref[i][j]; bad2 += hout[1024 + i * 32 + j] != ref2[i][j]; }
        printf("MX mfma 32x32x64 e4m3: assumed layout + per-lane scale byte 0: %d of 1024 results differ; opsel 1 (constant 2^-11 x 2^0): %d differ\n", bad, bad2);
        if (bad) for (int i = 0; i < 4; ++i) { for (int j = 0; j < 8; ++j) printf(" %9.3f/%9.3f", hout[i * 32 + j], ref[i][j]); printf("\n"); }
    }
    // ---- (3) ----
    {
        const float vals[] = {0.f, 1.f, 1.0625f, 1.1875f, 0.3f, 447.f, 448.f, 449.f, 480.f, 1000.f, 70000.f, 0.001953125f, 0.0009765625f, 0.0014f, 0.00292f, -0.7f, -500.f, 2.4e-7f, 3.0e-4f, 1.4e-4f,
                              240.f, 0.06f};
        const int n = sizeof(vals) / sizeof(float) / 2;
        float* din; unsigned* dout;
        (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dout, n * 12);
        (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
        const float scale = ldexpf(1.f, -11);
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, din, scale, dout, n);
        unsigned ho[64];
        (void)hipMemcpy(ho, dout, n * 12, hipMemcpyDeviceToHost);
        printf("v_cvt_pk_fp8_f32 (dst pre-set to 0xAAAAAAAA) and v_cvt_scalef32_pk_fp8_f32 with scale operand 2^-11:\n");
        for (int i = 0; i < n; ++i) {
            const unsigned p = ho[3 * i], q = ho[3 * i + 1], p2 = ho[3 * i + 2];
            printf("  x = (%g, %g): cvt_pk -> %08x = (%g, %g) | scaled -> %08x = (%g, %g) [x 2^11 = (%g, %g)] | op_sel hi word -> %08x\n", vals[2 * i], vals[2 * i + 1], p, e4m3_to_float(p & 255),
                   e4m3_to_float((p >> 8) & 255), q, e4m3_to_float(q & 255), e4m3_to_float((q >> 8) & 255), vals[2 * i] * 2048.f, vals[2 * i + 1] * 2048.f, p2);
        }
    }
    return 0;
}

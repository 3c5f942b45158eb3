// Split-f16 "3-pass" NT GEMM on the gfx950 matrix cores: fp32-class accuracy at 16/3 the rate of the
// exact-fp32 MFMA.
//
// Used for the 1x1 convolutions inside the attentional GNN (reference attention_gnn.py:16-20, 41:
// in_proj_q/k/v, fc.0, fc.3 -- 98 % of the GEMM FLOPs of the path).  fp32 MFMA runs at 1/16 of the f16
// rate and the parity bar (1e-3 on log-scores) rules out plain f16/bf16 operands, so
// every operand is carried as TWO f16 planes  x = hi + lo * 2^-11  (lo pre-scaled by 2^11: it has the
// magnitude of x, never an f16 subnormal) and
//        X Wᵀ  =  Xh Whᵀ  +  2^-11 (Xh Wlᵀ + Xl Whᵀ)            (the lo*lo term is 2^-22 relative: dropped)
// with fp32 accumulation in two accumulators: measured error equals the fp32 GEMM's.
// Operand range: |x| < 65504 (f16); activations of this network are O(10).
//
// Planes are produced by the PRODUCER's epilogue (this kernel, the attention kernel, the fp32 GEMM of the
// encoder) and weights are split once at pack time, so no conversion happens on the load path: tiles go
// global -> registers -> LDS as 16-byte chunks.
//
// MFMA orientation: D[outch][token] = W · Xᵀ  (A operand = weight tile, B operand = token tile), so a lane
// owns ONE token (column l&31) and, per 4-register group, FOUR CONSECUTIVE output channels: bias, residual
// and all stores (fp32 float4 / f16x4 planes) are vectorised along the channel axis.
// v_mfma_f32_32x32x16_f16; block tile 128 tokens x OC channels x 32 k, 4 waves as 2x2; LDS rows of 40
// halves (80 B) keep every 16-lane ds_read_b128 group on 16 distinct 4-bank slots.
#include "og_common.h"

namespace {

constexpr int TOK = 128;
constexpr int BKH = 32;          // k per tile (halves)
constexpr int LW = BKH + 8;      // padded LDS row (halves)
constexpr float LO_INV = 1.f / 2048.f;
constexpr float LO_SCALE = 2048.f;

template <int OC>
__global__ __launch_bounds__(256, 2) void gemm_nt_f16x3_kernel(GemmHArgs g, int tiles_m, int tiles_n) {
    constexpr int TI = OC / 64;            // MFMA tiles per wave along channels
    constexpr int WP = OC / 64;            // staging passes for W (64 rows per pass)
    __shared__ __attribute__((aligned(16))) _Float16 Xh[TOK * LW];
    __shared__ __attribute__((aligned(16))) _Float16 Xl[TOK * LW];
    __shared__ __attribute__((aligned(16))) _Float16 Wh[OC * LW];
    __shared__ __attribute__((aligned(16))) _Float16 Wl[OC * LW];

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int tm = (local / tiles_n) * 8 + xcd;       // all channel tiles of a token tile on one XCD
    const int tn = local % tiles_n;
    if (tm >= tiles_m) return;
    const int t0 = tm * TOK, n0 = tn * OC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wo = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int srow = tid >> 2, sc8 = (tid & 3) * 8;   // staging: row within a 64-row pass, k offset (halves)

    f16x8 rxh[2], rxl[2], rwh[WP], rwl[WP];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_tiles = [&](int k0) {
        const int kk = k0 + sc8;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = t0 + srow + 64 * p;
            if (row < g.M && kk < g.K) {
                rxh[p] = *reinterpret_cast<const f16x8*>(g.Ah + (int64_t)row * g.lda + kk);
                rxl[p] = *reinterpret_cast<const f16x8*>(g.Al + (int64_t)row * g.lda + kk);
            } else { rxh[p] = zero8; rxl[p] = zero8; }
        }
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int row = n0 + srow + 64 * p;
            if (row < g.N && kk < g.K) {
                rwh[p] = *reinterpret_cast<const f16x8*>(g.Bh + (int64_t)row * g.ldb + kk);
                rwl[p] = *reinterpret_cast<const f16x8*>(g.Bl + (int64_t)row * g.ldb + kk);
            } else { rwh[p] = zero8; rwl[p] = zero8; }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f16x8*>(&Xh[(srow + 64 * p) * LW + sc8]) = rxh[p];
            *reinterpret_cast<f16x8*>(&Xl[(srow + 64 * p) * LW + sc8]) = rxl[p];
        }
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            *reinterpret_cast<f16x8*>(&Wh[(srow + 64 * p) * LW + sc8]) = rwh[p];
            *reinterpret_cast<f16x8*>(&Wl[(srow + 64 * p) * LW + sc8]) = rwl[p];
        }
    };

    f32x16 acc0[TI][2], acc1[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int x_off = (wt * 64 + l31) * LW + 8 * hi;
    const int w_off = (wo * (OC / 2) + l31) * LW + 8 * hi;

    const int nk = (g.K + BKH - 1) / BKH;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BKH);
#pragma unroll
        for (int ks = 0; ks < BKH / 16; ++ks) {
            f16x8 wh[TI], wl[TI], xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                wh[i] = *reinterpret_cast<const f16x8*>(&Wh[w_off + i * 32 * LW + 16 * ks]);
                wl[i] = *reinterpret_cast<const f16x8*>(&Wl[w_off + i * 32 * LW + 16 * ks]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xh[j] = *reinterpret_cast<const f16x8*>(&Xh[x_off + j * 32 * LW + 16 * ks]);
                xl[j] = *reinterpret_cast<const f16x8*>(&Xl[x_off + j * 32 * LW + 16 * ks]);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xl[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xh[j], acc1[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- epilogue: lane = one token, 4 consecutive channels per register group ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = t0 + wt * 64 + j * 32 + l31;
        if (tok >= g.M) continue;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = n0 + wo * (OC / 2) + i * 32 + 8 * q + 4 * hi;
                if (oc >= g.N) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc0[i][j][4 * q + e] + acc1[i][j][4 * q + e] * LO_INV;
                if (g.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + oc);
                    v += b;
                }
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (g.res) v += *reinterpret_cast<const f32x4*>(g.res + (int64_t)tok * g.ldr + oc);
                if (g.C32) *reinterpret_cast<f32x4*>(g.C32 + (int64_t)tok * g.ldc + oc) = v;
                if (g.Ch) {
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const _Float16 h = (_Float16)v[e];
                        vh[e] = h;
                        vl[e] = (_Float16)((v[e] - (float)h) * LO_SCALE);
                    }
                    *reinterpret_cast<f16x4*>(g.Ch + (int64_t)tok * g.ldch + oc) = vh;
                    *reinterpret_cast<f16x4*>(g.Cl + (int64_t)tok * g.ldch + oc) = vl;
                }
            }
    }
}

// x -> (hi, lo) planes, elementwise (test helper and weight/activation conversion outside the GEMMs)
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int64_t n4, _Float16* __restrict__ h,
                                                        _Float16* __restrict__ l) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    f16x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 t = (_Float16)v[e];
        vh[e] = t;
        vl[e] = (_Float16)((v[e] - (float)t) * LO_SCALE);
    }
    *reinterpret_cast<f16x4*>(h + 4 * i) = vh;
    *reinterpret_cast<f16x4*>(l + 4 * i) = vl;
}

}  // namespace

int og_launch_gemm_f16x3(const GemmHArgs& a, hipStream_t stream) {
    if (!a.Ah || !a.Al || !a.Bh || !a.Bl || a.M <= 0 || a.N <= 0 || a.K <= 0) return OG_E_INVALID;
    if (!a.C32 && !a.Ch) return OG_E_INVALID;
    if ((a.Ch == nullptr) != (a.Cl == nullptr)) return OG_E_INVALID;
    if ((a.lda & 7) || (a.ldb & 7) || (a.K & 7) || (a.N & 3)) return OG_E_ALIGN;
    if (((uintptr_t)a.Ah & 15) || ((uintptr_t)a.Al & 15) || ((uintptr_t)a.Bh & 15) || ((uintptr_t)a.Bl & 15)) return OG_E_ALIGN;
    if (a.C32 && (((uintptr_t)a.C32 & 15) || (a.ldc & 3))) return OG_E_ALIGN;
    if (a.Ch && (((uintptr_t)a.Ch & 7) || ((uintptr_t)a.Cl & 7) || (a.ldch & 3))) return OG_E_ALIGN;
    if (a.res && (((uintptr_t)a.res & 15) || (a.ldr & 3))) return OG_E_ALIGN;
    if (a.bias && ((uintptr_t)a.bias & 15)) return OG_E_ALIGN;
    const int tiles_m = (a.M + TOK - 1) / TOK;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    if (a.N > 64) {
        const int tiles_n = (a.N + 127) / 128;
        hipLaunchKernelGGL(gemm_nt_f16x3_kernel<128>, dim3(tiles_m8 * tiles_n), dim3(256), 0, stream, a, tiles_m, tiles_n);
    } else {
        hipLaunchKernelGGL(gemm_nt_f16x3_kernel<64>, dim3(tiles_m8), dim3(256), 0, stream, a, tiles_m, 1);
    }
    return og_launch_status();
}

int og_launch_split_f16(const float* x, int64_t n, void* hi, void* lo, hipStream_t stream) {
    if (!x || !hi || !lo || n <= 0) return OG_E_INVALID;
    if ((n & 3) || ((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return OG_E_ALIGN;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, n4, (_Float16*)hi,
                       (_Float16*)lo);
    return og_launch_status();
}

extern "C" int og_split_f16(const float* x, int64_t n, void* hi, void* lo, void* stream) {
    og_clear_status();
    return og_launch_split_f16(x, n, hi, lo, (hipStream_t)stream);
}

extern "C" int og_gemm_nt_f16x3(const void* Ah, const void* Al, int64_t lda, const void* Bh, const void* Bl, int64_t ldb,
                                int32_t M, int32_t N, int32_t K, const float* bias, int32_t relu, const float* res,
                                int64_t ldr, float* C32, int64_t ldc, void* Ch, void* Cl, int64_t ldch, void* stream) {
    og_clear_status();
    GemmHArgs g{};
    g.Ah = (const _Float16*)Ah; g.Al = (const _Float16*)Al; g.lda = lda;
    g.Bh = (const _Float16*)Bh; g.Bl = (const _Float16*)Bl; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu; g.res = res; g.ldr = ldr;
    g.C32 = C32; g.ldc = ldc; g.Ch = (_Float16*)Ch; g.Cl = (_Float16*)Cl; g.ldch = ldch;
    return og_launch_gemm_f16x3(g, (hipStream_t)stream);
}

// Exact-fp32 "NT" GEMM on the gfx950 matrix cores:  C = epilogue(A · Bᵀ),  A [M][K], B [N][K].
//
// Every 1x1 convolution of the reference (models/utils.py:52-57, attention_gnn.py:16-20,
// superglue.py:22) is a linear map on token rows, i.e. X[tokens][in] · W[out][in]ᵀ, and the score
// matrix (superglue.py:81-86) is g0[m][D] · g1[n][D]ᵀ -- the same NT shape with both operands
// K-contiguous, so one kernel serves all of them.
//
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-exact fmaf chain, 64 cycles/SIMD):
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  The k order inside a tile is
// free as long as A and B agree, so each lane reads FOUR consecutive k with one ds_read_b128
// (k = 8*kk + 4*(l>>5) + e) and feeds element e to the e-th MFMA.
//
// Tile: 128 x BN x 32, 4 waves as 2x2, each wave 64 x BN/2 (2 x BN/64 MFMA tiles).  LDS rows are
// padded to 36 floats: a 16-lane ds_read_b128 group then touches 16 distinct 4-bank slots
// (36*r mod 64 is a permutation of multiples of 4), i.e. conflict-free.
// Global -> register -> LDS staging with the next tile's loads in flight during the MFMAs.
// Block ids are remapped so that all N-tiles of one M-tile run on the same XCD (block b is
// dispatched to XCD b % 8): the A panel is then fetched into one L2 instead of up to eight.
#include "og_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDSW = BK + 4;   // padded LDS row, floats

template <int BN, class RD>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmArgs g, int tiles_m, int tiles_n, RD rd) {
    constexpr int TN = BN / 64;            // MFMA tiles per wave along N
    constexpr int BROWS = BN / 32;         // staging passes for B
    __shared__ __attribute__((aligned(16))) float As[BM * LDSW];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDSW];

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int tm = (local / tiles_n) * 8 + xcd;
    const int tn = local % tiles_n;
    if (tm >= tiles_m) return;
    const int z = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;

    const float* __restrict__ A = g.A + (int64_t)z * g.strideA;
    const float* __restrict__ B = g.B + (int64_t)z * g.strideB;
    if (rd.B > 0) {          // ragged: problem z = pair z, operands are row ranges of the packed token matrix
        g.M = rd.off0[z + 1] - rd.off0[z];
        g.N = rd.off1[z + 1] - rd.off1[z];
        if (m0 >= g.M || n0 >= g.N) return;
        A = g.A + (int64_t)rd.off0[z] * g.lda;
        B = g.B + (int64_t)(rd.off0[rd.B] + rd.off1[z]) * g.ldb;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3;          // 0..31: staging row within a 32-row pass
    const int lc4 = (tid & 7) * 4;      // staging column (floats)

    f32x4 ra[4], rb[BROWS];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = m0 + lrow + 32 * p;
            const int kk = k0 + lc4;
            if (row < g.M && kk < g.K) ra[p] = *reinterpret_cast<const f32x4*>(A + (int64_t)row * g.lda + kk);
            else ra[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < BROWS; ++p) {
            const int row = n0 + lrow + 32 * p;
            const int kk = k0 + lc4;
            if (row < g.N && kk < g.K) rb[p] = *reinterpret_cast<const f32x4*>(B + (int64_t)row * g.ldb + kk);
            else rb[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(&As[(lrow + 32 * p) * LDSW + lc4]) = ra[p];
#pragma unroll
        for (int p = 0; p < BROWS; ++p) *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * p) * LDSW + lc4]) = rb[p];
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = (wm * 64 + (lane & 31)) * LDSW + (lane >> 5) * 4;
    const int b_off = (wn * (BN / 2) + (lane & 31)) * LDSW + (lane >> 5) * 4;

    const int nk = (g.K + BK - 1) / BK;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[2], b[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[a_off + i * 32 * LDSW + kk * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[b_off + j * 32 * LDSW + kk * 8]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ----
    float* C = g.C + (int64_t)z * g.strideC;      // may alias R (in-place residual): no __restrict__
    const float* R = g.res ? g.res + (int64_t)z * g.strideR : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma clang fp contract(off)                  // og_split: hi and lo must see the same rounded value (og_common.h)
        const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
        const float al = g.alpha ? g.alpha[col] : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (g.relu == 1) v = fmaxf(v, 0.f);
                else if (g.relu == 2) v = sinf(30.f * v);      // Siren activation (reference models/utils.py:29)
                if (R) {
                    const float rr = R[(int64_t)row * g.ldr + col];
                    v = g.alpha ? al * v + (1.f - al) * rr : v + rr;
                }
                v *= g.scale;
                if (g.C) C[(int64_t)row * g.ldc + col] = v;
                if (g.Ch) {     // split-f16 copy for the f16x3 consumers (x = hi + lo)
                    const int64_t o = (int64_t)row * g.ldch + (g.c_hl ? og_hl_col(col) : (int64_t)col);
                    og_split(v, g.Ch[o], g.Cl[o]);
                }
                if (g.Ct) {
                    const int bz = row / g.ct_rows, ri = row - bz * g.ct_rows;
                    g.Ct[(int64_t)z * g.strideCt + (int64_t)bz * g.ldct * g.N + (int64_t)col * g.ldct + ri] = v;
                }
            }
        }
    }
}

}  // namespace

int og_launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (!a.A || !a.B || !(a.C || a.Ch) || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return OG_E_INVALID;
    if ((a.lda & 3) || (a.ldb & 3) || (a.K & 3)) return OG_E_ALIGN;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return OG_E_ALIGN;
    if ((a.strideA & 3) || (a.strideB & 3)) return OG_E_ALIGN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    GemmArgs k = a;
    k.rag = nullptr;
    auto launch = [&](auto rd) -> int {       // the per-pair descriptor is a kernel argument only for ragged launches (og_common.h)
        using RD = decltype(rd);
        if (a.N > 64) {
            const int tiles_n = (a.N + 127) / 128;
            hipLaunchKernelGGL((gemm_nt_f32_kernel<128, RD>), dim3(tiles_m8 * tiles_n, a.batch), dim3(256), 0, stream,
                               k, tiles_m, tiles_n, rd);
        } else {
            hipLaunchKernelGGL((gemm_nt_f32_kernel<64, RD>), dim3(tiles_m8, a.batch), dim3(256), 0, stream, k, tiles_m, 1, rd);
        }
        return og_launch_status();
    };
    return a.rag ? launch(*a.rag) : launch(RaggedNone{});
}

extern "C" int og_gemm_nt(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                          int64_t strideB, float* C, int64_t ldc, int64_t strideC, int32_t M, int32_t N,
                          int32_t K, int32_t batch, const float* bias, int32_t relu, const float* res,
                          int64_t ldr, const float* alpha, float scale, void* stream) {
    og_clear_status();
    GemmArgs g{};
    g.A = A; g.lda = lda; g.strideA = strideA;
    g.B = B; g.ldb = ldb; g.strideB = strideB;
    g.C = C; g.ldc = ldc; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.batch = batch;
    g.bias = bias; g.relu = relu;
    g.res = res; g.ldr = ldr; g.strideR = (int64_t)M * ldr;
    g.alpha = alpha; g.scale = scale;
    g.Ct = nullptr; g.ldct = 0; g.strideCt = 0; g.ct_rows = 1;
    g.Ch = nullptr; g.Cl = nullptr; g.ldch = 0; g.c_hl = 0; g.rag = nullptr;
    return og_launch_gemm(g, (hipStream_t)stream);
}

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
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDSW = BK + 4;   // padded LDS row, floats

// TA / TB: the operand is stored K-MAJOR (A[k][m] with row stride lda, B[k][n]) -- the layouts the backward products of a 1x1 conv
// and of attention come in (dW = dZ^T X: both operands [tokens][channels]; dX = dZ W: W [out][in]) -- so no transposed copy of a
// token-sized tensor is ever made.  Staged into LDS k-major ([32][tile + 4], float4 along the tile), fragments by four ds_read_b32.
template <int BM, int BN, bool TA, bool TB, class RD>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmArgs g, int tiles_m, int tiles_n, RD rd) {
    constexpr int TM = BM / 64;            // MFMA tiles per wave along M
    constexpr int TN = BN / 64;            // MFMA tiles per wave along N
    constexpr int AROWS = BM / 32;         // staging passes for A
    constexpr int BROWS = BN / 32;         // staging passes for B
    constexpr int LDTA = BM + 4, LDTB = BN + 4;     // k-major LDS rows
    constexpr int AS = TA ? (BK * LDTA > BM * LDSW ? BK * LDTA : BM * LDSW) : BM * LDSW;
    constexpr int BS = TB ? (BK * LDTB > BN * LDSW ? BK * LDTB : BN * LDSW) : BN * LDSW;
    __shared__ __attribute__((aligned(16))) float As[AS];
    __shared__ __attribute__((aligned(16))) float Bs[BS];

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    int tm = (local / tiles_n) * 8 + xcd;
    const int tn = local % tiles_n;
    int z = blockIdx.y;
    if constexpr (std::is_same<RD, RaggedNone>::value) {
        // uniform batch: ONE 1-D grid over the batch * tiles_m "virtual" M-tiles, dealt round-robin to the XCDs (workgroup b runs on
        // XCD b % 8; with a (tiles, batch) grid and tiles_m < 8 -- a split-K weight gradient has 2-4 -- only tiles_m of the 8 XCDs worked)
        z = tm / tiles_m;
        tm -= z * tiles_m;
        if (z >= g.batch) return;
    } else {
        if (tm >= tiles_m) return;
    }
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

    if (g.ktot > 0) {        // split-K: problem z contracts rows [z K, min((z+1) K, ktot)) of the k-major operands
        const int left = g.ktot - z * g.K;
        g.K = left < g.K ? (left > 0 ? left : 0) : g.K;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3;          // 0..31: staging row within a 32-row pass
    const int lc4 = (tid & 7) * 4;      // staging column (floats)
    // k-major staging: a k-row of the tile is BX / 4 float4; 256 threads cover 1024 / BX k-rows per pass, BX / 32 passes
    const int ta_k = tid / (BM / 4), ta_c4 = (tid % (BM / 4)) * 4;                      // A tile: BM wide
    constexpr int TA_KROWS = 1024 / BM;
    const int tb_k = tid / (BN / 4), tb_c4 = (tid % (BN / 4)) * 4;                      // B tile: BN wide
    constexpr int TB_KROWS = 1024 / BN;

    auto load_kmajor = [&](const float* __restrict__ P, int64_t ld, int krow, int col, int ncols) -> f32x4 {
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (krow < g.K && col < ncols) {
            const float* src = P + (int64_t)krow * ld + col;
            if (col + 3 < ncols) v = *reinterpret_cast<const f32x4*>(src);
            else
                for (int e = 0; e < 4; ++e)
                    if (col + e < ncols) v[e] = src[e];
        }
        return v;
    };

    f32x4 ra[AROWS], rb[BROWS];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            if constexpr (TA) {
                ra[p] = load_kmajor(A, g.lda, k0 + ta_k + TA_KROWS * p, m0 + ta_c4, g.M);
            } else {
                const int row = m0 + lrow + 32 * p;
                const int kk = k0 + lc4;
                if (row < g.M && kk < g.K) ra[p] = *reinterpret_cast<const f32x4*>(A + (int64_t)row * g.lda + kk);
                else ra[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int p = 0; p < BROWS; ++p) {
            if constexpr (TB) {
                rb[p] = load_kmajor(B, g.ldb, k0 + tb_k + TB_KROWS * p, n0 + tb_c4, g.N);
            } else {
                const int row = n0 + lrow + 32 * p;
                const int kk = k0 + lc4;
                if (row < g.N && kk < g.K) rb[p] = *reinterpret_cast<const f32x4*>(B + (int64_t)row * g.ldb + kk);
                else rb[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            if constexpr (TA) *reinterpret_cast<f32x4*>(&As[(ta_k + TA_KROWS * p) * LDTA + ta_c4]) = ra[p];
            else *reinterpret_cast<f32x4*>(&As[(lrow + 32 * p) * LDSW + lc4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < BROWS; ++p) {
            if constexpr (TB) *reinterpret_cast<f32x4*>(&Bs[(tb_k + TB_KROWS * p) * LDTB + tb_c4]) = rb[p];
            else *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * p) * LDSW + lc4]) = rb[p];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = TA ? (lane >> 5) * 4 * LDTA + wm * (BM / 2) + (lane & 31) : (wm * (BM / 2) + (lane & 31)) * LDSW + (lane >> 5) * 4;
    const int b_off = TB ? (lane >> 5) * 4 * LDTB + wn * (BN / 2) + (lane & 31) : (wn * (BN / 2) + (lane & 31)) * LDSW + (lane >> 5) * 4;

    const int nk = (g.K + BK - 1) / BK;
    float asum = 0.f;                   // g.a_colsum: thread t < 128 sums column m0 + t of the k-major A tile over this problem's k range
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        if constexpr (TA) {
            if (g.a_colsum && tn == 0 && tid < BM) {
#pragma unroll
                for (int r = 0; r < BK; ++r) asum += As[r * LDTA + tid];
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (TA) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = As[a_off + (kk * 8 + e) * LDTA + i * 32];
                } else {
                    a[i] = *reinterpret_cast<const f32x4*>(&As[a_off + i * 32 * LDSW + kk * 8]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (TB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bs[b_off + (kk * 8 + e) * LDTB + j * 32];
                } else {
                    b[j] = *reinterpret_cast<const f32x4*>(&Bs[b_off + j * 32 * LDSW + kk * 8]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ----
    float* C = g.C + (int64_t)z * g.strideC;      // may alias R (in-place residual): no __restrict__
    if constexpr (TA) {
        if (g.a_colsum && tn == 0 && tid < BM && m0 + tid < g.M) C[(int64_t)(m0 + tid) * g.ldc + g.N] = asum * g.scale;
    }
    const float* R = g.res ? g.res + (int64_t)z * g.strideR : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma clang fp contract(off)                  // og_split: hi and lo must see the same rounded value (og_common.h)
        const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
        const float al = g.alpha ? g.alpha[col] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + mfma32_row(r, lane);
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

// The partial products of a split-K weight gradient (og_gemm_kmajor, batch = parts, a_colsum) summed: dW[r][c] = sum_p part[p][r][c] (c < cols),
// db[r] = sum_p part[p][r][cols] -- one launch instead of a reduction plus two strided copies.  A workgroup owns 64 float4 columns; its four
// waves take the parts p = w, w + 4, ... (four independent loads in flight per thread) and meet in LDS in wave order: the summation order is fixed.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int parts, int rows, int64_t ld, int cols,
                                                            float* __restrict__ dW, float* __restrict__ db) {
    __shared__ f32x4 red[3][64];
    const int c4n = cols / 4 + (db ? 1 : 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < (int64_t)rows * c4n;
    const int r = live ? (int)(i / c4n) : 0, c = live ? (int)(i % c4n) * 4 : 0;
    const float* src = part + (int64_t)r * ld + c;
    const int64_t sp = (int64_t)rows * ld;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    if (live) {
        int p = wave;
        for (; p + 12 < parts; p += 16) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (int64_t)p * sp), v1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(p + 4) * sp);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(p + 8) * sp), v3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(p + 12) * sp);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; p < parts; p += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)p * sp);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave || !live) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const f32x4 v = red[w][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    if (c < cols) *reinterpret_cast<f32x4*>(dW + (int64_t)r * cols + c) = acc;
    else db[r] = acc[0];
}

}  // namespace

extern "C" int og_splitk_reduce(const float* part, int32_t parts, int32_t rows, int64_t ld, int32_t cols, float* dW, float* db, void* stream) {
    og_clear_status();
    if (!part || !dW || parts <= 0 || rows <= 0 || cols <= 0) return OG_E_INVALID;
    if ((cols & 3) || (ld & 3) || ld < cols + (db ? 4 : 0) || ((uintptr_t)part & 15) || ((uintptr_t)dW & 15)) return OG_E_ALIGN;
    const int64_t n = (int64_t)rows * (cols / 4 + (db ? 1 : 0));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, part, parts, rows, ld, cols, dW, db);
    return og_launch_status();
}

int og_launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (!a.A || !a.B || !(a.C || a.Ch) || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return OG_E_INVALID;
    if ((a.lda & 3) || (a.ldb & 3)) return OG_E_ALIGN;
    if ((!a.ta || !a.tb) && (a.K & 3)) return OG_E_ALIGN;          // a K-contiguous operand is read in float4 along K
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return OG_E_ALIGN;
    if ((a.strideA & 3) || (a.strideB & 3)) return OG_E_ALIGN;
    if (a.ta && !a.tb) return OG_E_SHAPE;                    // forms: NT (default), A normal x B k-major, both k-major
    if ((a.ta || a.tb) && (a.rag || a.Ch || a.Ct)) return OG_E_SHAPE;
    if (a.ktot > 0 && !(a.ta && a.tb)) return OG_E_SHAPE;
    if (a.a_colsum && (!a.ta || a.ldc <= a.N)) return OG_E_SHAPE;
    // Tile shape: 128 x 64, or 64 x 64 where the former gives fewer than 1024 workgroups (four per CU); OG_GEMM_F32_BN=128 asks for the 128 x 128 form,
    // OG_GEMM_F32_BM=64 / 128 forces the tile height (experiments).  A workgroup is four waves = ONE per SIMD, and the k loop has two barriers per
    // 32-deep tile with nothing else to run between them -- what hides them is a second and third workgroup on the CU, and the smaller tiles are
    // what provides those: the training step's convs are 4096 / 8192 token rows x 256 ... 768 channels = 64 ... 384 workgroups of 128 x 128 on 256
    // CUs.  The 128 x 64 form reads 1.5x the operand bytes per flop (L2 hits) and was faster or equal wherever it was measured
    // (profiles/r05_v_*): training step 39.3 -> 32.3 ms at 4 pairs, 81.4 -> 75.0 ms at 16 pairs (512 ... 1536 wide workgroups per launch), the
    // encoder convs of the inference path 0.093 -> 0.068 ms at C2; 64 x 64 below 1024 workgroups: a 4096 x 256 x 256 launch 19.0 -> 12.3 us, the
    // training step 28.9 -> 26.0 ms at 4 pairs, no change at 16 (profiles/r05_x_*).
    static const int force_bn = [] { const char* e = getenv("OG_GEMM_F32_BN"); return e ? atoi(e) : 0; }();
    static const int force_bm = [] { const char* e = getenv("OG_GEMM_F32_BM"); return e ? atoi(e) : 0; }();
    static const int64_t short_below = [] { const char* e = getenv("OG_GEMM_F32_SHORT_BELOW"); return e ? atoll(e) : 1024LL; }();
    const bool narrow = a.N <= 64 || force_bn != 128;
    const int64_t wgs_tall = (int64_t)((a.M + 127) / 128) * a.batch * ((a.N + 63) / 64);
    const bool shortt = narrow && (force_bm == 64 ? true : force_bm == 128 ? false : wgs_tall < short_below);
    const int bm = shortt ? 64 : 128;
    const int tiles_m = (a.M + bm - 1) / bm;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    const int64_t vtiles8 = ((int64_t)tiles_m * a.batch + 7) / 8 * 8;      // uniform batches: see the kernel
    if (vtiles8 * ((a.N + 63) / 64) > 0x7fffffffLL) return OG_E_SHAPE;
    GemmArgs k = a;
    k.rag = nullptr;
    const int tiles_n = narrow ? (a.N + 63) / 64 : (a.N + 127) / 128;
    auto launch = [&](auto rd) -> int {       // the per-pair descriptor is a kernel argument only for ragged launches (og_common.h)
        using RD = decltype(rd);
        constexpr bool uniform = std::is_same<RD, RaggedNone>::value;
        const dim3 grid = uniform ? dim3((unsigned)(vtiles8 * tiles_n)) : dim3(tiles_m8 * tiles_n, a.batch);
        if (!narrow) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 128, false, false, RD>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, rd);
        else if (!shortt) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 64, false, false, RD>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, rd);
        else hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, false, false, RD>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, rd);
        return og_launch_status();
    };
    auto launch_t = [&](auto ta, auto tb) -> int {
        constexpr bool TA_ = decltype(ta)::value, TB_ = decltype(tb)::value;
        const dim3 grid((unsigned)(vtiles8 * tiles_n));
        if (!narrow) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 128, TA_, TB_, RaggedNone>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, RaggedNone{});
        else if (!shortt) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 64, TA_, TB_, RaggedNone>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, RaggedNone{});
        else hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, TA_, TB_, RaggedNone>), grid, dim3(256), 0, stream, k, tiles_m, tiles_n, RaggedNone{});
        return og_launch_status();
    };
    if (a.ta) return launch_t(std::true_type{}, std::true_type{});
    if (a.tb) return launch_t(std::false_type{}, std::true_type{});
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

extern "C" int og_gemm_kmajor(const float* A, int64_t lda, int64_t strideA, int32_t a_kmajor, const float* B, int64_t ldb, int64_t strideB,
                              float* C, int64_t ldc, int64_t strideC, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t k_total,
                              int32_t a_colsum, float scale, void* stream) {
    og_clear_status();
    GemmArgs g{};
    g.A = A; g.lda = lda; g.strideA = strideA;
    g.B = B; g.ldb = ldb; g.strideB = strideB;
    g.C = C; g.ldc = ldc; g.strideC = strideC;
    g.M = M; g.N = N; g.K = K; g.batch = batch;
    g.scale = scale; g.ct_rows = 1;
    g.ta = a_kmajor ? 1 : 0; g.tb = 1; g.ktot = k_total; g.a_colsum = a_colsum ? 1 : 0;
    return og_launch_gemm(g, (hipStream_t)stream);
}

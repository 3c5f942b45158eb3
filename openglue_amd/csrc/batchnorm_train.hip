// Train-mode BatchNorm statistics + normalisation for the token-major activations of the MLPs (training slice).
//
// Replaces nn.BatchNorm1d in training mode inside FeedForwardNet (reference models/utils.py:48-58: Conv1d -> ReLU ->
// BatchNorm1d): per channel c over all tokens t of the call (the reference's [B, C, N] tensor: statistics over B and N)
//     mean_c = E_t x[t][c]      var_c = E_t (x[t][c] - mean_c)^2  (biased)      y = (x - mean) / sqrt(var + eps) * w + b
//     running_mean = (1 - mom) running_mean + mom mean        running_var = (1 - mom) running_var + mom var * T / (T - 1)
// (torch.nn.functional.batch_norm semantics).  Eval mode needs none of this: the running statistics are folded into the next
// 1x1 conv at pack time (api.hip).
// HBM-bound, three passes over x: (1) per-row-slab partial sums of (x - k) and (x - k)^2 with k = x[0][c] (a per-channel shift
// removes the cancellation of E x^2 - (E x)^2 for channels whose mean dominates their spread), coalesced along the channel
// axis; (2) one block folds the partials in double, updates the running statistics and emits scale = w * invstd,
// shift = b - mean * scale (+ mean / invstd for a later backward); (3) y = x * scale + shift, 16 bytes per lane.
#include "og_common.h"

namespace {

constexpr int BN_ROWS_PER_BLOCK = 32;       // rows folded by one block of pass 1 (round 5: 256 -> 32; 8192 x 512 activations were 256 blocks whose
                                            // threads each walked 64 rows one 4-byte load at a time: 18 us for 16 MB)

// Pass 1.  grid (ceil(rows / BN_ROWS_PER_BLOCK), ceil(C / 256)), 256 threads = 4 waves; a lane owns FOUR channels (one 16-byte load per row: a wave
// reads 1 KB of a row), wave w the rows r0 + w, r0 + w + 4, ...; the waves meet in LDS in wave order.  part[(blk * C + c) * 2 + {0, 1}].
// F(lane's 4 values of this row, lane's 4 channels) -> the two terms to accumulate
template <class F>
__device__ __forceinline__ void bn_fold_rows(int64_t rows, int C, float* __restrict__ part, F term) {
    __shared__ f32x4 red[2][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
    const int64_t r1 = r0 + BN_ROWS_PER_BLOCK < rows ? r0 + BN_ROWS_PER_BLOCK : rows;
    f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
#pragma unroll 4
        for (int64_t r = r0 + wave; r < r1; r += 4) term(r, c, s1, s2);
    }
    if (wave) { red[0][wave - 1][lane] = s1; red[1][wave - 1][lane] = s2; }
    __syncthreads();
    if (wave || c >= C) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const f32x4 a = red[0][w][lane], b = red[1][w][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += a[e]; s2[e] += b[e]; }
    }
    float* dst = part + ((int64_t)blockIdx.x * C + c) * 2;
    *reinterpret_cast<f32x4*>(dst) = f32x4{s1[0], s2[0], s1[1], s2[1]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{s1[2], s2[2], s1[3], s2[3]};
}

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C, float* __restrict__ part) {
    const int c0 = blockIdx.y * 256 + (threadIdx.x & 63) * 4;
    const f32x4 k = c0 < C ? *reinterpret_cast<const f32x4*>(x + c0) : f32x4{0.f, 0.f, 0.f, 0.f};      // shift: row 0 of the channel
    bn_fold_rows(rows, C, part, [&](int64_t r, int c, f32x4& s1, f32x4& s2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[e] - k[e];
            s1[e] += d;
            s2[e] = fmaf(d, d, s2[e]);
        }
    });
}

// The per-channel fold of the partials, in double: 64 channels per block, the four waves take the slabs i = w, w + 4, ... and meet in LDS in wave order
// (one thread per channel walking every slab was fine at 32 slabs; there are 256 now).  -> (s1, s2) in the threads of wave 0, others return false.
__device__ __forceinline__ bool bn_fold_partials(const float* __restrict__ part, int nblk, int C, int c, double& s1, double& s2) {
    __shared__ double red[2][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s1 = 0.0; s2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int i = wave; i < nblk; i += 4) {
            const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)i * C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    }
    if (wave) { red[0][wave - 1][lane] = s1; red[1][wave - 1][lane] = s2; }
    __syncthreads();
    if (wave || c >= C) return false;
#pragma unroll
    for (int w = 0; w < 3; ++w) { s1 += red[0][w][lane]; s2 += red[1][w][lane]; }
    return true;
}

// grid ceil(C / 64), 256 threads
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ part, int nblk, int64_t rows, int C,
                                                          const float* __restrict__ w, const float* __restrict__ b, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ scale_shift, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s1, s2;
    if (!bn_fold_partials(part, nblk, C, c, s1, s2)) return;
    const double n = (double)rows;
    const double dm = s1 / n;                                   // mean - k
    const double mean = (double)x[c] + dm;
    double var = s2 / n - dm * dm;                              // biased
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((w ? (double)w[c] : 1.0) * invstd);
    scale_shift[c] = sc;
    scale_shift[C + c] = (float)((b ? (double)b[c] : 0.0) - mean * (double)sc);
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    if (running_var) {
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// C % 4 == 0, 16-byte aligned rows: one float4 per thread
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C4, const float* __restrict__ scale_shift,
                                                       float* __restrict__ y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale_shift + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(scale_shift + 4 * C4 + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], sc[e], sh[e]);
    *reinterpret_cast<f32x4*>(y + r * ldy + c) = o;
}

}  // namespace

// include/openglue_amd.h
extern "C" size_t og_batchnorm_train_workspace_bytes(int64_t rows, int32_t channels) {
    if (rows < 1 || channels < 4 || (channels & 3)) return 0;
    const int64_t nblk = (rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK;
    return (size_t)((nblk * channels * 2 + 3 * channels) * (int64_t)sizeof(float));     // partials + scale/shift (forward) or 3 coefficient rows (backward)
}

extern "C" int og_batchnorm_train_forward(const float* x, int64_t ldx, int64_t rows, int32_t channels, const float* weight, const float* bias, float eps,
                                          float momentum, float* running_mean, float* running_var, float* y, int64_t ldy, float* save_mean,
                                          float* save_invstd, void* workspace, void* stream) {
    og_clear_status();
    if (!x || !y || !workspace || rows < 1 || channels < 4 || !(eps > 0.f) || !(momentum >= 0.f && momentum <= 1.f)) return OG_E_INVALID;
    if ((channels & 3) || (ldx & 3) || (ldy & 3) || ldx < channels || ldy < channels) return OG_E_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)workspace & 15)) return OG_E_ALIGN;
    if (rows * (int64_t)(channels / 4) > ((int64_t)1 << 40)) return OG_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)((rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK);
    float* part = reinterpret_cast<float*>(workspace);
    float* scale_shift = part + (int64_t)nblk * channels * 2;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblk, (channels + 255) / 256), dim3(256), 0, st, x, ldx, rows, channels, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((channels + 63) / 64), dim3(256), 0, st, x, part, nblk, rows, channels, weight, bias, eps, momentum,
                       running_mean, running_var, scale_shift, save_mean, save_invstd);
    const int64_t n4 = rows * (channels / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, ldx, rows, channels / 4, scale_shift, y, ldy);
    return og_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the train-mode block  a = relu(z) ; y = batchnorm_train(a)  (models/utils.py:52-56, autograd of
// torch.nn.functional.batch_norm(training=True) and relu):
//     xhat = (a - mean) invstd        dbias_c = sum_t dy        dweight_c = sum_t dy xhat
//     da = (w invstd / T) (T dy - dbias - xhat dweight)         dz = da * [a > 0]   (relu_mask: `a` IS the ReLU output)
// Same three passes as the forward: partial sums per row slab, one fold in double, one elementwise pass.
namespace {

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ dy, int64_t lddy,
                                                             int64_t rows, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float* __restrict__ part) {
    const int c0 = blockIdx.y * 256 + (threadIdx.x & 63) * 4;
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    const f32x4 mu = c0 < C ? *reinterpret_cast<const f32x4*>(mean + c0) : zero, is = c0 < C ? *reinterpret_cast<const f32x4*>(invstd + c0) : zero;
    bn_fold_rows(rows, C, part, [&](int64_t r, int c, f32x4& s1, f32x4& s2) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + r * lda + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1[e] += g[e];
            s2[e] = fmaf(g[e], (v[e] - mu[e]) * is[e], s2[e]);
        }
    });
}

// coef[c] = w invstd, coef[C + c] = dbias / T, coef[2C + c] = dweight / T;  grid ceil(C / 64), 256 threads
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int64_t rows, int C, const float* __restrict__ w,
                                                              const float* __restrict__ invstd, float* __restrict__ dweight, float* __restrict__ dbias,
                                                              float* __restrict__ coef) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s1, s2;
    if (!bn_fold_partials(part, nblk, C, c, s1, s2)) return;
    if (dbias) dbias[c] = (float)s1;
    if (dweight) dweight[c] = (float)s2;
    coef[c] = (w ? w[c] : 1.f) * invstd[c];
    coef[C + c] = (float)(s1 / (double)rows);
    coef[2 * C + c] = (float)(s2 / (double)rows);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ dy, int64_t lddy, int64_t rows,
                                                           int C4, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, int relu_mask, float* __restrict__ dz, int64_t lddz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4, C = 4 * C4;
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + r * lda + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + c), k1 = *reinterpret_cast<const f32x4*>(coef + C + c);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xhat = (av[e] - mu[e]) * is[e];
        const float d = k0[e] * (g[e] - k1[e] - xhat * k2[e]);
        o[e] = (relu_mask && !(av[e] > 0.f)) ? 0.f : d;
    }
    *reinterpret_cast<f32x4*>(dz + r * lddz + c) = o;
}

// dst[c][r] = src[r][c]: 64 x 64 tiles through LDS (the weight-gradient GEMM wants both operands K-contiguous: dW = dZ^T X)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int64_t lds_, int64_t rows, int cols, float* __restrict__ dst,
                                                            int64_t ldd, int64_t stride_src, int64_t stride_dst) {
    __shared__ float tile[64][65];
    src += (int64_t)blockIdx.z * stride_src;
    dst += (int64_t)blockIdx.z * stride_dst;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const int64_t r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = tile[tx][i];
    }
}

// out[c] = sum_r x[r][c] (bias gradient): per-slab partials + one fold in double
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
    float s1 = 0.f;
    if (c < C) {
        const int64_t r1 = r0 + BN_ROWS_PER_BLOCK < rows ? r0 + BN_ROWS_PER_BLOCK : rows;
        for (int64_t r = r0 + rg; r < r1; r += 4) s1 += x[r * ldx + c];
    }
    red[rg][threadIdx.x & 63] = s1;
    __syncthreads();
    if (rg == 0 && c < C) {
        const int l = threadIdx.x;
        part[(int64_t)blockIdx.x * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += (double)part[(int64_t)i * C + c];
    out[c] = (float)s;
}

}  // namespace

extern "C" int og_batchnorm_train_backward(const float* a, int64_t lda, const float* dy, int64_t lddy, int64_t rows, int32_t channels, const float* weight,
                                           const float* save_mean, const float* save_invstd, int32_t relu_mask, float* dz, int64_t lddz,
                                           float* dweight, float* dbias, void* workspace, void* stream) {
    og_clear_status();
    if (!a || !dy || !dz || !save_mean || !save_invstd || !workspace || rows < 1 || channels < 4) return OG_E_INVALID;
    if ((channels & 3) || (lda & 3) || (lddy & 3) || (lddz & 3) || lda < channels || lddy < channels || lddz < channels) return OG_E_SHAPE;
    if (((uintptr_t)a & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dz & 15) || ((uintptr_t)workspace & 15) || ((uintptr_t)save_mean & 15) ||
        ((uintptr_t)save_invstd & 15)) return OG_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)((rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK);
    float* part = reinterpret_cast<float*>(workspace);
    float* coef = part + (int64_t)nblk * channels * 2;       // 3 * channels floats: og_batchnorm_train_workspace_bytes reserves them
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk, (channels + 255) / 256), dim3(256), 0, st, a, lda, dy, lddy, rows, channels, save_mean, save_invstd, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((channels + 63) / 64), dim3(256), 0, st, part, nblk, rows, channels, weight, save_invstd, dweight,
                       dbias, coef);
    const int64_t n4 = rows * (channels / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, lda, dy, lddy, rows, channels / 4, save_mean,
                       save_invstd, coef, relu_mask, dz, lddz);
    return og_launch_status();
}

extern "C" int og_transpose_f32(const float* src, int64_t ld_src, int64_t rows, int32_t cols, float* dst, int64_t ld_dst, void* stream) {
    og_clear_status();
    if (!src || !dst || rows < 1 || cols < 1 || ld_src < cols || ld_dst < rows) return OG_E_INVALID;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((rows + 63) / 64), (cols + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows,
                       cols, dst, ld_dst, (int64_t)0, (int64_t)0);
    return og_launch_status();
}

extern "C" int og_transpose_f32_batched(const float* src, int64_t ld_src, int64_t stride_src, int64_t rows, int32_t cols, float* dst, int64_t ld_dst,
                                        int64_t stride_dst, int32_t batch, void* stream) {
    og_clear_status();
    if (!src || !dst || rows < 1 || cols < 1 || batch < 1 || batch > 65535 || ld_src < cols || ld_dst < rows) return OG_E_INVALID;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((rows + 63) / 64), (cols + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       rows, cols, dst, ld_dst, stride_src, stride_dst);
    return og_launch_status();
}

extern "C" int og_colsum_f32(const float* x, int64_t ldx, int64_t rows, int32_t channels, float* out, void* workspace, void* stream) {
    og_clear_status();
    if (!x || !out || !workspace || rows < 1 || channels < 1 || ldx < channels) return OG_E_INVALID;
    const int nblk = (int)((rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, (channels + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, channels, (float*)workspace);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((channels + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, channels, out);
    return og_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Training-mode attention with the attention matrix MATERIALISED, like the reference (attention.py:8-19) -- autograd needs P:
//   forward   P = softmax_rows(S)                  (S = scale * Q K^T from the batched exact-fp32 GEMM, in place)
//   backward  dS = scale * P o (dP - sum_j dP o P) (dP = dO V^T from the same GEMM, in place)
// One wave per row, columns [cols, ld) are zeroed (the matrices are the K operand of the next GEMM, K padded to a multiple of 4).
namespace {

__device__ __forceinline__ float bn_wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float bn_wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int64_t ld, int64_t rows, int cols) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float* row = S + r * ld;
    float mx = OG_NEG_INF;
    for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, row[j]);
    mx = bn_wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    sum = bn_wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < cols; j += 64) row[j] *= inv;
    for (int j = cols + lane; j < ld; j += 64) row[j] = 0.f;
}

__global__ __launch_bounds__(256) void softmax_rows_backward_kernel(const float* __restrict__ P, float* __restrict__ dP, int64_t ld, int64_t rows, int cols,
                                                                    float scale) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + r * ld;
    float* d = dP + r * ld;
    float dot = 0.f;
    for (int j = lane; j < cols; j += 64) dot = fmaf(d[j], p[j], dot);
    dot = bn_wave_sum(dot);
    for (int j = lane; j < cols; j += 64) d[j] = scale * p[j] * (d[j] - dot);
    for (int j = cols + lane; j < ld; j += 64) d[j] = 0.f;
}

}  // namespace

extern "C" int og_softmax_rows(float* S, int64_t ld, int64_t rows, int32_t cols, void* stream) {
    og_clear_status();
    if (!S || rows < 1 || cols < 1 || ld < cols) return OG_E_INVALID;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, ld, rows, cols);
    return og_launch_status();
}

extern "C" int og_softmax_rows_backward(const float* P, float* dP, int64_t ld, int64_t rows, int32_t cols, float scale, void* stream) {
    og_clear_status();
    if (!P || !dP || rows < 1 || cols < 1 || ld < cols) return OG_E_INVALID;
    hipLaunchKernelGGL(softmax_rows_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, ld, rows, cols, scale);
    return og_launch_status();
}

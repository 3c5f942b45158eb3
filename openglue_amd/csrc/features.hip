// The steps immediately BEFORE and AFTER the matcher in OpenGlue's inference loop:
//
//  * og_prepare_features: models/features/utils.py:54-65 `prepare_features_output` with the LAF -> side-info
//    converters of models/laf_converter.py:22-128 -- keypoints = LAF centre column, side_info = [response
//    (optionally log(response + 0.1)), log scale, orientation / affine shape normalised by the scale].
//    The LAF scale is kornia.feature.laf.get_laf_scale (third-party, kornia>=0.6.1, NOT vendored in the
//    reference): sqrt(|a00*a11 - a10*a01 + 1e-10|), restated from its published source.
//  * og_compact_matches: inference.py:192-209 -- the valid matches of a batch in (pair, keypoint) order with
//    their confidence, the matched LAFs of both images and the keypoint centres (kornia get_laf_center =
//    LAF[..., 2]).  Order-preserving compaction: per-256 counts, one scan block, scatter.
#include "og_common.h"

namespace {

// method: 0 none, 1 scale, 2 rotation, 3 scale_rotation, 4 affine   (laf_converter.py:106-128)
__device__ __forceinline__ int side_dim(int method) { return method == 0 ? 1 : method == 1 ? 2 : method == 2 ? 3 : method == 3 ? 4 : 6; }

__global__ __launch_bounds__(256) void prepare_features_kernel(const float* __restrict__ lafs, const float* __restrict__ resp,
                                                               int64_t tokens, int method, int log_response,
                                                               float* __restrict__ kpts, float* __restrict__ side) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tokens) return;
    const float* L = lafs + t * 6;            // [[a00 a01 x], [a10 a11 y]]
    const float a00 = L[0], a01 = L[1], x = L[2], a10 = L[3], a11 = L[4], y = L[5];
    kpts[2 * t] = x; kpts[2 * t + 1] = y;
    const int s = side_dim(method);
    float* o = side + t * s;
    float r = resp[t];
    if (log_response) r = logf(r + 0.1f);      // features/utils.py:58-59
    o[0] = r;
    if (method == 0) return;
    // kornia get_laf_scale; separately rounded products like the torch expression (no fma contraction: the
    // determinant cancels for thin frames and a contracted form differs in the last bits, amplified by 1/scale)
    const float scale = sqrtf(fabsf(__fadd_rn(__fsub_rn(__fmul_rn(a00, a11), __fmul_rn(a10, a01)), 1e-10f)));
    int c = 1;
    if (method == 1 || method == 3 || method == 4) o[c++] = logf(scale);  // LAF2LogScale
    if (method == 2 || method == 3) { o[c++] = a01 / scale; o[c++] = a00 / scale; }   // flip(lafs[..., 0, :-1]) / scale
    if (method == 4) { o[c++] = a00 / scale; o[c++] = a01 / scale; o[c++] = a10 / scale; o[c++] = a11 / scale; }
}

__global__ __launch_bounds__(256) void compact_count_kernel(const int64_t* __restrict__ matches0, int64_t total,
                                                            int* __restrict__ block_counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool v = i < total && matches0[i] >= 0;
    const unsigned long long bal = __ballot(v);
    __shared__ int wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// exclusive scan of the block counts by ONE workgroup (<= a few thousand blocks); total -> count_out
__global__ __launch_bounds__(256) void compact_scan_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ count_out) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (nblocks + 255) / 256;
    int s = 0;
    for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < nblocks) s += block_counts[i]; }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = acc; acc += v; } *count_out = acc; }
    __syncthreads();
    int acc = part[tid];
    for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < nblocks) { const int v = block_counts[i]; block_counts[i] = acc; acc += v; } }
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(const int64_t* __restrict__ matches0, const float* __restrict__ ms0,
                                                              const float* __restrict__ lafs0, const float* __restrict__ lafs1,
                                                              int64_t total, int M, int N, const int* __restrict__ block_off,
                                                              int64_t* __restrict__ idxs, int64_t* __restrict__ batch,
                                                              float* __restrict__ conf, float* __restrict__ ml0, float* __restrict__ ml1,
                                                              float* __restrict__ k0, float* __restrict__ k1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t mj = i < total ? matches0[i] : -1;
    const bool v = mj >= 0;
    const unsigned long long bal = __ballot(v);
    __shared__ int wc[4];
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    if (!v) return;
    int pos = block_off[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += wc[w];
    const int64_t b = i / M, q = i - b * M;
    idxs[2 * (int64_t)pos] = q; idxs[2 * (int64_t)pos + 1] = mj;
    batch[pos] = b;
    conf[pos] = ms0[i];
    if (lafs0 && lafs1) {
        const float* A = lafs0 + (b * M + q) * 6;
        const float* Bp = lafs1 + (b * N + mj) * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) { ml0[(int64_t)pos * 6 + e] = A[e]; ml1[(int64_t)pos * 6 + e] = Bp[e]; }
        k0[2 * (int64_t)pos] = A[2]; k0[2 * (int64_t)pos + 1] = A[5];
        k1[2 * (int64_t)pos] = Bp[2]; k1[2 * (int64_t)pos + 1] = Bp[5];
    }
}

}  // namespace

extern "C" int og_prepare_features(const float* lafs, const float* responses, int64_t tokens, int32_t method, int32_t log_response,
                                   float* keypoints, float* side_info, void* stream) {
    og_clear_status();
    if (!lafs || !responses || !keypoints || !side_info || tokens <= 0) return OG_E_INVALID;
    if (method < 0 || method > 4) return OG_E_FLAG;
    hipLaunchKernelGGL(prepare_features_kernel, dim3((unsigned)((tokens + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lafs,
                       responses, tokens, method, log_response, keypoints, side_info);
    return og_launch_status();
}

extern "C" size_t og_compact_workspace_bytes(int32_t batch, int32_t m) {
    if (batch <= 0 || m <= 0) return 0;
    return sizeof(int) * (((size_t)batch * m + 255) / 256 + 4);
}

extern "C" int og_compact_matches(const int64_t* matches0, const float* matching_scores0, const float* lafs0, const float* lafs1,
                                  int32_t batch, int32_t m, int32_t n, int64_t* matching_idxs, int64_t* batch_indexes,
                                  float* confidence, float* mlafs0, float* mlafs1, float* keypoints0, float* keypoints1,
                                  int32_t* count_dev, void* workspace_dev, void* stream) {
    og_clear_status();
    if (!matches0 || !matching_scores0 || !matching_idxs || !batch_indexes || !confidence || !count_dev || !workspace_dev) return OG_E_INVALID;
    if (batch <= 0 || m <= 0 || n <= 0) return OG_E_INVALID;
    if ((lafs0 == nullptr) != (lafs1 == nullptr)) return OG_E_INVALID;
    if (lafs0 && (!mlafs0 || !mlafs1 || !keypoints0 || !keypoints1)) return OG_E_INVALID;
    const int64_t total = (int64_t)batch * m;
    const int nblocks = (int)((total + 255) / 256);
    int* bc = (int*)workspace_dev;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, st, matches0, total, bc);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, st, bc, nblocks, count_dev);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblocks), dim3(256), 0, st, matches0, matching_scores0, lafs0, lafs1, total, m, n,
                       bc, matching_idxs, batch_indexes, confidence, mlafs0, mlafs1, keypoints0, keypoints1);
    return og_launch_status();
}

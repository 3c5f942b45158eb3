// Mutual-nearest-neighbour match extraction from the log-assignment matrix, and the keypoint
// encoder's input preparation.
//
// Match extraction replaces models/matching_module.py:174-187 (matches0 / matching_scores0) and
// inference.py:176-190 (matches1 / matching_scores1):
//   max0/idx0 = row max / argmax of scores[:, :-1, :-1],  max1/idx1 = column max / argmax,
//   mutual0[i] = (idx1[idx0[i]] == i),  ms0 = mutual0 ? exp(max0) : 0,  valid0 = mutual0 & (ms0 > thr),
//   matches0 = valid0 ? idx0 : -1 ; image-1 side symmetrically through gathers.
// torch.max returns the FIRST maximal index on ties; every reduction below keeps "greater value, or
// equal value and lower index".
#include "og_common.h"

namespace {

struct MatchWs {
    int* idx0;                    // [B][m]
    float* max0;                  // [B][m]
    unsigned long long* colbest;  // [B][n] packed (orderable(value) << 32) | (0xFFFFFFFF - row)
};

static MatchWs mw_layout(void* ws, int B, int m, int n) {
    MatchWs w{};
    char* p = (char*)ws;
    w.colbest = (unsigned long long*)p; p += sizeof(unsigned long long) * (size_t)B * n;
    w.idx0 = (int*)p; p += sizeof(int) * (size_t)B * m;
    w.max0 = (float*)p;
    return w;
}

__device__ __forceinline__ unsigned int orderable(float v) {
    const unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// one wave per row i < m
template <class RD>
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ scores, int M, int N,
                                                         int* __restrict__ idx0, float* __restrict__ max0, RD rd) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int Mmax = M;                          // workspace stride
    int64_t so = (int64_t)b * (M + 1) * (N + 1);
    if (rd.B > 0) { M = rd.off0[b + 1] - rd.off0[b]; N = rd.off1[b + 1] - rd.off1[b]; so = rd.soff[b]; }
    if (row >= M) return;
    const float* sp = scores + so + (int64_t)row * (N + 1);
    float best = OG_NEG_INF;
    int bi = 0x7FFFFFFF;
    for (int j = lane; j < N; j += 64) {
        const float v = sp[j];
        if (v > best || bi == 0x7FFFFFFF) { best = v; bi = j; }     // ascending j: strict > keeps the first
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { idx0[(int64_t)b * Mmax + row] = bi; max0[(int64_t)b * Mmax + row] = best; }
}

// grid (ceil(n/256), ceil(m/64), B): thread = one column over a 64-row slab, then one 64-bit atomicMax
template <class RD>
__global__ __launch_bounds__(256) void col_argmax_kernel(const float* __restrict__ scores, int M, int N,
                                                         unsigned long long* __restrict__ colbest, RD rd) {
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int Nmax = N;
    int64_t so = (int64_t)b * (M + 1) * (N + 1);
    if (rd.B > 0) { M = rd.off0[b + 1] - rd.off0[b]; N = rd.off1[b + 1] - rd.off1[b]; so = rd.soff[b]; }
    const int i0 = blockIdx.y * 64;
    if (j >= N || i0 >= M) return;
    const int i1 = min(i0 + 64, M);
    const float* sp = scores + so + j;
    float best = sp[(int64_t)i0 * (N + 1)];
    int bi = i0;
    int i = i0 + 1;
    // eight rows in flight per thread (a thread's loads are 4 bytes each: one row at a time left the memory pipe mostly idle -- 3.2 TB/s at 4096 x 4096);
    // compared in ascending row order: strict > keeps the first maximum
    for (; i + 8 <= i1; i += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sp[(int64_t)(i + k) * (N + 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (v[k] > best) { best = v[k]; bi = i + k; }
    }
    for (; i < i1; ++i) {
        const float v = sp[(int64_t)i * (N + 1)];
        if (v > best) { best = v; bi = i; }
    }
    const unsigned long long key = ((unsigned long long)orderable(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bi);
    atomicMax(colbest + (int64_t)b * Nmax + j, key);
}

template <class RD>
__global__ __launch_bounds__(256) void mutual_kernel(int M, int N, float thr, const int* __restrict__ idx0,
                                                     const float* __restrict__ max0,
                                                     const unsigned long long* __restrict__ colbest,
                                                     int64_t* __restrict__ matches0, float* __restrict__ ms0,
                                                     int64_t* __restrict__ matches1, float* __restrict__ ms1, RD rd) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int Mmax = M, Nmax = N;                    // workspace strides
    int64_t o0 = (int64_t)b * M, o1 = (int64_t)b * N;  // output offsets (packed per pair when ragged)
    if (rd.B > 0) { M = rd.off0[b + 1] - rd.off0[b]; N = rd.off1[b + 1] - rd.off1[b]; o0 = rd.off0[b]; o1 = rd.off1[b]; }
    if (t < M) {
        const int j = idx0[(int64_t)b * Mmax + t];
        const int i1 = (int)(0xFFFFFFFFu - (unsigned)(colbest[(int64_t)b * Nmax + j] & 0xFFFFFFFFull));
        const bool mutual = i1 == t;
        const float s = mutual ? expf(max0[(int64_t)b * Mmax + t]) : 0.f;
        const bool valid = mutual && s > thr;
        matches0[o0 + t] = valid ? (int64_t)j : (int64_t)-1;
        ms0[o0 + t] = s;
    }
    if (matches1 && t < N) {
        const int i = (int)(0xFFFFFFFFu - (unsigned)(colbest[(int64_t)b * Nmax + t] & 0xFFFFFFFFull));
        const bool mutual = idx0[(int64_t)b * Mmax + i] == t;          // then mutual0[i] holds as well
        const float s = mutual ? expf(max0[(int64_t)b * Mmax + i]) : 0.f;
        const bool valid = mutual && s > thr;
        matches1[o1 + t] = valid ? (int64_t)i : (int64_t)-1;
        ms1[o1 + t] = s;
    }
}

// rows of the keypoint-encoder input: [2*x/(W-1)-1, 2*y/(H-1)-1, side_info..., 0 ... 0]  (32 floats/token)
// reference: superglue.py:74-78 (normalize_keypoints), positional_encoding.py:18 (cat + transpose).
// Ragged batches: every pair has its own image size (er.B > 0: pair b owns tokens er.off[b] .. er.off[b+1]).
__global__ __launch_bounds__(256) void encoder_input_kernel(const float* __restrict__ kpts, const float* __restrict__ side,
                                                            int64_t tokens, int s, float wm1, float hm1,
                                                            float* __restrict__ out, EncoderRagged er) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t = g >> 3;
    const int c4 = (int)(g & 7) * 4;
    if (t >= tokens) return;
    if (er.B > 0) {
        int lo = 0, hi = er.B - 1;               // last b with off[b] <= t
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (er.off[mid] <= (int)t) lo = mid; else hi = mid - 1;
        }
        wm1 = er.wm1[lo]; hm1 = er.hm1[lo];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c4 + e;
        float v = 0.f;
        if (c == 0) v = 2.f * kpts[2 * t] / wm1 - 1.f;
        else if (c == 1) v = 2.f * kpts[2 * t + 1] / hm1 - 1.f;
        else if (c < 2 + s) v = side[t * s + (c - 2)];
        o[e] = v;
    }
    *reinterpret_cast<f32x4*>(out + t * 32 + c4) = o;
}

}  // namespace

extern "C" size_t og_matches_workspace_bytes(int32_t batch, int32_t m, int32_t n) {
    if (batch <= 0 || m <= 0 || n <= 0) return 0;
    return sizeof(unsigned long long) * (size_t)batch * n + (sizeof(int) + sizeof(float)) * (size_t)batch * m + 16;
}

namespace {
template <class RD>
int matches_run(const float* scores, int B, int m, int n, float thr, int64_t* matches0, float* ms0,
                int64_t* matches1, float* ms1, void* workspace, hipStream_t st, const RD& rd, bool rows_done) {
    if (!scores || !matches0 || !ms0 || !workspace || B <= 0 || m <= 0 || n <= 0) return OG_E_INVALID;
    if ((matches1 == nullptr) != (ms1 == nullptr)) return OG_E_INVALID;
    if ((uintptr_t)workspace & 15) return OG_E_ALIGN;
    const MatchWs w = mw_layout(workspace, B, m, n);
    hipError_t e = hipMemsetAsync(w.colbest, 0, sizeof(unsigned long long) * (size_t)B * n, st);
    if (e != hipSuccess) return (int)e;
    if (!rows_done) hipLaunchKernelGGL(row_argmax_kernel<RD>, dim3((m + 3) / 4, B), dim3(256), 0, st, scores, m, n, w.idx0, w.max0, rd);
    hipLaunchKernelGGL(col_argmax_kernel<RD>, dim3((n + 255) / 256, (m + 63) / 64, B), dim3(256), 0, st, scores, m, n, w.colbest, rd);
    const int mx = m > n ? m : n;
    hipLaunchKernelGGL(mutual_kernel<RD>, dim3((mx + 255) / 256, B), dim3(256), 0, st, m, n, thr, w.idx0, w.max0, w.colbest,
                       matches0, ms0, matches1, ms1, rd);
    return og_launch_status();
}
}  // namespace

int og_launch_matches(const float* scores, int B, int m, int n, float thr, int64_t* matches0, float* ms0,
                      int64_t* matches1, float* ms1, void* workspace, hipStream_t st, const RaggedDesc* rag, bool rows_done) {
    if (rag) {
        if (rag->B != B) return OG_E_INVALID;
        return matches_run<RaggedDesc>(scores, B, m, n, thr, matches0, ms0, matches1, ms1, workspace, st, *rag, rows_done);
    }
    return matches_run<RaggedNone>(scores, B, m, n, thr, matches0, ms0, matches1, ms1, workspace, st, RaggedNone{}, rows_done);
}

RowBest og_matches_row_best(void* matches_workspace, int B, int m, int n) {
    const MatchWs w = mw_layout(matches_workspace, B, m, n);
    return RowBest{w.idx0, w.max0, m};
}

extern "C" int og_extract_matches(const float* scores, int32_t batch, int32_t m, int32_t n, float match_threshold,
                                  int64_t* matches0, float* matching_scores0, int64_t* matches1,
                                  float* matching_scores1, void* workspace_dev, void* stream) {
    og_clear_status();
    return og_launch_matches(scores, batch, m, n, match_threshold, matches0, matching_scores0, matches1,
                             matching_scores1, workspace_dev, (hipStream_t)stream, nullptr, false);
}

int og_launch_encoder_input(const float* kpts, const float* side, int64_t tokens, int s, float wx, float wy,
                            float* out, hipStream_t st, const EncoderRagged* er) {
    if (!kpts || !out || tokens <= 0 || s < 0 || 2 + s > 32 || (s > 0 && !side)) return OG_E_INVALID;
    const int64_t threads = tokens * 8;
    EncoderRagged e;
    e.B = 0;
    if (er) e = *er;
    hipLaunchKernelGGL(encoder_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, kpts, side, tokens,
                       s, wx - 1.f, wy - 1.f, out, e);
    return og_launch_status();
}

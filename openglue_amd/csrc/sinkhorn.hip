// Log-domain Sinkhorn optimal transport with implicit dustbins, gfx950.
//
// Replaces SuperGlue.get_matching_probs + log_otp_solver (reference superglue.py:88-111,
// optimal_transport.py:20-28):
//     S~ = [[S, z], [z, z]] / reg ;  u = v = 0
//     repeat iters:  u_i = log a_i - LSE_j(S~_ij + v_j) ;  v_j = log b_j - LSE_i(S~_ij + u_i)
//     scores = S~ + u_i + v_j - norm,  norm = -log(m+n), log a = [norm]*m + [log n + norm], log b likewise.
//
// The reference materialises S~ and, per half-iteration, S~ + v plus the amax/sub/exp/sum chain of
// logsumexp (4-5 sweeps of a (m+1)(n+1) matrix).  Here
//   * S~ is never built: the dustbin row/column are the scalar z/reg and enter every LSE in closed form;
//   * ONE sweep of S per iteration: a workgroup keeps R rows x all columns of S in registers, computes the
//     row log-sum-exps (-> new u for its rows) and, from the same registers, the partial column
//     (max, sum-exp) over its R rows with the NEW u.  A small second kernel combines the per-row-block
//     partials into v (and handles the dustbin row / column, which only need u and v).
// HBM/L2 traffic per iteration: 4mn bytes read + 8n(m/R) bytes of partials, instead of >= 8mn.
// Numerics: exact max-subtracted two-pass LSE like torch.logsumexp, in fp32.
#include <stdlib.h>

#include "og_common.h"

namespace {

struct SinkhornWs {
    float* u;       // [B][ldu]
    float* v[2];    // [B][ldv] ping-pong
    float* pm;      // [B][RB][ldp] partial column max
    float* ps;      // [B][RB][ldp] partial column sum-exp
    int ldu, ldv, ldp, RB, R, CPT;
};

__host__ __device__ inline int sk_cpt(int n) { return n <= 1024 ? 1 : (n <= 2048 ? 2 : 4); }
// rows-per-wave sweep (n <= 2048): float4 chunks per lane and rows per block
inline int sk_rw() {                                                // rows per wave (block = 4 waves)
    static const int rw = [] { const char* e = getenv("OG_SINKHORN_RW"); const int v = e ? atoi(e) : 8; return (v == 4 || v == 16) ? v : 8; }();
    return rw;
}
inline int sk_cpl(int n) { return n <= 256 ? 1 : (n <= 512 ? 2 : (n <= 1024 ? 4 : 8)); }
inline bool sk_rows_variant(int n) { return n <= 2048; }

static SinkhornWs sk_layout(void* ws, int B, int m, int n) {
    SinkhornWs w{};
    w.CPT = sk_cpt(n);
    w.R = sk_rows_variant(n) ? 4 * sk_rw() : 16 / w.CPT;
    w.RB = (m + w.R - 1) / w.R;
    w.ldu = (int)og_round_up(m + 1, 4);
    w.ldv = (int)og_round_up(n + 1, 4);
    w.ldp = (int)og_round_up(n, 4);
    float* p = (float*)ws;
    w.u = p; p += (int64_t)B * w.ldu;
    w.v[0] = p; p += (int64_t)B * w.ldv;
    w.v[1] = p; p += (int64_t)B * w.ldv;
    p = (float*)og_round_up((int64_t)(uintptr_t)p, 16);
    w.pm = p; p += (int64_t)B * w.RB * w.ldp;
    w.ps = p;
    return w;
}

// block-wide reductions over 256 threads (4 waves); `sm` holds >= 4 floats; trailing barrier makes
// `sm` immediately reusable.
__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return v;
}

// One sweep: rows [rb*R, rb*R+R) of pair b.
template <int CPT, int R>
__global__ __launch_bounds__(256) void sinkhorn_sweep_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                             const float* __restrict__ zdev, float zhost,
                                                             float inv_reg, float la,
                                                             const float* __restrict__ v_in, int ldv,
                                                             float* __restrict__ u, int ldu,
                                                             float* __restrict__ pm, float* __restrict__ ps,
                                                             int ldp, int RB) {
    __shared__ float redm[4][R];
    __shared__ float reds[4][R];
    const int b = blockIdx.y, rb = blockIdx.x, row0 = rb * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;

    float vv[CPT][4];
    const float* vb = v_in + (int64_t)b * ldv;
    const float vN = vb[N];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c0 = 4 * tid + 1024 * k;
        if (c0 < N) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(vb + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[k][e] = t[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[k][e] = 0.f;
        }
    }

    float x[R][CPT][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        const float* sp = S + ((int64_t)b * M + row) * lds;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c0 = 4 * tid + 1024 * k;
            if (row < M && c0 < N) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(sp + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[r][k][e] = (c0 + e < N) ? t[e] * inv_reg : OG_NEG_INF;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[r][k][e] = OG_NEG_INF;
            }
        }
    }

    // ---- phase A: row log-sum-exp over the n columns + dustbin column ----
    const float dcol = zr + vN;
    float rmax[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float m = OG_NEG_INF;
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, x[r][k][e] + vv[k][e]);
        m = wave_max(m);
        if (lane == 0) redm[wave][r] = m;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r)
        rmax[r] = fmaxf(fmaxf(fmaxf(redm[0][r], redm[1][r]), fmaxf(redm[2][r], redm[3][r])), dcol);
    float ur[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __expf(x[r][k][e] + vv[k][e] - rmax[r]);
        s = wave_sum(s);
        if (lane == 0) reds[wave][r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float s = (reds[0][r] + reds[1][r]) + (reds[2][r] + reds[3][r]) + __expf(dcol - rmax[r]);
        ur[r] = la - (rmax[r] + __logf(s));
        if (tid == r && row0 + r < M) u[(int64_t)b * ldu + row0 + r] = ur[r];
    }

    // ---- phase B: partial column (max, sum-exp) over this block's rows, with the NEW u ----
    float* pmb = pm + ((int64_t)b * RB + rb) * ldp;
    float* psb = ps + ((int64_t)b * RB + rb) * ldp;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c0 = 4 * tid + 1024 * k;
        if (c0 >= N) continue;
        f32x4 om, os;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = OG_NEG_INF;
#pragma unroll
            for (int r = 0; r < R; ++r) m = fmaxf(m, x[r][k][e] + ur[r]);
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) s += __expf(x[r][k][e] + ur[r] - m);
            om[e] = m; os[e] = s;       // columns >= N inside the float4: m = -inf, s = NaN -- never read
        }
        *reinterpret_cast<f32x4*>(pmb + c0) = om;
        *reinterpret_cast<f32x4*>(psb + c0) = os;
    }
}

// ---------------------------------------------------------------------------------------------------
// Sweep, n <= 2048: "a wave owns rows".  Block = 4 waves x RW rows; a wave streams its rows in groups of
// RG, lane l holds columns 4l + 256k + e (k < CPL).  Row log-sum-exps are wave-local (no LDS, no barrier);
// the column (max, sum-exp) partials are carried ONLINE in registers across the wave's row groups and the
// four waves are merged once per block through LDS.  Row blocks of 4*RW rows -> RB partials per column.
template <int CPL, int RG, int RW>
__global__ __launch_bounds__(256) void sinkhorn_sweep_rows_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                                  const float* __restrict__ zdev, float zhost,
                                                                  float inv_reg, float la,
                                                                  const float* __restrict__ v_in, int ldv,
                                                                  float* __restrict__ u, int ldu,
                                                                  float* __restrict__ pm, float* __restrict__ ps,
                                                                  int ldp, int RB, int abl) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [4 waves][2][256*CPL*4] (max | sum)
    constexpr int NC = 256 * CPL;                                   // columns covered by a wave
    const int b = blockIdx.y, rb = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    const float* vb = v_in + (int64_t)b * ldv;
    const float dcol = zr + vb[N];

    float vv[CPL][4], cm[CPL][4], cs[CPL][4];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = 4 * lane + 256 * k;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c0 < N) t = *reinterpret_cast<const f32x4*>(vb + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vv[k][e] = t[e]; cm[k][e] = OG_NEG_INF; cs[k][e] = 0.f; }
    }

    const int wrow0 = rb * (4 * RW) + wave * RW;
#pragma unroll 1
    for (int g0 = 0; g0 < RW; g0 += RG) {
        const int row0 = wrow0 + g0;
        if (row0 >= M) break;                                       // wave-uniform
        float x[RG][CPL][4];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int row = row0 + r;
            const float* sp = S + ((int64_t)b * M + row) * lds;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c0 = 4 * lane + 256 * k;
                if (row < M && c0 < N) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(sp + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[r][k][e] = (c0 + e < N) ? t[e] * inv_reg : OG_NEG_INF;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[r][k][e] = OG_NEG_INF;
                }
            }
        }
        // row log-sum-exp (wave-local) -> u for the RG rows
        float ur[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float mx = OG_NEG_INF;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, x[r][k][e] + vv[k][e]);
            mx = fmaxf(wave_max(mx), dcol);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += __expf(x[r][k][e] + vv[k][e] - mx);
            sum = wave_sum(sum) + __expf(dcol - mx);
            ur[r] = la - (mx + __logf(sum));
            if (abl & 1) ur[r] = la - mx;
            if (lane == 0 && row0 + r < M) u[(int64_t)b * ldu + row0 + r] = ur[r];
        }
        // online column partials with the new u
        if (abl & 2) {      // profiling: skip the column half (keep x alive)
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) cm[k][e] = fmaxf(cm[k][e], x[0][k][e] + ur[0]);
            continue;
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gm = cm[k][e];
#pragma unroll
                for (int r = 0; r < RG; ++r) gm = fmaxf(gm, x[r][k][e] + ur[r]);
                float acc = cs[k][e] * __expf(cm[k][e] - gm);        // first group: 0 * exp(-inf - gm) = 0
#pragma unroll
                for (int r = 0; r < RG; ++r) acc += __expf(x[r][k][e] + ur[r] - gm);
                // columns >= N: gm = -inf -> NaN, never stored (guarded below)
                cm[k][e] = gm; cs[k][e] = acc;
            }
    }

    // merge the four waves (a wave with no valid row contributes (-inf, 0))
    float* smm = sm + (size_t)wave * 2 * NC;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = 4 * lane + 256 * k;
        *reinterpret_cast<f32x4*>(smm + c0) = f32x4{cm[k][0], cm[k][1], cm[k][2], cm[k][3]};
        *reinterpret_cast<f32x4*>(smm + NC + c0) = f32x4{cs[k][0], cs[k][1], cs[k][2], cs[k][3]};
    }
    __syncthreads();
    float* pmb = pm + ((int64_t)b * RB + rb) * ldp;
    float* psb = ps + ((int64_t)b * RB + rb) * ldp;
    for (int c0 = 4 * tid; c0 < N; c0 += 1024) {
        f32x4 wm[4], wsum[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wm[w] = *reinterpret_cast<const f32x4*>(sm + (size_t)w * 2 * NC + c0);
            wsum[w] = *reinterpret_cast<const f32x4*>(sm + (size_t)w * 2 * NC + NC + c0);
        }
        f32x4 om, os;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mx = fmaxf(fmaxf(wm[0][e], wm[1][e]), fmaxf(wm[2][e], wm[3][e]));
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc += (wm[w][e] == OG_NEG_INF) ? 0.f : wsum[w][e] * __expf(wm[w][e] - mx);
            om[e] = mx; os[e] = acc;
        }
        *reinterpret_cast<f32x4*>(pmb + c0) = om;
        *reinterpret_cast<f32x4*>(psb + c0) = os;
    }
}

// Combine: grid (ceil((N+1)/256), B).  Finishes iteration t: dustbin-row u, all v.
__global__ __launch_bounds__(256) void sinkhorn_combine_kernel(int M, int N, const float* __restrict__ zdev,
                                                               float zhost, float inv_reg, float la_bin, float lb,
                                                               float lb_bin, const float* __restrict__ v_in,
                                                               float* __restrict__ v_out, int ldv,
                                                               float* __restrict__ u, int ldu,
                                                               const float* __restrict__ pm,
                                                               const float* __restrict__ ps, int ldp, int RB) {
    __shared__ float sm[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    const float* vb = v_in + (int64_t)b * ldv;
    // dustbin row: u_M = log a_M - (z + LSE_{j<=N} v_j)
    float m = OG_NEG_INF;
    for (int j = tid; j <= N; j += 256) m = fmaxf(m, vb[j]);
    m = block_max(m, sm);
    float s = 0.f;
    for (int j = tid; j <= N; j += 256) s += __expf(vb[j] - m);
    s = block_sum(s, sm);
    const float uM = la_bin - (zr + m + __logf(s));
    if (blockIdx.x == 0 && tid == 0) u[(int64_t)b * ldu + M] = uM;

    const int j = blockIdx.x * 256 + tid;
    if (j < N) {
        const float* pmb = pm + (int64_t)b * RB * ldp + j;
        const float* psb = ps + (int64_t)b * RB * ldp + j;
        float cm = zr + uM;                      // dustbin row entry of column j
        // chunks of 8 row blocks: 16 independent loads in flight per thread (the naive dependent
        // loop is latency-bound), online (max, sum) merge per chunk
        float cs = 1.f;                          // exp(zr + uM - cm)
        for (int rb0 = 0; rb0 < RB; rb0 += 8) {
            float pmv[8], psv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool ok = rb0 + q < RB;
                pmv[q] = ok ? pmb[(int64_t)(rb0 + q) * ldp] : OG_NEG_INF;
                psv[q] = ok ? psb[(int64_t)(rb0 + q) * ldp] : 0.f;
            }
            float mx = cm;
#pragma unroll
            for (int q = 0; q < 8; ++q) mx = fmaxf(mx, pmv[q]);
            cs *= __expf(cm - mx);
#pragma unroll
            for (int q = 0; q < 8; ++q) cs += psv[q] * __expf(pmv[q] - mx);
            cm = mx;
        }
        v_out[(int64_t)b * ldv + j] = lb - (cm + __logf(cs));
    }
    if (blockIdx.x == gridDim.x - 1) {
        // dustbin column: v_N = log b_N - (z + LSE_{i<=M} u_i), with the new u (u_M from above)
        const float* ub = u + (int64_t)b * ldu;
        float um = uM;
        for (int i = tid; i < M; i += 256) um = fmaxf(um, ub[i]);
        um = block_max(um, sm);
        float us = 0.f;
        for (int i = tid; i < M; i += 256) us += __expf(ub[i] - um);
        us = block_sum(us, sm) + __expf(uM - um);
        if (tid == 0) v_out[(int64_t)b * ldv + N] = lb_bin - (zr + um + __logf(us));
    }
}

// scores[b][i][j] = ((S~_ij + u_i) + v_j) - norm   (same association as optimal_transport.py:28, superglue.py:111)
__global__ __launch_bounds__(256) void sinkhorn_scores_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                              const float* __restrict__ zdev, float zhost,
                                                              float inv_reg, float norm,
                                                              const float* __restrict__ u, int ldu,
                                                              const float* __restrict__ v, int ldv,
                                                              float* __restrict__ scores) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row > M) return;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    const float ui = u[(int64_t)b * ldu + row];
    const float* vb = v + (int64_t)b * ldv;
    float* out = scores + ((int64_t)b * (M + 1) + row) * (N + 1);
    if (row < M) {
        const float* sp = S + ((int64_t)b * M + row) * lds;
        for (int j = lane; j < N; j += 64) out[j] = ((sp[j] * inv_reg + ui) + vb[j]) - norm;
    } else {
        for (int j = lane; j < N; j += 64) out[j] = ((zr + ui) + vb[j]) - norm;
    }
    if (lane == 0) out[N] = ((zr + ui) + vb[N]) - norm;
}

template <int CPT>
void launch_sweep(const float* S, int64_t lds, int B, int m, int n, const float* zdev, float zhost, float inv_reg, float la,
                  const float* v_in, const SinkhornWs& w, hipStream_t st) {
    constexpr int R = 16 / CPT;
    hipLaunchKernelGGL((sinkhorn_sweep_kernel<CPT, R>), dim3(w.RB, B), dim3(256), 0, st, S, lds, m, n, zdev, zhost, inv_reg, la,
                       v_in, w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB);
}

}  // namespace

extern "C" size_t og_sinkhorn_workspace_bytes(int32_t batch, int32_t m, int32_t n) {
    if (batch <= 0 || m <= 0 || n <= 0 || n > 4096) return 0;
    const int cpt = sk_cpt(n), R = sk_rows_variant(n) ? 4 * sk_rw() : 16 / cpt, RB = (m + R - 1) / R;
    const int64_t ldu = og_round_up(m + 1, 4), ldv = og_round_up(n + 1, 4), ldp = og_round_up(n, 4);
    const int64_t floats = (int64_t)batch * (ldu + 2 * ldv) + 4 + 2 * (int64_t)batch * RB * ldp;
    return (size_t)floats * sizeof(float);
}

int og_launch_sinkhorn(const float* S, int64_t lds, const float* zdev, float dustbin, int B, int m, int n, int iters, float reg,
                       float* scores, void* workspace, hipStream_t st) {
    if (!S || !scores || !workspace || B <= 0 || m <= 0 || n <= 0 || iters < 0 || !(reg > 0.f)) return OG_E_INVALID;
    if (n > 4096) return OG_E_SHAPE;
    if ((lds & 3) || ((uintptr_t)S & 15) || ((uintptr_t)workspace & 15)) return OG_E_ALIGN;
    const SinkhornWs w = sk_layout(workspace, B, m, n);
    const float inv_reg = 1.f / reg;
    const double norm = -log((double)m + (double)n);
    const float la = (float)norm, lb = (float)norm;
    const float la_bin = (float)norm + (float)log((double)n);   // log_a[-1] += log(n) in fp32 (superglue.py:100)
    const float lb_bin = (float)norm + (float)log((double)m);
    // u = v = 0 (optimal_transport.py:22)
    hipError_t e = hipMemsetAsync(w.u, 0, sizeof(float) * (size_t)B * (w.ldu + 2 * (size_t)w.ldv), st);
    if (e != hipSuccess) return (int)e;
    static const int sk_abl = [] { const char* e = getenv("OG_SINKHORN_ABLATE"); return e ? atoi(e) : 0; }();
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (sk_rows_variant(n)) {
            const dim3 grid(w.RB, B), block(256);
#define OG_SWEEP_ROWS_RW(CPL, RG, RW)                                                                                   \
    hipLaunchKernelGGL((sinkhorn_sweep_rows_kernel<CPL, RG, RW>), grid, block, sizeof(float) * 4 * 2 * 256 * CPL, st, S,        \
                       lds, m, n, zdev, dustbin, inv_reg, la, w.v[cur], w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB, sk_abl)
#define OG_SWEEP_ROWS(CPL, RG)                                                                                          \
    do {                                                                                                                \
        if (sk_rw() == 4) OG_SWEEP_ROWS_RW(CPL, RG, 4);                                                                 \
        else if (sk_rw() == 16) OG_SWEEP_ROWS_RW(CPL, RG, 16);                                                          \
        else OG_SWEEP_ROWS_RW(CPL, RG, 8);                                                                              \
    } while (0)
            static const int sk_rg = [] { const char* e = getenv("OG_SINKHORN_RG"); return e ? atoi(e) : 2; }();
            switch (sk_cpl(n)) {
                case 1: OG_SWEEP_ROWS(1, 4); break;
                case 2: OG_SWEEP_ROWS(2, 4); break;
                case 4:
                    if (sk_rg == 1) OG_SWEEP_ROWS(4, 1);
                    else if (sk_rg == 2) OG_SWEEP_ROWS(4, 2);
                    else OG_SWEEP_ROWS(4, 4);
                    break;
                default: OG_SWEEP_ROWS(8, 2); break;
            }
#undef OG_SWEEP_ROWS
#undef OG_SWEEP_ROWS_RW
        } else
        switch (w.CPT) {
            case 1: launch_sweep<1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, w.v[cur], w, st); break;
            case 2: launch_sweep<2>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, w.v[cur], w, st); break;
            default: launch_sweep<4>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, w.v[cur], w, st); break;
        }
        hipLaunchKernelGGL(sinkhorn_combine_kernel, dim3((n + 1 + 255) / 256, B), dim3(256), 0, st, m, n, zdev, dustbin, inv_reg, la_bin,
                           lb, lb_bin, w.v[cur], w.v[cur ^ 1], w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB);
        cur ^= 1;
    }
    hipLaunchKernelGGL(sinkhorn_scores_kernel, dim3((m + 1 + 3) / 4, B), dim3(256), 0, st, S, lds, m, n, zdev, dustbin, inv_reg,
                       (float)norm, w.u, w.ldu, w.v[cur], w.ldv, scores);
    return og_launch_status();
}

extern "C" int og_sinkhorn(const float* S, int64_t lds, float dustbin, int32_t batch, int32_t m, int32_t n,
                           int32_t iters, float reg, float* scores, void* workspace_dev, void* stream) {
    og_clear_status();
    return og_launch_sinkhorn(S, lds, nullptr, dustbin, batch, m, n, iters, reg, scores, workspace_dev, (hipStream_t)stream);
}

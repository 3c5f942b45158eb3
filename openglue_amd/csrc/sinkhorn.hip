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
//   * ONE sweep of S per iteration (sinkhorn_sweep_kernel): rows are streamed through registers in small
//     groups; a group's row log-sum-exps give the new u for those rows at once, and the same registers then
//     update ONLINE (max, sum-exp) column partials with the new u.  A small second kernel merges the
//     per-row-block partials into v and handles the dustbin row / column, which only need u and v.
// Traffic per iteration: 4mn bytes of S + 16n(m/32) bytes of partials, instead of >= 8mn.
// The kernel is bandwidth/latency-bound (removing half of its arithmetic changes its time by 7 %), so the
// geometry is chosen for bytes in flight: 32 rows per workgroup, 2 rows per group, <= 1024 columns per wave.
// Numerics: max-subtracted LSEs like torch.logsumexp, fp32, v_exp_f32 / v_log_f32.
#include <stdlib.h>
#include <type_traits>

#include "og_common.h"

namespace {

constexpr int SK_ROWS = 32;     // rows per workgroup -> ceil(m/32) partials per column

struct SinkhornGeom { int CPL, WPR; };   // float4 chunks per lane, waves per row
inline SinkhornGeom sk_geom(int n) {
    if (n <= 256) return {1, 1};
    if (n <= 512) return {2, 1};
    if (n <= 1024) return {4, 1};
    if (n <= 2048) return {4, 2};
    if (n <= 4096) return {4, 4};
    return {8, 4};               // <= 8192
}

struct SinkhornWs {
    float* u;       // [B][ldu]
    float* v[2];    // [B][ldv] ping-pong
    unsigned* flags; // [4] zeroed with u, v on every call: [0] != 0 = a non-finite value reached the scores (og_sinkhorn_status 3); [1] = the status
                     // word of the resident schedule (0 / 1 timed out / 2 recomputed by the safety net): no memset of its own
    float* pm;      // [B][RB][ldp] partial column max
    float* ps;      // [B][RB][ldp] partial column sum-exp
    int ldu, ldv, ldp, RB;
};

static SinkhornWs sk_layout(void* ws, int B, int m, int n) {
    SinkhornWs w{};
    w.RB = (m + SK_ROWS - 1) / SK_ROWS;
    w.ldu = (int)og_round_up(m + 1, 4);
    w.ldv = (int)og_round_up(n + 1, 4);
    w.ldp = (int)og_round_up(n, 4);
    float* p = (float*)ws;
    w.u = p; p += (int64_t)B * w.ldu;
    w.v[0] = p; p += (int64_t)B * w.ldv;
    w.v[1] = p; p += (int64_t)B * w.ldv;
    w.flags = (unsigned*)p; p += 4;
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

// exp(a - b) with the convention exp(-inf - (-inf)) = 0 (an empty set contributes nothing)
__device__ __forceinline__ float exp_rel(float a, float b) { return a == OG_NEG_INF ? 0.f : __expf(a - b); }

// Sweep of rows [32 rb, 32 rb + 32) of pair b.  Block = 4 waves = (4/WPR) row streams x WPR column parts;
// a wave streams its rows in groups of RG, lane l holds columns cpart*256*CPL + 4l + 256k + e (k < CPL).
template <int CPL, int RG, int WPR, class RD>
__global__ __launch_bounds__(256) void sinkhorn_sweep_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                             const float* __restrict__ zdev, float zhost, float inv_reg,
                                                             float la, const float* __restrict__ v_in, int ldv,
                                                             float* __restrict__ u, int ldu, float* __restrict__ pm,
                                                             float* __restrict__ ps, int ldp, int RB, int64_t strideS,
                                                             RD rd) {
    constexpr int NRS = 4 / WPR;                 // row streams
    constexpr int RW = SK_ROWS / NRS;            // rows per stream
    constexpr int NCW = 256 * CPL;               // columns per wave
    constexpr int NCB = NCW * WPR;               // columns per block
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xch = sm + NRS * 2 * NCB;             // [2 parity][NRS][RG][WPR][2]: row (max, sum) exchange between column parts

    const int b = blockIdx.y, rb = blockIdx.x;
    if (rd.B > 0) {                              // ragged: this pair's own size; la = norm = -log(m_b + n_b)
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        if (rb * SK_ROWS >= M) return;           // whole block: no barrier is skipped by a subset of waves
        la = -__logf((float)(M + N));
    }
    S += (int64_t)b * strideS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cp = wave % WPR, rs = wave / WPR;
    const int cbase = cp * NCW + 4 * lane;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    const float* vb = v_in + (int64_t)b * ldv;
    const float dcol = zr + vb[N];               // dustbin column entry of every row

    float vv[CPL][4], cm[CPL][4], cs[CPL][4];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = cbase + 256 * k;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c0 < N) t = *reinterpret_cast<const f32x4*>(vb + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vv[k][e] = t[e]; cm[k][e] = OG_NEG_INF; cs[k][e] = 0.f; }
    }

    const int srow0 = rb * SK_ROWS + rs * RW;
    int parity = 0;
    // raw rows of the NEXT group are fetched before the current group is processed (register double buffer):
    // the sweep is bound by bytes in flight, not by arithmetic
    f32x4 nx[RG][CPL];
    auto fetch = [&](int row0) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int row = row0 + r;
            const float* sp = S + (int64_t)row * lds;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c0 = cbase + 256 * k;
                if (row < M && c0 < N) nx[r][k] = *reinterpret_cast<const f32x4*>(sp + c0);
                else nx[r][k] = f32x4{OG_NEG_INF, OG_NEG_INF, OG_NEG_INF, OG_NEG_INF};
            }
        }
    };
    fetch(srow0);
#pragma unroll 1
    for (int g0 = 0; g0 < RW; g0 += RG) {        // same trip count in every wave (barrier inside when WPR > 1)
        const int row0 = srow0 + g0;
        float x[RG][CPL][4];
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c0 = cbase + 256 * k;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[r][k][e] = (c0 + e < N) ? nx[r][k][e] * inv_reg : OG_NEG_INF;
            }
        if (g0 + RG < RW) fetch(row0 + RG);
        // ---- row log-sum-exp -> u of the RG rows ----
        float mx[RG], sum[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float m_ = OG_NEG_INF;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) m_ = fmaxf(m_, x[r][k][e] + vv[k][e]);
            m_ = wave_max(m_);
            const float ms = m_ == OG_NEG_INF ? 0.f : m_;      // empty part: every term below is exp(-inf) = 0
            float s_ = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s_ += __expf(x[r][k][e] + vv[k][e] - ms);
            mx[r] = m_; sum[r] = wave_sum(s_);
        }
        float ur[RG];
        if (WPR > 1) {
            float* xs = xch + ((parity * NRS + rs) * RG) * WPR * 2;
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < RG; ++r) { xs[(r * WPR + cp) * 2] = mx[r]; xs[(r * WPR + cp) * 2 + 1] = sum[r]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                float mt = dcol;
#pragma unroll
                for (int w = 0; w < WPR; ++w) mt = fmaxf(mt, xs[(r * WPR + w) * 2]);
                float st = __expf(dcol - mt);
#pragma unroll
                for (int w = 0; w < WPR; ++w) st += xs[(r * WPR + w) * 2 + 1] * exp_rel(xs[(r * WPR + w) * 2], mt);
                ur[r] = la - (mt + __logf(st));
            }
            parity ^= 1;
        } else {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const float mt = fmaxf(mx[r], dcol);
                const float st = sum[r] * exp_rel(mx[r], mt) + __expf(dcol - mt);
                ur[r] = la - (mt + __logf(st));
            }
        }
        if (cp == 0 && lane == 0) {
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (row0 + r < M) u[(int64_t)b * ldu + row0 + r] = ur[r];
        }
        // ---- online column partials with the new u (rows >= M hold -inf and contribute exp(-inf) = 0) ----
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gm = cm[k][e];
#pragma unroll
                for (int r = 0; r < RG; ++r) gm = fmaxf(gm, x[r][k][e] + ur[r]);
                const float gs = gm == OG_NEG_INF ? 0.f : gm;
                float acc = cs[k][e] * exp_rel(cm[k][e], gs);
#pragma unroll
                for (int r = 0; r < RG; ++r) acc += __expf(x[r][k][e] + ur[r] - gs);
                cm[k][e] = gm; cs[k][e] = acc;
            }
    }

    // ---- merge the row streams of each column part and write the block's partials ----
    float* pmb = pm + ((int64_t)b * RB + rb) * ldp;
    float* psb = ps + ((int64_t)b * RB + rb) * ldp;
    if (NRS == 1) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c0 = cbase + 256 * k;
            if (c0 < N) {
                *reinterpret_cast<f32x4*>(pmb + c0) = f32x4{cm[k][0], cm[k][1], cm[k][2], cm[k][3]};
                *reinterpret_cast<f32x4*>(psb + c0) = f32x4{cs[k][0], cs[k][1], cs[k][2], cs[k][3]};
            }
        }
        return;
    }
    float* smm = sm + (size_t)rs * 2 * NCB;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = cbase + 256 * k;
        *reinterpret_cast<f32x4*>(smm + c0) = f32x4{cm[k][0], cm[k][1], cm[k][2], cm[k][3]};
        *reinterpret_cast<f32x4*>(smm + NCB + c0) = f32x4{cs[k][0], cs[k][1], cs[k][2], cs[k][3]};
    }
    __syncthreads();
    for (int c0 = 4 * tid; c0 < N; c0 += 1024) {
        f32x4 wm[NRS], wsum[NRS];
#pragma unroll
        for (int w = 0; w < NRS; ++w) {
            wm[w] = *reinterpret_cast<const f32x4*>(sm + (size_t)w * 2 * NCB + c0);
            wsum[w] = *reinterpret_cast<const f32x4*>(sm + (size_t)w * 2 * NCB + NCB + c0);
        }
        f32x4 om, os;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mxe = wm[0][e];
#pragma unroll
            for (int w = 1; w < NRS; ++w) mxe = fmaxf(mxe, wm[w][e]);
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NRS; ++w) acc += wsum[w][e] * exp_rel(wm[w][e], mxe);
            om[e] = mxe; os[e] = acc;        // columns >= N inside the float4: never read
        }
        *reinterpret_cast<f32x4*>(pmb + c0) = om;
        *reinterpret_cast<f32x4*>(psb + c0) = os;
    }
}

// Combine: grid (ceil((N+1)/256), B).  Finishes iteration t: dustbin-row u, all v.
template <class RD>
__global__ __launch_bounds__(256) void sinkhorn_combine_kernel(int M, int N, const float* __restrict__ zdev,
                                                               float zhost, float inv_reg, float la_bin, float lb,
                                                               float lb_bin, const float* __restrict__ v_in,
                                                               float* __restrict__ v_out, int ldv,
                                                               float* __restrict__ u, int ldu,
                                                               const float* __restrict__ pm,
                                                               const float* __restrict__ ps, int ldp, int RB, RD rd) {
    __shared__ float sm[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    int RBv = RB;                                // row blocks that hold valid partials for this pair
    if (rd.B > 0) {
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        if ((int)blockIdx.x * 256 > N) return;
        RBv = (M + SK_ROWS - 1) / SK_ROWS;
        const float norm = -__logf((float)(M + N));
        lb = norm; la_bin = norm + __logf((float)N); lb_bin = norm + __logf((float)M);
    }
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
        for (int rb0 = 0; rb0 < RBv; rb0 += 8) {
            float pmv[8], psv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool ok = rb0 + q < RBv;
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
    if ((int)blockIdx.x == N / 256) {          // the block that owns column N
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

// ---------------------------------------------------------------------------------------------------
// Dual-stabilised iterations (every iteration after the first).
// After one max-subtracted iteration the plan entries  P_ij = exp(S~_ij + u_i + v_j)  are <= max(a_i, b_j) < 1 and stay so
// under every later half-update (a row update makes the row sums a_i, a column update the column sums b_j), so the
// log-sum-exps can use the CURRENT duals as their reference instead of a running maximum:
//     rowsum_i = sum_j P_ij            u_i += log a_i - log rowsum_i
//     colsum_j = sum_i P_ij f_i        v_j += log b_j - log colsum_j,     f_i = a_i / rowsum_i  (the row's own correction)
// No overflow is possible (all terms <= 1), underflow only zeroes entries below 2^-126, and a row (column) sum cannot
// vanish: it was a_i (b_j) one half-step earlier and the other side moved it by at most a factor (m+n).
// Same recursion as optimal_transport.py:24-26 in exact arithmetic; in fp32 the two forms differ by rounding (~1e-6).
// Work per element: one fma, one add, one v_exp_f32, one add for the row pass and ONE fma for the column pass (the
// exponentials of the row pass are reused), against ~18 VALU + 2.5 exp of the max-subtracted sweep, which is VALU-bound
// at 1024 keypoints (27 us of VALU for 22 us of HBM time).  Everything is kept in base 2 inside the kernels
// (v_exp_f32 / v_log_f32); u and v stay in natural units in memory.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int CPL, int RG, int WPR, class RD, int ROWS = SK_ROWS, bool NT = false>
__global__ __launch_bounds__(256) void sinkhorn_sweep_fast_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                                  const float* __restrict__ zdev, float zhost, float inv_reg,
                                                                  float la, const float* __restrict__ v_in, int ldv,
                                                                  float* __restrict__ u, int ldu, float* __restrict__ ps,
                                                                  int ldp, int RB, int64_t strideS, float in_scale,
                                                                  float out_scale, RD rd) {
    // ROWS = rows per workgroup (a multiple of 32: fewer, larger workgroups mean fewer column partials to write and to merge, as long
    // as the grid still fills the chip; sinkhorn_run); NT = stream the score rows with non-temporal loads (matrices that do not fit
    // the 256 MB Infinity Cache: 4.6 -> 5.0-5.2 TB/s at 537 MB; with 64 / 128 rows per workgroup 5.6-5.8)
    constexpr int rows = ROWS;
    // u, v in memory: base-2 units between two dual-stabilised iterations (no per-iteration unit conversion: at
    // convergence the increments vanish and the duals stop moving, instead of random-walking by an ulp per round trip);
    // in_scale = log2(e) when the previous iteration left natural units, out_scale = ln 2 on the last iteration.
    constexpr int NRS = 4 / WPR;                 // row streams
    constexpr int RW = ROWS / NRS;               // rows per stream
    constexpr int NCW = 256 * CPL;               // columns per wave
    constexpr int NCB = NCW * WPR;               // columns per block
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xch = sm + NRS * NCB;                 // [2 parity][NRS][RG][WPR]: row sums exchanged between column parts

    const int b = blockIdx.y, rb = blockIdx.x;
    if (rd.B > 0) {
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        if (rb * rows >= M) return;
        la = -__logf((float)(M + N));
    }
    S += (int64_t)b * strideS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cp = wave % WPR, rs = wave / WPR;
    const int cbase = cp * NCW + 4 * lane;
    const float c2 = inv_reg * LOG2E;
    const float zr2 = (zdev ? zdev[0] : zhost) * c2;
    const float* vb = v_in + (int64_t)b * ldv;
    float* ub = u + (int64_t)b * ldu;
    const float dcol2 = zr2 + vb[N] * in_scale;  // dustbin column entry of every row (without u)
    const float la2 = la * LOG2E;

    float vv[CPL][4], cs[CPL][4];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = cbase + 256 * k;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c0 < N) t = *reinterpret_cast<const f32x4*>(vb + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vv[k][e] = t[e] * in_scale; cs[k][e] = 0.f; }
    }

    const int srow0 = rb * rows + rs * RW;
    int parity = 0;
    f32x4 nx[RG][CPL];
    float nu[RG];                                // u of the next group's rows
    auto fetch = [&](int row0) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int row = row0 + r;
            const float* sp = S + (int64_t)row * lds;
            nu[r] = row < M ? ub[row] : 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c0 = cbase + 256 * k;
                if (row < M && c0 < N) nx[r][k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + c0))
                                                     : *reinterpret_cast<const f32x4*>(sp + c0);
                else nx[r][k] = f32x4{OG_NEG_INF, OG_NEG_INF, OG_NEG_INF, OG_NEG_INF};
            }
        }
    };
    fetch(srow0);
#pragma unroll 1
    for (int g0 = 0; g0 < RW; g0 += RG) {
        const int row0 = srow0 + g0;
        float p[RG][CPL][4], uo[RG], sum[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            uo[r] = nu[r] * in_scale;
            float s_ = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c0 = cbase + 256 * k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xs = (c0 + e < N) ? nx[r][k][e] : OG_NEG_INF;      // exp2(-inf) = 0
                    p[r][k][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(xs, c2, vv[k][e]) + uo[r]);
                    s_ += p[r][k][e];
                }
            }
            sum[r] = s_;
        }
        if (g0 + RG < RW) fetch(row0 + RG);
#pragma unroll
        for (int r = 0; r < RG; ++r) sum[r] = wave_sum(sum[r]);
        if (WPR > 1) {
            float* xs = xch + ((parity * NRS + rs) * RG) * WPR;
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < RG; ++r) xs[r * WPR + cp] = sum[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                float st = 0.f;
#pragma unroll
                for (int w = 0; w < WPR; ++w) st += xs[r * WPR + w];
                sum[r] = st;
            }
            parity ^= 1;
        }
        float f[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const float rowsum = sum[r] + __builtin_amdgcn_exp2f(dcol2 + uo[r]);
            const float un = uo[r] + la2 - __builtin_amdgcn_logf(rowsum);           // v_log_f32 = log2
            f[r] = __builtin_amdgcn_exp2f(un - uo[r]);
            if (cp == 0 && lane == 0 && row0 + r < M) ub[row0 + r] = un * out_scale;
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < RG; ++r) cs[k][e] = __builtin_fmaf(p[r][k][e], f[r], cs[k][e]);
    }

    float* psb = ps + ((int64_t)b * RB + rb) * ldp;
    if (NRS == 1) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c0 = cbase + 256 * k;
            if (c0 < N) *reinterpret_cast<f32x4*>(psb + c0) = f32x4{cs[k][0], cs[k][1], cs[k][2], cs[k][3]};
        }
        return;
    }
    float* smm = sm + (size_t)rs * NCB;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = cbase + 256 * k;
        *reinterpret_cast<f32x4*>(smm + c0) = f32x4{cs[k][0], cs[k][1], cs[k][2], cs[k][3]};
    }
    __syncthreads();
    for (int c0 = 4 * tid; c0 < N; c0 += 1024) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(sm + c0);
#pragma unroll
        for (int w = 1; w < NRS; ++w) acc += *reinterpret_cast<const f32x4*>(sm + (size_t)w * NCB + c0);
        *reinterpret_cast<f32x4*>(psb + c0) = acc;       // columns >= N inside the float4: never read
    }
}

// Combine of a dual-stabilised iteration: dustbin-row u, all v from the per-row-block partial column sums.
template <class RD>
__global__ __launch_bounds__(256) void sinkhorn_combine_fast_kernel(int M, int N, const float* __restrict__ zdev, float zhost,
                                                                    float inv_reg, float la_bin, float lb, float lb_bin,
                                                                    const float* __restrict__ v_in, float* __restrict__ v_out,
                                                                    int ldv, float* __restrict__ u, int ldu,
                                                                    const float* __restrict__ ps, int ldp, int RB, float in_scale,
                                                                    float out_scale, int rows, RD rd) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const float u_scale = out_scale == 1.f ? 1.f : LOG2E;      // the sweep of this iteration stored u * out_scale
    int RBv = RB;
    if (rd.B > 0) {
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        if ((int)blockIdx.x * 256 > N) return;
        RBv = (M + rows - 1) / rows;
        const float norm = -__logf((float)(M + N));
        lb = norm; la_bin = norm + __logf((float)N); lb_bin = norm + __logf((float)M);
    }
    const float zr2 = (zdev ? zdev[0] : zhost) * inv_reg * LOG2E;
    const float* vb = v_in + (int64_t)b * ldv;
    float* ub = u + (int64_t)b * ldu;
    __shared__ float smx[3][4];
    const int lane = tid & 63, wave = tid >> 6;
    const bool owns_bin = (int)blockIdx.x == N / 256;          // the block that owns column N

    // Everything that does not depend on the new u_M is issued first, so its latency overlaps the reduction below:
    // (1) the column partials of this thread's column, (2) the dustbin-column terms of the new u (owning block only).
    const int j = blockIdx.x * 256 + tid;
    float colsum = 0.f, vo = 0.f;
    if (j < N) {
        const float* psb = ps + (int64_t)b * RB * ldp + j;
        vo = vb[j] * in_scale;
        for (int rb0 = 0; rb0 < RBv; rb0 += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = rb0 + q < RBv ? psb[(int64_t)(rb0 + q) * ldp] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) colsum += t[q];
        }
    }
    const float vNo = vb[N] * in_scale;
    float us = 0.f;
    if (owns_bin)
        for (int i = tid; i < M; i += 256) us += __builtin_amdgcn_exp2f(zr2 + vNo + ub[i] * u_scale);

    // dustbin row: u_M = log a_M - (z + LSE_{j<=N} v_j) from the old v alone (every block of the pair recomputes it; a
    // form relative to the old u_M would race with the block that publishes the new one).  Max-subtracted, one exchange:
    // per-wave (max, sum) pairs through LDS, together with the dustbin-column partial sums.
    float m = OG_NEG_INF;
    for (int jj = tid; jj <= N; jj += 256) m = fmaxf(m, vb[jj]);
    m = wave_max(m) * in_scale;
    float sv = 0.f;
    if (m != OG_NEG_INF)
        for (int jj = tid; jj <= N; jj += 256) sv += __builtin_amdgcn_exp2f(vb[jj] * in_scale - m);
    sv = wave_sum(sv);
    us = wave_sum(us);
    if (lane == 0) { smx[0][wave] = m; smx[1][wave] = sv; smx[2][wave] = us; }
    __syncthreads();
    const float mm = fmaxf(fmaxf(smx[0][0], smx[0][1]), fmaxf(smx[0][2], smx[0][3]));
    float st = 0.f, ust = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        st += smx[0][w] == OG_NEG_INF ? 0.f : smx[1][w] * __builtin_amdgcn_exp2f(smx[0][w] - mm);
        ust += smx[2][w];
    }
    const float uM = la_bin * LOG2E - (zr2 + mm + __builtin_amdgcn_logf(st));      // base 2
    if (blockIdx.x == 0 && tid == 0) ub[M] = uM * out_scale;

    if (j < N) {
        const float cs = colsum + __builtin_amdgcn_exp2f(zr2 + vo + uM);      // + the dustbin row entry with the new u_M
        v_out[(int64_t)b * ldv + j] = (vo + lb * LOG2E - __builtin_amdgcn_logf(cs)) * out_scale;
    }
    if (owns_bin && tid == 0) {
        // dustbin column: v_N += log b_N - log sum_{i<=M} P_iN with the new u (u_M from above)
        const float tot = ust + __builtin_amdgcn_exp2f(zr2 + vNo + uM);
        v_out[(int64_t)b * ldv + N] = (vNo + lb_bin * LOG2E - __builtin_amdgcn_logf(tot)) * out_scale;
    }
}

// scores[b][i][j] = ((S~_ij + u_i) + v_j) - norm   (same association as optimal_transport.py:28, superglue.py:111)
template <class RD>
__global__ __launch_bounds__(256) void sinkhorn_scores_kernel(const float* __restrict__ S, int64_t lds, int M, int N,
                                                              const float* __restrict__ zdev, float zhost,
                                                              float inv_reg, float norm,
                                                              const float* __restrict__ u, int ldu,
                                                              const float* __restrict__ v, int ldv,
                                                              float* __restrict__ scores, int64_t strideS, RD rd, RowBest rb,
                                                              unsigned* __restrict__ nonfinite) {
    const int b = blockIdx.y;
    int64_t so = (int64_t)b * (M + 1) * (N + 1);
    if (rd.B > 0) {
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        norm = -__logf((float)(M + N));
        so = rd.soff[b];
    }
    S += (int64_t)b * strideS;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row > M) return;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    const float ui = u[(int64_t)b * ldu + row];
    const float* vb = v + (int64_t)b * ldv;
    float* out = scores + so + (int64_t)row * (N + 1);
    float chk = 0.f;            // stays 0 while every value written is finite (inf * 0 = NaN): one FMA per element of a memory-bound kernel
    if (row < M) {
        const float* sp = S + (int64_t)row * lds;
        if (rb.idx) {        // + row max / argmax over j < N on the values just written ("first maximal index wins", matches.hip)
            float best = OG_NEG_INF;
            int bi = 0x7FFFFFFF;
            for (int j = lane; j < N; j += 64) {
                const float val = ((sp[j] * inv_reg + ui) + vb[j]) - norm;
                out[j] = val;
                chk = fmaf(val, 0.f, chk);
                if (val > best || bi == 0x7FFFFFFF) { best = val; bi = j; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { rb.idx[(int64_t)b * rb.stride + row] = bi; rb.val[(int64_t)b * rb.stride + row] = best; }
        } else {
            for (int j = lane; j < N; j += 64) { const float val = ((sp[j] * inv_reg + ui) + vb[j]) - norm; out[j] = val; chk = fmaf(val, 0.f, chk); }
        }
    } else {
        for (int j = lane; j < N; j += 64) { const float val = ((zr + ui) + vb[j]) - norm; out[j] = val; chk = fmaf(val, 0.f, chk); }
    }
    if (lane == 0) { const float val = ((zr + ui) + vb[N]) - norm; out[N] = val; chk = fmaf(val, 0.f, chk); }
    // a non-finite value anywhere upstream (an activation beyond the binary16 range of the (hi, lo) operands, NaN inputs or weights)
    // ends up here: report it through the status word instead of handing back NaN scores silently
    if (nonfinite && __any(chk != chk) && lane == 0) atomicOr(nonfinite, 1u);
}

template <int CPL, int RG, int WPR, class RD>
void launch_sweep(const float* S, int64_t lds, int B, int m, int n, const float* zdev, float zhost, float inv_reg, float la,
                  const float* v_in, const SinkhornWs& w, hipStream_t st, const RD& rd) {
    constexpr int NRS = 4 / WPR;
    const size_t shmem = sizeof(float) * ((size_t)NRS * 2 * 256 * CPL * WPR + 2 * NRS * RG * WPR * 2);
    hipLaunchKernelGGL((sinkhorn_sweep_kernel<CPL, RG, WPR, RD>), dim3(w.RB, B), dim3(256), shmem, st, S, lds, m, n, zdev, zhost,
                       inv_reg, la, v_in, w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB, (int64_t)m * lds, rd);
}

template <int CPL, int RG, int WPR, class RD, int ROWS, bool NT>
void launch_sweep_fast_g(const float* S, int64_t lds, int B, int m, int n, const float* zdev, float zhost, float inv_reg, float la,
                         const float* v_in, const SinkhornWs& w, hipStream_t st, const RD& rd, float in_scale, float out_scale) {
    constexpr int NRS = 4 / WPR;
    const size_t shmem = sizeof(float) * ((size_t)NRS * 256 * CPL * WPR + 2 * NRS * RG * WPR);
    const int RB = (m + ROWS - 1) / ROWS;        // <= w.RB (the workspace is laid out for 32 rows per workgroup)
    hipLaunchKernelGGL((sinkhorn_sweep_fast_kernel<CPL, RG, WPR, RD, ROWS, NT>), dim3(RB, B), dim3(256), shmem, st, S, lds, m, n, zdev, zhost,
                       inv_reg, la, v_in, w.ldv, w.u, w.ldu, w.ps, w.ldp, RB, (int64_t)m * lds, in_scale, out_scale, rd);
}
// rows per workgroup / streaming loads are compile-time forms of the wide geometries (n > 1024) of UNIFORM batches only; everything
// else runs 32 rows per workgroup with plain loads
template <int CPL, int RG, int WPR, class RD>
void launch_sweep_fast(const float* S, int64_t lds, int B, int m, int n, const float* zdev, float zhost, float inv_reg, float la,
                       const float* v_in, const SinkhornWs& w, hipStream_t st, const RD& rd, float in_scale, float out_scale, int rows,
                       int nt) {
#define OG_SKF(ROWS_, NT_) launch_sweep_fast_g<CPL, RG, WPR, RD, ROWS_, NT_>(S, lds, B, m, n, zdev, zhost, inv_reg, la, v_in, w, st, rd, in_scale, out_scale)
    if constexpr (std::is_same<RD, RaggedNone>::value && WPR > 1) {
        if (rows == 128) { if (nt) OG_SKF(128, true); else OG_SKF(128, false); return; }
        if (rows == 64) { if (nt) OG_SKF(64, true); else OG_SKF(64, false); return; }
        if (nt) { OG_SKF(SK_ROWS, true); return; }
    }
    OG_SKF(SK_ROWS, false);
#undef OG_SKF
}

// Safety net of the on-chip-resident schedule (sinkhorn_resident.hip): its workgroups wait for each other with bounded spins and
// give up with status = 1 when a peer never arrives (the co-residency of B * G workgroups is inferred from the CU count; another
// stream or process holding CUs breaks it).  This kernel is enqueued right behind it; it returns at once while the status word is
// 0 and otherwise solves the problem AGAIN from u = v = 0 -- ONE workgroup per pair, all `iters` max-subtracted iterations exactly
// as optimal_transport.py:22-26 with the dustbin row / column in closed form -- leaving u, v where the scores kernel reads them
// and status = 2 ("recomputed by the fallback, scores valid").  Slow (two sweeps of the pair's matrix per iteration from one CU:
// milliseconds), never wrong.  n <= 4096 (resident shapes): up to four columns per thread; ragged batches take their sizes from rd.
template <class RD>
__global__ __launch_bounds__(1024) void sinkhorn_fallback_kernel(const float* __restrict__ S, int64_t lds, int64_t strideS, int M, int N,
                                                                 const float* __restrict__ zdev, float zhost, float inv_reg, float la,
                                                                 float la_bin, float lb, float lb_bin, float* __restrict__ u, int ldu,
                                                                 float* __restrict__ v, int ldv, int iters, unsigned* status, RD rd) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    __shared__ float uL[8196];          // m + 1 duals when they fit (else they stay in global memory)
    __shared__ float vL[4100];
    __shared__ float red[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rd.B > 0) {                     // ragged: this pair's own size and marginals (as the ragged streaming kernels)
        M = rd.off0[b + 1] - rd.off0[b];
        N = rd.off1[b + 1] - rd.off1[b];
        const float norm = -__logf((float)(M + N));
        la = norm; lb = norm; la_bin = norm + __logf((float)N); lb_bin = norm + __logf((float)M);
    }
    const float* Sb = S + (int64_t)b * strideS;
    float* ub = u + (int64_t)b * ldu;
    float* vb = v + (int64_t)b * ldv;
    float* uw = M + 1 <= 8196 ? uL : ub;
    const float zr = (zdev ? zdev[0] : zhost) * inv_reg;
    for (int j = tid; j <= N; j += 1024) vL[j] = 0.f;
    __syncthreads();
    auto block_lse = [&](float mx, float sm) -> float {      // (max, sum of exp(x - max)) per thread -> log-sum-exp over the block
        float m2 = wave_max(mx);
        sm = wave_sum(sm * (mx == OG_NEG_INF ? 0.f : expf(mx - m2)));
        if (lane == 0) { red[0][wave] = m2; red[1][wave] = sm; }
        __syncthreads();
        float bm = OG_NEG_INF, bs = 0.f;
        for (int w = 0; w < 16; ++w) bm = fmaxf(bm, red[0][w]);
        for (int w = 0; w < 16; ++w) bs += red[1][w] * (red[0][w] == OG_NEG_INF ? 0.f : expf(red[0][w] - bm));
        __syncthreads();
        return bm + logf(bs);
    };
    for (int it = 0; it < iters; ++it) {
        // rows: u_i = log a_i - LSE_j (S_ij / reg + v_j), the dustbin column (z + v_N) included; one wave per row
        for (int i = wave; i < M; i += 16) {
            const float* sp = Sb + (int64_t)i * lds;
            float mx = OG_NEG_INF;
            for (int j = lane; j < N; j += 64) mx = fmaxf(mx, sp[j] * inv_reg + vL[j]);
            mx = wave_max(fmaxf(mx, zr + vL[N]));
            float sm = 0.f;
            for (int j = lane; j < N; j += 64) sm += expf(sp[j] * inv_reg + vL[j] - mx);
            sm = wave_sum(sm) + expf(zr + vL[N] - mx);
            if (lane == 0) uw[i] = la - (mx + logf(sm));
        }
        {   // the dustbin row: every entry is z
            float mx = OG_NEG_INF, sm = 0.f;
            for (int j = tid; j <= N; j += 1024) { const float x = zr + vL[j]; const float nm = fmaxf(mx, x); sm = sm * (mx == OG_NEG_INF ? 0.f : expf(mx - nm)) + expf(x - nm); mx = nm; }
            const float l = block_lse(mx, sm);
            if (tid == 0) uw[M] = la_bin - l;
        }
        __syncthreads();                 // (also orders the global-memory u of very tall pairs inside the workgroup)
        // columns: v_j = log b_j - LSE_i (S_ij / reg + u_i), the dustbin row (z + u_M) included; columns tid, tid + 1024, ...; rows streamed
        for (int j = tid; j < N; j += 1024) {
            float mx = zr + uw[M], sm = 1.f;
            for (int i = 0; i < M; ++i) {
                const float x = Sb[(int64_t)i * lds + j] * inv_reg + uw[i];
                const float nm = fmaxf(mx, x);
                sm = sm * expf(mx - nm) + expf(x - nm);
                mx = nm;
            }
            vb[j] = lb - (mx + logf(sm));
        }
        float nv;
        {   // the dustbin column
            float mx = OG_NEG_INF, sm = 0.f;
            for (int i = tid; i <= M; i += 1024) { const float x = zr + uw[i]; const float nm = fmaxf(mx, x); sm = sm * (mx == OG_NEG_INF ? 0.f : expf(mx - nm)) + expf(x - nm); mx = nm; }
            nv = lb_bin - block_lse(mx, sm);
        }
        __syncthreads();
        for (int j = tid; j < N; j += 1024) vL[j] = vb[j];
        if (tid == 0) { vL[N] = nv; vb[N] = nv; }
        __syncthreads();
    }
    if (uw != ub) for (int i = tid; i <= M; i += 1024) ub[i] = uw[i];
    __syncthreads();
    if (tid == 0 && b == 0) { __threadfence(); __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

}  // namespace

extern "C" size_t og_sinkhorn_workspace_bytes(int32_t batch, int32_t m, int32_t n) {
    if (batch <= 0 || m <= 0 || n <= 0 || n > 8192) return 0;
    const int64_t RB = (m + SK_ROWS - 1) / SK_ROWS;
    const int64_t ldu = og_round_up(m + 1, 4), ldv = og_round_up(n + 1, 4), ldp = og_round_up(n, 4);
    const int64_t floats = (int64_t)batch * (ldu + 2 * ldv) + 4 + 2 * (int64_t)batch * RB * ldp;
    // + the exchange area of the on-chip-resident iteration kernel (sinkhorn_resident.hip), 256-byte aligned
    return (size_t)og_round_up(floats * (int64_t)sizeof(float), 256) + og_sinkhorn_resident_ws_bytes(batch, m, n);
}

static inline void* sk_resident_ws(void* ws, int B, int m, int n) {
    const int64_t RB = (m + SK_ROWS - 1) / SK_ROWS;
    const int64_t ldu = og_round_up(m + 1, 4), ldv = og_round_up(n + 1, 4), ldp = og_round_up(n, 4);
    const int64_t floats = (int64_t)B * (ldu + 2 * ldv) + 4 + 2 * (int64_t)B * RB * ldp;
    return (char*)ws + og_round_up(floats * (int64_t)sizeof(float), 256);
}

namespace {
// RD = RaggedDesc (per-pair sizes by value in the kernarg segment) or RaggedNone (uniform batch: nothing; og_common.h)
template <class RD>
int sinkhorn_run(const float* S, int64_t lds, const float* zdev, float dustbin, int B, int m, int n, int iters, float reg,
                 float* scores, void* workspace, hipStream_t st, const RD& rd, const RowBest* row_best, bool trusted_padding) {
    if (!S || !scores || !workspace || B <= 0 || m <= 0 || n <= 0 || iters < 0 || !(reg > 0.f)) return OG_E_INVALID;
    if (n > 8192) return OG_E_SHAPE;
    if ((lds & 3) || ((uintptr_t)S & 15) || ((uintptr_t)workspace & 15)) return OG_E_ALIGN;
    const SinkhornWs w = sk_layout(workspace, B, m, n);
    const float inv_reg = 1.f / reg;
    const double norm = -log((double)m + (double)n);
    const float la = (float)norm, lb = (float)norm;
    const float la_bin = (float)norm + (float)log((double)n);   // log_a[-1] += log(n) in fp32 (superglue.py:100)
    const float lb_bin = (float)norm + (float)log((double)m);
    // u = v = 0 (optimal_transport.py:22)
    hipError_t e = hipMemsetAsync(w.u, 0, sizeof(float) * ((size_t)B * (w.ldu + 2 * (size_t)w.ldv) + 4), st);      // ... and the flags behind them
    if (e != hipSuccess) return (int)e;
    const SinkhornGeom g = sk_geom(n);
    int cur = 0;
    static const bool robust_only = [] { const char* e = getenv("OG_SINKHORN_ROBUST"); return e && atoi(e) != 0; }();   // experiments
    // 0: streaming kernels only; 1 (default): on-chip-resident iterations when the batch is co-resident and big enough to pay
    // off; 2: whenever it is co-resident (tests)
    const char* rm_env = getenv("OG_SINKHORN_RESIDENT");          // read per call: the parity tests switch it
    const int resident_mode = rm_env ? atoi(rm_env) : 1;
    bool resident = !robust_only && iters > 1;
    if constexpr (std::is_same<RD, RaggedNone>::value) resident = resident && og_sinkhorn_resident_wanted(B, m, n, resident_mode);
    else resident = resident && og_sinkhorn_resident_ws_bytes(B, m, n) > 0 && og_sinkhorn_resident_ragged_wanted(rd, resident_mode);
    // Geometry of the dual-stabilised streaming sweeps (uniform batches): the largest 32 / 64 / 128 rows per workgroup that still gives
    // >= 512 workgroups -- half the column partials to write and merge per doubling (2048 columns x 32 pairs: 11.6 -> 9.2 ms per 100
    // iterations at 128 rows with streaming loads; 4096 x 8 pairs: 11.4 -> 9.5 at 64; below 512 workgroups the chip runs dry: 16.6 ms at
    // 128 workgroups) -- and non-temporal loads when the score matrices exceed the 256 MB Infinity Cache (smaller ones get SLOWER with
    // them: 3.9 -> 4.2 ms at 173 MB).  OG_SK_FAST_ROWS / OG_SK_FAST_NT override (experiments).
    int fast_rows = SK_ROWS, fast_nt = 0;
    if (std::is_same<RD, RaggedNone>::value) {
        for (int r = 2 * SK_ROWS; r <= 4 * SK_ROWS; r *= 2)
            if ((int64_t)((m + r - 1) / r) * B >= 512) fast_rows = r;
        fast_nt = (int64_t)B * m * lds * (int64_t)sizeof(float) > ((int64_t)256 << 20);
    }
    { const char* e = getenv("OG_SK_FAST_ROWS"); if (e && (atoi(e) == 32 || atoi(e) == 64 || atoi(e) == 128)) fast_rows = atoi(e); }
    { const char* e = getenv("OG_SK_FAST_NT"); if (e) fast_nt = atoi(e) != 0; }
    for (int it = 0; it < iters; ++it) {
        const float* vin = w.v[cur];
        if (it > 0 && resident) {         // iterations 2 .. iters in ONE launch, S read once (sinkhorn_resident.hip)
            unsigned* status = w.flags + 1;          // zeroed with u, v, flags[0] at the top of every call
            void* xws = sk_resident_ws(workspace, B, m, n);
            const char* ft = getenv("OG_SINKHORN_FORCE_TIMEOUT");     // tests: behave as if a peer workgroup never arrived
            if (ft && atoi(ft) != 0) {
                e = hipMemsetAsync(status, 1, sizeof(unsigned), st);
                if (e != hipSuccess) return (int)e;
            } else {
                int rc;
                if constexpr (std::is_same<RD, RaggedNone>::value)
                    rc = og_launch_sinkhorn_resident(S, lds, zdev, dustbin, B, m, n, iters - 1, inv_reg, la, la_bin, lb, lb_bin, w.u, w.ldu, w.v[cur],
                                                     w.v[cur ^ 1], w.ldv, xws, status, st, trusted_padding);
                else
                    rc = og_launch_sinkhorn_resident_ragged(S, lds, zdev, dustbin, rd, m, n, iters - 1, inv_reg, w.u, w.ldu, w.v[cur], w.v[cur ^ 1],
                                                            w.ldv, xws, status, st, trusted_padding);
                if (rc) return rc;
            }
            cur ^= 1;
            // the safety net: a no-op while status == 0, else the whole solve again by one workgroup per pair (status -> 2)
            hipLaunchKernelGGL(sinkhorn_fallback_kernel<RD>, dim3(B), dim3(1024), 0, st, S, lds, (int64_t)m * lds, m, n, zdev, dustbin, inv_reg, la,
                               la_bin, lb, lb_bin, w.u, w.ldu, w.v[cur], w.ldv, iters, status, rd);
            break;
        }
        if (it > 0 && !robust_only) {     // dual-stabilised form: valid once one max-subtracted iteration has been done
            const float is = it == 1 ? LOG2E : 1.f, os = it == iters - 1 ? LN2 : 1.f;     // duals stay in base 2 in between
            const int rows = (std::is_same<RD, RaggedNone>::value && g.WPR > 1) ? fast_rows : SK_ROWS, RBf = (m + rows - 1) / rows;
            if (g.CPL == 1) launch_sweep_fast<1, 4, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            else if (g.CPL == 2) launch_sweep_fast<2, 4, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            else if (g.CPL == 8) launch_sweep_fast<8, 1, 4>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            else if (g.WPR == 1) launch_sweep_fast<4, 2, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            else if (g.WPR == 2) launch_sweep_fast<4, 2, 2>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            else launch_sweep_fast<4, 2, 4>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd, is, os, rows, fast_nt);
            hipLaunchKernelGGL(sinkhorn_combine_fast_kernel<RD>, dim3((n + 1 + 255) / 256, B), dim3(256), 0, st, m, n, zdev, dustbin,
                               inv_reg, la_bin, lb, lb_bin, w.v[cur], w.v[cur ^ 1], w.ldv, w.u, w.ldu, w.ps, w.ldp, RBf, is, os, rows, rd);
            cur ^= 1;
            continue;
        }
        if (g.CPL == 1) launch_sweep<1, 4, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.CPL == 2) launch_sweep<2, 4, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.CPL == 8) launch_sweep<8, 1, 4>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.WPR == 1) launch_sweep<4, 2, 1>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.WPR == 2) launch_sweep<4, 2, 2>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        else launch_sweep<4, 2, 4>(S, lds, B, m, n, zdev, dustbin, inv_reg, la, vin, w, st, rd);
        hipLaunchKernelGGL(sinkhorn_combine_kernel<RD>, dim3((n + 1 + 255) / 256, B), dim3(256), 0, st, m, n, zdev, dustbin, inv_reg, la_bin,
                           lb, lb_bin, w.v[cur], w.v[cur ^ 1], w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB, rd);
        cur ^= 1;
    }
    hipLaunchKernelGGL(sinkhorn_scores_kernel<RD>, dim3((m + 1 + 3) / 4, B), dim3(256), 0, st, S, lds, m, n, zdev, dustbin, inv_reg,
                       (float)norm, w.u, w.ldu, w.v[cur], w.ldv, scores, (int64_t)m * lds, rd, row_best ? *row_best : RowBest{nullptr, nullptr, 0}, w.flags);
    return og_launch_status();
}
}  // namespace

// Training forward (sinkhorn_train.hip): max-subtracted iterations only, u_t / v_t of every iteration kept:
// U [iters][B][ldu], V [iters + 1][B][ldv] with V[0] = 0; ldu / ldv as sk_layout.
int og_launch_sinkhorn_trajectory(const float* S, int64_t lds, const float* dustbin_dev, float dustbin, int B, int m, int n, int iters, float reg, float* scores,
                                  void* workspace, float* U, float* V, hipStream_t st) {
    if (n > 8192) return OG_E_SHAPE;
    SinkhornWs w = sk_layout(workspace, B, m, n);
    const float inv_reg = 1.f / reg;
    const double norm = -log((double)m + (double)n);
    const float la = (float)norm, lb = (float)norm;
    const float la_bin = (float)norm + (float)log((double)n), lb_bin = (float)norm + (float)log((double)m);
    hipError_t e = hipMemsetAsync(V, 0, sizeof(float) * (size_t)B * w.ldv, st);          // v_0 = 0 (optimal_transport.py:22)
    if (e != hipSuccess) return (int)e;
    const SinkhornGeom g = sk_geom(n);
    const RaggedNone rd{};
    for (int it = 0; it < iters; ++it) {
        const float* vin = V + (size_t)it * B * w.ldv;
        float* vout = V + (size_t)(it + 1) * B * w.ldv;
        w.u = U + (size_t)it * B * w.ldu;                      // the sweep writes u_t of every row, the combine u_t of the dustbin row
        if (g.CPL == 1) launch_sweep<1, 4, 1>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.CPL == 2) launch_sweep<2, 4, 1>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.CPL == 8) launch_sweep<8, 1, 4>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.WPR == 1) launch_sweep<4, 2, 1>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        else if (g.WPR == 2) launch_sweep<4, 2, 2>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        else launch_sweep<4, 2, 4>(S, lds, B, m, n, dustbin_dev, dustbin, inv_reg, la, vin, w, st, rd);
        hipLaunchKernelGGL(sinkhorn_combine_kernel<RaggedNone>, dim3((n + 1 + 255) / 256, B), dim3(256), 0, st, m, n, dustbin_dev, dustbin,
                           inv_reg, la_bin, lb, lb_bin, vin, vout, w.ldv, w.u, w.ldu, w.pm, w.ps, w.ldp, w.RB, rd);
    }
    hipLaunchKernelGGL(sinkhorn_scores_kernel<RaggedNone>, dim3((m + 1 + 3) / 4, B), dim3(256), 0, st, S, lds, m, n, dustbin_dev, dustbin,
                       inv_reg, (float)norm, U + (size_t)(iters - 1) * B * w.ldu, w.ldu, V + (size_t)iters * B * w.ldv, w.ldv, scores,
                       (int64_t)m * lds, rd, RowBest{nullptr, nullptr, 0}, (unsigned*)nullptr);
    return og_launch_status();
}

int og_launch_sinkhorn(const float* S, int64_t lds, const float* zdev, float dustbin, int B, int m, int n, int iters, float reg,
                       float* scores, void* workspace, hipStream_t st, const RaggedDesc* rag, const RowBest* row_best, bool trusted_padding) {
    if (rag) {
        if (rag->B != B) return OG_E_INVALID;
        return sinkhorn_run<RaggedDesc>(S, lds, zdev, dustbin, B, m, n, iters, reg, scores, workspace, st, *rag, row_best, trusted_padding);
    }
    return sinkhorn_run<RaggedNone>(S, lds, zdev, dustbin, B, m, n, iters, reg, scores, workspace, st, RaggedNone{}, row_best, trusted_padding);
}

extern "C" int og_sinkhorn_schedule(int32_t batch, int32_t m, int32_t n, int32_t iters) {
    static const bool robust_only = [] { const char* e = getenv("OG_SINKHORN_ROBUST"); return e && atoi(e) != 0; }();
    const char* rm_env = getenv("OG_SINKHORN_RESIDENT");
    const int resident_mode = rm_env ? atoi(rm_env) : 1;
    return (!robust_only && iters > 1 && og_sinkhorn_resident_wanted(batch, m, n, resident_mode)) ? og_sinkhorn_resident_rounds(batch, m, n) : 0;
}

extern "C" int og_sinkhorn_schedule_ragged(int32_t batch, const int32_t* lens0, const int32_t* lens1, int32_t iters) {
    static const bool robust_only = [] { const char* e = getenv("OG_SINKHORN_ROBUST"); return e && atoi(e) != 0; }();
    if (batch <= 0 || batch > OG_MAX_RAGGED || !lens0 || !lens1 || robust_only || iters <= 1) return 0;
    const char* rm_env = getenv("OG_SINKHORN_RESIDENT");
    const int resident_mode = rm_env ? atoi(rm_env) : 1;
    RaggedDesc rd{};
    rd.B = batch;
    int mmax = 0, nmax = 0;
    for (int b = 0; b < batch; ++b) {
        rd.off0[b + 1] = rd.off0[b] + lens0[b]; rd.off1[b + 1] = rd.off1[b] + lens1[b];
        if (lens0[b] > mmax) mmax = lens0[b];
        if (lens1[b] > nmax) nmax = lens1[b];
    }
    if (og_sinkhorn_resident_ws_bytes(batch, mmax, nmax) == 0 || !og_sinkhorn_resident_ragged_wanted(rd, resident_mode)) return 0;
    int launches = 0;
    if (og_launch_sinkhorn_resident_ragged(nullptr, 0, nullptr, 0.f, rd, mmax, nmax, iters - 1, 1.f, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, true, &launches)) return 0;
    return launches;
}

// Reads the two status words on a private non-blocking stream -- after waiting, through an event recorded on the NULL stream, for the
// work the caller has already enqueued there; callers on other streams synchronise their stream first (openglue_amd/superglue.py drains
// the stream the call was enqueued on).  No device-wide synchronisation.  The copies land in a pinned-by-lifetime heap block, and the
// private stream is drained on EVERY path before anything it may still write to goes away (the first version returned from
// an error path with an asynchronous copy into its stack frame still pending).
extern "C" int og_sinkhorn_status(const void* workspace_dev, int32_t batch, int32_t m, int32_t n) {
    if (!workspace_dev || batch <= 0 || m <= 0 || n <= 0 || n > 8192) return -1;
    const SinkhornWs w = sk_layout(const_cast<void*>(workspace_dev), batch, m, n);
    const bool has_res = og_sinkhorn_resident_ws_bytes(batch, m, n) != 0;
    unsigned words[2] = {0, 0};                          // {non-finite flag, resident status}
    hipStream_t q = nullptr;
    if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return -1;
    int rc = 0;
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, nullptr) != hipSuccess ||
        hipStreamWaitEvent(q, ev, 0) != hipSuccess) rc = -1;
    if (!rc && hipMemcpyAsync(&words[0], w.flags, sizeof(unsigned) * (has_res ? 2 : 1), hipMemcpyDeviceToHost, q) != hipSuccess) rc = -1;
    if (hipStreamSynchronize(q) != hipSuccess) rc = -1;  // always: nothing of this call is in flight past this line
    if (ev) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(q);
    if (rc) return rc;
    if (words[0]) return 3;                              // non-finite scores were written
    if (!has_res) return 0;
    return words[1] == 0 ? 0 : words[1] == 2 ? 2 : 1;
}

extern "C" int og_sinkhorn(const float* S, int64_t lds, float dustbin, int32_t batch, int32_t m, int32_t n,
                           int32_t iters, float reg, float* scores, void* workspace_dev, void* stream) {
    og_clear_status();
    // the padding columns [n, lds) of a caller's S may hold anything (og_forward's own S buffer holds finite values there)
    return og_launch_sinkhorn(S, lds, nullptr, dustbin, batch, m, n, iters, reg, scores, workspace_dev, (hipStream_t)stream, nullptr, nullptr, false);
}

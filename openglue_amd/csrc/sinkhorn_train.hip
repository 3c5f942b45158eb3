// Training slice of the optimal-transport layer: forward that KEEPS the dual trajectory, and the
// backward pass of the unrolled log-domain Sinkhorn iterations -- what torch autograd does through
// SuperGlue.get_matching_probs + log_otp_solver (reference superglue.py:88-111, optimal_transport.py:20-28) when
// MatchingTrainingModule.training_step back-propagates the NLL of utils/losses.py:7-53 through `scores`.
//
// Forward (og_sinkhorn_train_forward): the max-subtracted kernels of sinkhorn.hip, one (sweep, combine) pair per iteration,
// writing u_t and v_t of every iteration t = 1..T into the trajectory  U [T][B][ldu], V [T+1][B][ldv]  (V[0] = 0).
//
// Backward (og_sinkhorn_backward).  With Z = S~/reg (dustbins included), la / lb the log-marginals, G = dL/dscores:
//     scores = Z + u_T + v_T - norm      =>  dZ = G,  du_T = rowsum(G),  dv_T = colsum(G)
//     v_t = lb - LSE_i(Z_ij + u_t,i)     =>  Wv_ij = exp(Z_ij + u_t,i + v_t,j - lb_j)   (the softmax weights of the LSE)
//                                            dZ_ij -= dv_t,j Wv_ij ;  du_t,i -= sum_j dv_t,j Wv_ij
//     u_t = la - LSE_j(Z_ij + v_t-1,j)   =>  Wu_ij = exp(Z_ij + v_t-1,j + u_t,i - la_i)
//                                            dZ_ij -= du_t,i Wu_ij ;  dv_t-1,j -= sum_i du_t,i Wu_ij
// for t = T .. 1 (u_t only feeds v_t and, for t = T, the output; v_0 = 0 is a constant).  The two steps of an iteration are
// ONE sweep: a row is owned by one wave, so du_t,i is complete after the row's first pass and the u-step of the same row
// follows at once; dv_t-1 is accumulated per wave in registers and added atomically per column (fp32 atomics: the summation
// order, hence the last bits of the gradients, can differ from run to run).  The augmented (m+1) x (n+1) matrix is
// materialised here (training keeps activations anyway); dS = dZ[:m,:n] / reg, d dustbin = sum of dZ over the dustbin row
// and column / reg.  Training-path code: written for clarity, not tuned.
#include <stdlib.h>

#include "og_common.h"

// sinkhorn.hip
int og_launch_sinkhorn_trajectory(const float* S, int64_t lds, const float* dustbin_dev, float dustbin, int B, int m, int n, int iters, float reg, float* scores,
                                  void* workspace, float* U, float* V, hipStream_t st);

namespace {

struct TrainWs {
    float* U;      // [T][B][ldu]
    float* V;      // [T+1][B][ldv]
    float* Za;     // [B][m+1][lda]  augmented scores / reg
    float* dZ;     // [B][m+1][lda]
    float* du;     // [B][ldu]
    float* dv[2];  // [B][ldv] ping-pong (dv_t in, dv_t-1 out)
    void* fwd;     // workspace of the forward kernels (og_sinkhorn_workspace_bytes)
    int ldu, ldv, lda;
    size_t total;
};

TrainWs tw_layout(void* ws, int B, int m, int n, int T) {
    TrainWs w{};
    w.ldu = (int)og_round_up(m + 1, 4); w.ldv = (int)og_round_up(n + 1, 4); w.lda = (int)og_round_up(n + 1, 4);
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += og_round_up((int64_t)bytes, 256); return r; };
    w.U = (float*)take(sizeof(float) * (size_t)T * B * w.ldu);
    w.V = (float*)take(sizeof(float) * (size_t)(T + 1) * B * w.ldv);
    w.Za = (float*)take(sizeof(float) * (size_t)B * (m + 1) * w.lda);
    w.dZ = (float*)take(sizeof(float) * (size_t)B * (m + 1) * w.lda);
    w.du = (float*)take(sizeof(float) * (size_t)B * w.ldu);
    w.dv[0] = (float*)take(sizeof(float) * (size_t)B * w.ldv);
    w.dv[1] = (float*)take(sizeof(float) * (size_t)B * w.ldv);
    w.fwd = take(og_sinkhorn_workspace_bytes(B, m, n));
    w.total = (size_t)(p - (char*)ws);
    return w;
}

// Za = [[S, z], [z, z]] / reg ; dZ = G ; du = rowsum(G) ; dv += colsum(G) (dv zeroed before).  One wave per row.
__global__ __launch_bounds__(256) void sk_bwd_init_kernel(const float* __restrict__ S, int64_t lds, const float* __restrict__ z_dev, float z,
                                                          float inv_reg, int M, int N,
                                                          const float* __restrict__ G, float* __restrict__ Za, float* __restrict__ dZ,
                                                          int lda, float* __restrict__ du, int ldu, float* __restrict__ dv, int ldv) {
    const int b = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row > M) return;
    if (z_dev) z = *z_dev;
    const float* g = G + ((int64_t)b * (M + 1) + row) * (N + 1);
    float* za = Za + ((int64_t)b * (M + 1) + row) * lda;
    float* dz = dZ + ((int64_t)b * (M + 1) + row) * lda;
    const float* s = S + ((int64_t)b * M + (row < M ? row : 0)) * lds;
    float acc = 0.f;
    for (int j = lane; j <= N; j += 64) {
        const float gv = g[j];
        za[j] = ((row < M && j < N) ? s[j] : z) * inv_reg;
        dz[j] = gv;
        acc += gv;
        atomicAdd(dv + (int64_t)b * ldv + j, gv);
    }
    acc = wave_sum(acc);
    if (lane == 0) du[(int64_t)b * ldu + row] = acc;
}

// One backward iteration (both half-steps) over the rows of one pair; a wave owns rows blockIdx.x*4 + wave, + 4*gridDim.x, ...
template <int CH>      // column chunks of 64 per lane: (N + 1) <= 64 * CH
__global__ __launch_bounds__(256) void sk_bwd_iter_kernel(const float* __restrict__ Za, float* __restrict__ dZ, int lda, int M, int N,
                                                          const float* __restrict__ u_t, int ldu, const float* __restrict__ v_t,
                                                          const float* __restrict__ v_prev, int ldv, float la, float la_bin, float lb,
                                                          float lb_bin, float* __restrict__ du, int du_is_fresh,
                                                          const float* __restrict__ dv_t, float* __restrict__ dv_prev) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ub = u_t + (int64_t)b * ldu;
    const float* vt = v_t + (int64_t)b * ldv;
    const float* vp = v_prev + (int64_t)b * ldv;
    const float* dvt = dv_t + (int64_t)b * ldv;
    float colacc[CH];                                            // the per-column vectors are re-read per row (L1/L2 hits): registers
#pragma unroll                                                  // hold one row of Z, one of weights and the column accumulators
    for (int c = 0; c < CH; ++c) colacc[c] = 0.f;
    for (int row = blockIdx.x * 4 + wave; row <= M; row += 4 * gridDim.x) {
        const float* za = Za + ((int64_t)b * (M + 1) + row) * lda;
        float* dz = dZ + ((int64_t)b * (M + 1) + row) * lda;
        const float ui = ub[row];
        const float lai = row < M ? la : la_bin;
        float zr[CH], wv[CH];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = lane + 64 * c;
            zr[c] = j <= N ? za[j] : OG_NEG_INF;
            const float vtl = j <= N ? vt[j] - (j < N ? lb : lb_bin) : 0.f;             // v_t,j - lb_j
            const float dvl = j <= N ? dvt[j] : 0.f;
            wv[c] = dvl * __expf(zr[c] + ui + vtl);              // dv_t,j * Wv_ij   (exp(-inf) = 0 beyond the matrix)
            acc += wv[c];
        }
        acc = wave_sum(acc);
        const float dui = (du_is_fresh ? du[(int64_t)b * ldu + row] : 0.f) - acc;      // complete du_t,i
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = lane + 64 * c;
            if (j <= N) {
                const float wu = dui * __expf(zr[c] + vp[j] + ui - lai);                // du_t,i * Wu_ij
                dz[j] -= wv[c] + wu;
                colacc[c] -= wu;
            }
        }
    }
    // the four waves of the workgroup meet in LDS first (wave order), then ONE atomic per column and workgroup: the row grid grew from 64 to 128
    // workgroups per pair (a wave walks two rows instead of four to five) with half the atomics per workgroup-row it had: backward of
    // 4 x 1024 x 1024 x 20 iterations 1.23 -> 0.93 ms
    __shared__ float red[3][CH * 64];
    if (wave) {
#pragma unroll
        for (int c = 0; c < CH; ++c) red[wave - 1][c * 64 + lane] = colacc[c];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int j = lane + 64 * c;
        const float sum = ((colacc[c] + red[0][c * 64 + lane]) + red[1][c * 64 + lane]) + red[2][c * 64 + lane];
        if (j <= N && sum != 0.f) atomicAdd(dv_prev + (int64_t)b * ldv + j, sum);
    }
}

// dS[b][i][j] = dZ[b][i][j] / reg ; d dustbin += (sum of dZ over the dustbin row and column) / reg
__global__ __launch_bounds__(256) void sk_bwd_final_kernel(const float* __restrict__ dZ, int lda, int M, int N, float inv_reg,
                                                           float* __restrict__ dS, int64_t lds, float* __restrict__ d_dustbin) {
    const int b = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row > M) return;
    const float* dz = dZ + ((int64_t)b * (M + 1) + row) * lda;
    float acc = 0.f;
    if (row < M) {
        float* ds = dS + ((int64_t)b * M + row) * lds;
        for (int j = lane; j < N; j += 64) ds[j] = dz[j] * inv_reg;
        if (lane == 0) acc = dz[N];
    } else {
        for (int j = lane; j <= N; j += 64) acc += dz[j];
    }
    acc = wave_sum(acc);
    if (lane == 0 && d_dustbin) atomicAdd(d_dustbin, acc * inv_reg);
}

}  // namespace

extern "C" size_t og_sinkhorn_train_workspace_bytes(int32_t batch, int32_t m, int32_t n, int32_t iters) {
    if (batch <= 0 || m <= 0 || n <= 0 || n > 4159 || iters < 1) return 0;       // backward: (n + 1) <= 64 * 65 columns per wave
    return tw_layout(nullptr, batch, m, n, iters).total;
}

extern "C" int og_sinkhorn_train_forward(const float* S, int64_t lds, float dustbin, const float* dustbin_dev, int32_t batch, int32_t m, int32_t n, int32_t iters,
                                         float reg, float* scores, void* train_workspace_dev, void* stream) {
    og_clear_status();
    if (!S || !scores || !train_workspace_dev || batch <= 0 || m <= 0 || n <= 0 || n > 4159 || iters < 1 || !(reg > 0.f)) return OG_E_INVALID;
    if ((lds & 3) || ((uintptr_t)S & 15) || ((uintptr_t)train_workspace_dev & 255)) return OG_E_ALIGN;
    const TrainWs w = tw_layout(train_workspace_dev, batch, m, n, iters);
    return og_launch_sinkhorn_trajectory(S, lds, dustbin_dev, dustbin, batch, m, n, iters, reg, scores, w.fwd, w.U, w.V, (hipStream_t)stream);
}

extern "C" int og_sinkhorn_backward(const float* S, int64_t lds, float dustbin, const float* dustbin_dev, int32_t batch, int32_t m, int32_t n, int32_t iters,
                                    float reg, const float* grad_scores, void* train_workspace_dev, float* dS, int64_t ldds,
                                    float* d_dustbin, void* stream) {
    og_clear_status();
    if (!S || !grad_scores || !train_workspace_dev || !dS || batch <= 0 || m <= 0 || n <= 0 || n > 4159 || iters < 1 || !(reg > 0.f)) return OG_E_INVALID;
    if ((lds & 3) || ((uintptr_t)train_workspace_dev & 255) || ldds < n) return OG_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch, T = iters;
    const TrainWs w = tw_layout(train_workspace_dev, B, m, n, T);
    const float inv_reg = 1.f / reg;
    const double norm = -log((double)m + (double)n);
    const float la = (float)norm, lb = (float)norm;
    const float la_bin = (float)norm + (float)log((double)n), lb_bin = (float)norm + (float)log((double)m);   // as og_launch_sinkhorn
    hipError_t e = hipMemsetAsync(w.dv[0], 0, sizeof(float) * (size_t)B * w.ldv, st);
    if (e != hipSuccess) return (int)e;
    if (d_dustbin && (e = hipMemsetAsync(d_dustbin, 0, sizeof(float), st)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sk_bwd_init_kernel, dim3((m + 1 + 3) / 4, B), dim3(256), 0, st, S, lds, dustbin_dev, dustbin, inv_reg, m, n, grad_scores, w.Za,
                       w.dZ, w.lda, w.du, w.ldu, w.dv[0], w.ldv);
    static const int rows_cap = [] { const char* e = getenv("OG_SK_BWD_ROWS_GRID"); return e && atoi(e) > 0 ? atoi(e) : 128; }();   // experiments (64 ... 512 measured: profiles/r05_ab_*)
    const int rows_grid = (m + 1 + 3) / 4 < rows_cap ? (m + 1 + 3) / 4 : rows_cap;       // waves stride over the rows
    int cur = 0;
    for (int t = T; t >= 1; --t) {
        const float* u_t = w.U + (size_t)(t - 1) * B * w.ldu;
        const float* v_t = w.V + (size_t)t * B * w.ldv;
        const float* v_prev = w.V + (size_t)(t - 1) * B * w.ldv;
        if ((e = hipMemsetAsync(w.dv[cur ^ 1], 0, sizeof(float) * (size_t)B * w.ldv, st)) != hipSuccess) return (int)e;
        const int fresh = t == T ? 1 : 0;                       // du_T = rowsum(G); du_t = 0 for t < T before the v-step adds to it
        const int ch = (n + 1 + 63) / 64;
#define OG_SKB(CH_)                                                                                                              \
        hipLaunchKernelGGL(sk_bwd_iter_kernel<CH_>, dim3(rows_grid, B), dim3(256), 0, st, w.Za, w.dZ, w.lda, m, n, u_t, w.ldu, v_t, \
                           v_prev, w.ldv, la, la_bin, lb, lb_bin, w.du, fresh, w.dv[cur], w.dv[cur ^ 1])
        if (ch <= 2) OG_SKB(2);
        else if (ch <= 4) OG_SKB(4);
        else if (ch <= 9) OG_SKB(9);
        else if (ch <= 17) OG_SKB(17);
        else if (ch <= 33) OG_SKB(33);
        else OG_SKB(65);
#undef OG_SKB
        cur ^= 1;
    }
    hipLaunchKernelGGL(sk_bwd_final_kernel, dim3((m + 1 + 3) / 4, B), dim3(256), 0, st, w.dZ, w.lda, m, n, inv_reg, dS, ldds, d_dustbin);
    return og_launch_status();
}

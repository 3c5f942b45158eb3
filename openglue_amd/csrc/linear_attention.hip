// Linear ("elu + 1") attention, the O(N) alternative selectable with attention_gnn.attention = 'linear'
// (reference models/superglue/attention.py:22-40, __init__.py:17-18):
//     q' = elu(q) + 1 + 1e-6,  k' = elu(k) + 1 + 1e-6            (per head, NO d^-1/2 scale)
//     kv[dk][dv] = sum_keys k'[key][dk] v[key][dv],  ksum[dk] = sum_keys k'[key][dk]
//     out[q][dv] = (q'[q] . kv[:, dv]) / (q'[q] . ksum)
// One workgroup per (problem, head): phase 1 reduces all keys into the d x d matrix kv (fp32 FMAs, operands
// rebuilt from the split-f16 planes: hi + lo), phase 2 streams the queries.  The work is O(N d^2)
// (8 MFLOP per head at 1024 keys, d = 64), i.e. negligible next to the GEMMs; plain VALU, no matrix cores.
// Same problem descriptors / plane I/O as attention.hip.
#include "og_common.h"

namespace {

constexpr float ELU_EPS = 1e-6f;

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.f + ELU_EPS; }   // F.elu(x) + 1 + eps

template <int DH>
__global__ __launch_bounds__(256) void linear_attention_kernel(AttnArgs a, RaggedDesc rd) {
    constexpr int KT = 64;                              // keys / queries per LDS tile
    __shared__ float ks[KT][DH + 1];
    __shared__ float vs[KT][DH + 1];
    __shared__ float kv[DH][DH + 1];
    __shared__ float ksum[DH];
    const int grp = blockIdx.x;
    const int z = grp / a.num_heads, h = grp - z * a.num_heads;
    const int gsel = z < a.split ? 0 : 1;
    const int zz = gsel ? z - a.split : z;
    int nq = a.nq[gsel], nk = a.nk[gsel];
    int64_t q_row0 = a.q_base[gsel] + (int64_t)zz * a.q_step[gsel];
    int64_t kv_row0 = a.kv_base[gsel] + (int64_t)zz * a.kv_step[gsel];
    if (rd.B > 0) {
        const int T0 = rd.off0[rd.B];
        const int b = z < rd.B ? z : z - rd.B;
        const int r0 = rd.off0[b], m_b = rd.off0[b + 1] - r0;
        const int r1 = T0 + rd.off1[b], n_b = rd.off1[b + 1] - rd.off1[b];
        const bool q_is0 = a.rag_mode == 1 ? z < rd.B : a.rag_mode == 2;
        const bool kv_is0 = a.rag_mode == 1 ? q_is0 : !q_is0;
        q_row0 = q_is0 ? r0 : r1; nq = q_is0 ? m_b : n_b;
        kv_row0 = kv_is0 ? r0 : r1; nk = kv_is0 ? m_b : n_b;
    }
    const int tid = threadIdx.x;
    // phase 1: thread (dk = tid % DH, part = tid / DH) accumulates kv[dk][dv] for dv in its part
    constexpr int PARTS = 256 / DH;                     // 4 (DH=64), 8 (32), 16 (16)
    constexpr int DVP = DH / PARTS;                     // dv per thread: 16, 4, 1
    const int dk = tid % DH, part = tid / DH;
    float acc[DVP];
#pragma unroll
    for (int i = 0; i < DVP; ++i) acc[i] = 0.f;
    float kacc = 0.f;
    for (int k0 = 0; k0 < nk; k0 += KT) {
        for (int i = tid; i < KT * DH; i += 256) {
            const int r = i / DH, c = i % DH;
            float kx = 0.f, vx = 0.f;
            if (k0 + r < nk) {
                const int64_t ko = (kv_row0 + k0 + r) * a.ldk + h * DH + c, vo = (kv_row0 + k0 + r) * a.ldv + h * DH + c;
                kx = elu1((float)a.kh[ko] + (float)a.kl[ko]);
                vx = (float)a.vh[vo] + (float)a.vl[vo];
            }
            ks[r][c] = kx; vs[r][c] = vx;               // rows beyond nk: k' = 0 contributes nothing
        }
        __syncthreads();
        for (int r = 0; r < KT; ++r) {
            const float kk = ks[r][dk];
            if (part == 0) kacc += kk;
#pragma unroll
            for (int i = 0; i < DVP; ++i) acc[i] += kk * vs[r][part * DVP + i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < DVP; ++i) kv[dk][part * DVP + i] = acc[i];
    if (part == 0) ksum[dk] = kacc;
    __syncthreads();
    // phase 2: queries in tiles of KT; thread (qi = tid / 4 within the tile, dv quarter = tid % 4)
    constexpr int QP = 4, DVQ = DH / QP;
    for (int q0 = 0; q0 < nq; q0 += KT) {
        for (int i = tid; i < KT * DH; i += 256) {
            const int r = i / DH, c = i % DH;
            float qx = 0.f;
            if (q0 + r < nq) {
                const int64_t qo = (q_row0 + q0 + r) * a.ldq + h * DH + c;
                qx = elu1((float)a.qh[qo] + (float)a.ql[qo]);
            }
            ks[r][c] = qx;
        }
        __syncthreads();
        const int qi = tid / QP, dq = tid % QP;
        float o[DVQ], nrm = 0.f;
#pragma unroll
        for (int i = 0; i < DVQ; ++i) o[i] = 0.f;
        for (int d = 0; d < DH; ++d) {
            const float qq = ks[qi][d];
            nrm += qq * ksum[d];
#pragma unroll
            for (int i = 0; i < DVQ; ++i) o[i] += qq * kv[d][dq * DVQ + i];
        }
        if (q0 + qi < nq) {
            const int64_t orow = (q_row0 + q0 + qi) * a.ldo;
#pragma unroll
            for (int i = 0; i < DVQ; ++i) {
                const float v = o[i] / nrm;
                const int c = h * DH + dq * DVQ + i;
                const int64_t oo = orow + (a.o_hl ? og_hl_col(c) : (int64_t)c);
                og_split(v, a.oh[oo], a.ol[oo]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// attention = 'favor_relu' (reference attention.py:43-95, __init__.py:19-25): generalised FAVOR+ attention,
//     phi(x) = relu(P (x d^-1/4)) + 1e-8,   out = phi(q) (phi(k)^T v) / (phi(q) . sum_keys phi(k)),
// one head of size d = D (the reference's matmul of the [2D, D] buffer P with the per-head tensors only type-checks for
// num_heads == 1), F = 2D random features.  The projection and the ReLU already happened in the q / k projection GEMM
// (og_pack_weights folds d^-1/4 P into in_proj_q / in_proj_k; api.hip), so q and k arrive as F-column feature planes and only
// "+ eps" is left of the feature map.  This is linear attention (above) with F != dv and a kv matrix of F x D floats (512 KB at
// D = 256), which does not fit one CU: a workgroup owns (problem, 32-column slice of v / out), keeps kv[F][32] and ksum[F] in LDS,
// phase 1 streams the keys (16 per tile; thread t accumulates features t and t + 256), phase 2 the queries.  O(N F D) flops on
// the vector ALUs; this variant is a config option of the reference, not the headline path.
constexpr float FAVOR_EPS = 1e-8f;
constexpr int FAVOR_MAXF = 512, FAVOR_CH = 32, FAVOR_KT = 16;

__global__ __launch_bounds__(256) void favor_attention_kernel(AttnArgs a, RaggedDesc rd) {
    __shared__ float fs[FAVOR_KT][FAVOR_MAXF + 1];      // feature rows of the current key / query tile
    __shared__ float vs[FAVOR_KT][FAVOR_CH + 1];
    __shared__ float kv[FAVOR_MAXF][FAVOR_CH + 1];
    __shared__ float ksum[FAVOR_MAXF];
    const int F = a.feat, D = a.dh;
    const int slices = D / FAVOR_CH;
    const int z = blockIdx.x / slices, c0 = (blockIdx.x - z * slices) * FAVOR_CH;
    const int gsel = z < a.split ? 0 : 1;
    const int zz = gsel ? z - a.split : z;
    int nq = a.nq[gsel], nk = a.nk[gsel];
    int64_t q_row0 = a.q_base[gsel] + (int64_t)zz * a.q_step[gsel];
    int64_t kv_row0 = a.kv_base[gsel] + (int64_t)zz * a.kv_step[gsel];
    if (rd.B > 0) {
        const int T0 = rd.off0[rd.B];
        const int b = z < rd.B ? z : z - rd.B;
        const int r0 = rd.off0[b], m_b = rd.off0[b + 1] - r0;
        const int r1 = T0 + rd.off1[b], n_b = rd.off1[b + 1] - rd.off1[b];
        const bool q_is0 = a.rag_mode == 1 ? z < rd.B : a.rag_mode == 2;
        const bool kv_is0 = a.rag_mode == 1 ? q_is0 : !q_is0;
        q_row0 = q_is0 ? r0 : r1; nq = q_is0 ? m_b : n_b;
        kv_row0 = kv_is0 ? r0 : r1; nk = kv_is0 ? m_b : n_b;
    }
    const int tid = threadIdx.x;
    // ---- phase 1: kv[f][c] = sum_keys phi(k)[key][f] v[key][c0 + c],  ksum[f] = sum_keys phi(k)[key][f] ----
    float acc[2][FAVOR_CH];
    float kacc[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < FAVOR_CH; ++c) acc[i][c] = 0.f;
    for (int k0 = 0; k0 < nk; k0 += FAVOR_KT) {
        for (int i = tid; i < FAVOR_KT * F; i += 256) {
            const int r = i / F, f = i - r * F;
            float x = 0.f;                                   // rows beyond nk contribute nothing
            if (k0 + r < nk) {
                const int64_t o = (kv_row0 + k0 + r) * a.ldk + f;
                x = ((float)a.kh[o] + (float)a.kl[o]) + FAVOR_EPS;
            }
            fs[r][f] = x;
        }
        for (int i = tid; i < FAVOR_KT * FAVOR_CH; i += 256) {
            const int r = i / FAVOR_CH, c = i - r * FAVOR_CH;
            float x = 0.f;
            if (k0 + r < nk) {
                const int64_t o = (kv_row0 + k0 + r) * a.ldv + c0 + c;
                x = (float)a.vh[o] + (float)a.vl[o];
            }
            vs[r][c] = x;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            if (f < F) {
                for (int r = 0; r < FAVOR_KT; ++r) {
                    const float kk = fs[r][f];
                    kacc[i] += kk;
#pragma unroll
                    for (int c = 0; c < FAVOR_CH; ++c) acc[i][c] += kk * vs[r][c];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        if (f < F) {
#pragma unroll
            for (int c = 0; c < FAVOR_CH; ++c) kv[f][c] = acc[i][c];
            ksum[f] = kacc[i];
        }
    }
    __syncthreads();
    // ---- phase 2: queries, 16 per tile; thread (query tid / 16, columns 2 (tid % 16), +1) ----
    const int qi = tid >> 4, cq = (tid & 15) * 2;
    for (int q0 = 0; q0 < nq; q0 += FAVOR_KT) {
        for (int i = tid; i < FAVOR_KT * F; i += 256) {
            const int r = i / F, f = i - r * F;
            float x = 0.f;
            if (q0 + r < nq) {
                const int64_t o = (q_row0 + q0 + r) * a.ldq + f;
                x = ((float)a.qh[o] + (float)a.ql[o]) + FAVOR_EPS;
            }
            fs[r][f] = x;
        }
        __syncthreads();
        float o0 = 0.f, o1 = 0.f, nrm = 0.f;
        for (int f = 0; f < F; ++f) {
            const float qq = fs[qi][f];
            nrm += qq * ksum[f];
            o0 += qq * kv[f][cq];
            o1 += qq * kv[f][cq + 1];
        }
        if (q0 + qi < nq) {
            const int64_t orow = (q_row0 + q0 + qi) * a.ldo;
            const float v0 = o0 / nrm, v1 = o1 / nrm;
            const int c = c0 + cq;
            const int64_t oo = orow + (a.o_hl ? og_hl_col(c) : (int64_t)c);     // c is even: c and c + 1 share a 32-channel group
            og_split(v0, a.oh[oo], a.ol[oo]);
            og_split(v1, a.oh[oo + 1], a.ol[oo + 1]);
        }
        __syncthreads();
    }
}

}  // namespace

int og_launch_favor_attention(const AttnArgs& a, hipStream_t stream) {
    if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.oh || !a.ol || a.nz <= 0) return OG_E_INVALID;
    if (a.num_heads != 1 || a.dh <= 0 || a.dh % FAVOR_CH || a.feat <= 0 || a.feat > FAVOR_MAXF) return OG_E_SHAPE;
    RaggedDesc rd;
    rd.B = 0;
    if (a.rag) rd = *a.rag;
    AttnArgs a2 = a;
    a2.rag = nullptr;
    hipLaunchKernelGGL(favor_attention_kernel, dim3(a.nz * (a.dh / FAVOR_CH)), dim3(256), 0, stream, a2, rd);
    return og_launch_status();
}

int og_launch_linear_attention(const AttnArgs& a, hipStream_t stream) {
    if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.oh || !a.ol || a.nz <= 0 || a.num_heads <= 0) return OG_E_INVALID;
    RaggedDesc rd;
    rd.B = 0;
    if (a.rag) rd = *a.rag;
    AttnArgs a2 = a;
    a2.rag = nullptr;
    dim3 grid(a.nz * a.num_heads), block(256);
    switch (a.dh) {
        case 16: hipLaunchKernelGGL(linear_attention_kernel<16>, grid, block, 0, stream, a2, rd); break;
        case 32: hipLaunchKernelGGL(linear_attention_kernel<32>, grid, block, 0, stream, a2, rd); break;
        case 64: hipLaunchKernelGGL(linear_attention_kernel<64>, grid, block, 0, stream, a2, rd); break;
        default: return OG_E_SHAPE;
    }
    return og_launch_status();
}

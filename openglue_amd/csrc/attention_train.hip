// Backward of multi-head softmax attention for the training path (reference models/superglue/attention.py:8-19 under autograd),
// flash style: the attention matrix is RECOMPUTED tile by tile in registers and never written -- neither in the forward
// (attention.hip) nor here.  Exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) throughout.
//
//   out = softmax(scale Q K^T) V      per (pair b, head h), q/out [B][nq][D], k/v [B][nk][D] token-major, head h = columns [h dh, (h+1) dh)
//
//   L_i     = log sum_j exp(scale q_i.k_j)                                   attention_lse_kernel (one pass over the keys)
//   P_ij    = exp(scale q_i.k_j - L_i)
//   dV_j    = sum_i P_ij dO_i
//   dP_ij   = dO_i . v_j,   delta_i = dO_i . out_i (host side: one elementwise product + row sum)
//   dS_ij   = scale P_ij (dP_ij - delta_i)
//   dK_j    = sum_i dS_ij q_i,      dQ_i = sum_j dS_ij k_j
//
// attention_bwd_kernel: one workgroup per (pair, head, 64-key block): K_j, V_j stay in LDS, the 64-query blocks stream through
// (register prefetch of the next block during the five products of the current one).  Wave w owns the 32 x 32 block
// (query half w >> 1, key half w & 1) of S and dP.  The MFMA accumulator layout (lane = column, registers = rows) of P and dS IS the
// A-operand layout of P^T dO and dS^T Q when the contraction runs over the rows in register order (k order inside an MFMA chain is
// free as long as the B operand follows it), so dV and dK take P and dS straight from the accumulators; only dQ = dS K needs dS with
// lane = query: one 32 x 32 transpose through LDS per wave.  dQ leaves as one partial per key block (summed by the caller):
// deterministic, no atomics.
#include "og_common.h"

namespace {

constexpr int BR = 64;     // queries per inner step
constexpr int BC = 64;     // keys per workgroup

struct AttnTrainArgs {
    const float* q; const float* k; const float* v; const float* dout; const float* lse; const float* delta;
    float* dq_part; float* dk; float* dv; float* lse_out;
    int B, nq, nk, H, D;
    float scale;
    int64_t ldq, ldk, ldv, lddk, lddv;     // row strides (floats) of q, k, v and of the dk, dv outputs; dout and dq_part rows are D wide
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// rows [row0, row0 + 64) x columns [0, DH) of a token-major matrix (row stride D) as float4 pieces: piece p of thread tid
template <int DH, int DHP>
struct TileIO {
    static constexpr int C4 = DHP / 4;            // float4 per LDS row
    static constexpr int RPP = 256 / C4;          // rows per pass
    static constexpr int PASSES = 64 / RPP;
    static constexpr int LD = DHP + 4;
    __device__ static f32x4 load(const float* __restrict__ base, int64_t ld, int row0, int nrows, int tid, int p) {
        const int r = tid / C4 + RPP * p, c = (tid % C4) * 4;
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (row0 + r < nrows && c < DH) v = *reinterpret_cast<const f32x4*>(base + (int64_t)(row0 + r) * ld + c);
        return v;
    }
    __device__ static void store(float* lds, int tid, int p, f32x4 v) {
        const int r = tid / C4 + RPP * p, c = (tid % C4) * 4;
        *reinterpret_cast<f32x4*>(&lds[r * LD + c]) = v;
    }
};

// acc += X[32 rows of xrow0][:] . Y[32 rows of yrow0][:]^T   (both tiles row-major in LDS, contraction over the DHP columns)
template <int DHP>
__device__ __forceinline__ f32x16 rows_dot_rows(const float* X, int xrow0, const float* Y, int yrow0, int lane) {
    constexpr int LD = DHP + 4;
    f32x16 acc = zero16();
    const float* xp = X + (xrow0 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* yp = Y + (yrow0 + (lane & 31)) * LD + (lane >> 5) * 4;
    // every fragment requested before the first MFMA (the workgroup is one wave per SIMD: nothing else hides an LDS round trip in front of a chain
    // of dependent MFMAs; read-then-use per k-step; the compiler still sinks part of them, forcing the order with sched_barrier was no faster: 182 vs 177 us per launch)
    constexpr int KK = DHP / 8;
    f32x4 a[KK], b[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a[kk] = *reinterpret_cast<const f32x4*>(xp + kk * 8);
        b[kk] = *reinterpret_cast<const f32x4*>(yp + kk * 8);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][e], b[kk][e], acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------------------------------------------------------
// L[b][h][i] = log sum_j exp(scale q_i . k_j): one workgroup per (pair, head, 64-query block).  S^T blocks (A = keys, B = queries):
// lane = query, registers = keys, so the running max / sum of a query live in its lane.
template <int DH>
__global__ __launch_bounds__(256) void attention_lse_kernel(AttnTrainArgs a) {
    constexpr int DHP = DH < 32 ? 32 : DH;
    using IO = TileIO<DH, DHP>;
    constexpr int LD = IO::LD;
    __shared__ __attribute__((aligned(16))) float Qs[BR * LD];
    __shared__ __attribute__((aligned(16))) float Ks[BC * LD];
    __shared__ float Ms[2][BR], Ss[2][BR];

    const int nqb = (a.nq + BR - 1) / BR;
    const int z = blockIdx.x / nqb, ib = blockIdx.x - z * nqb;
    const int b = z / a.H, h = z - b * a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, kj = wave & 1;
    const int i0 = ib * BR;
    const float* qz = a.q + (int64_t)b * a.nq * a.ldq + h * DH;
    const float* kz = a.k + (int64_t)b * a.nk * a.ldk + h * DH;

#pragma unroll
    for (int p = 0; p < IO::PASSES; ++p) IO::store(Qs, tid, p, IO::load(qz, a.ldq, i0, a.nq, tid, p));
    f32x4 rk[IO::PASSES];
#pragma unroll
    for (int p = 0; p < IO::PASSES; ++p) rk[p] = IO::load(kz, a.ldk, 0, a.nk, tid, p);

    float m = -INFINITY, l = 0.f;
    const int nkb = (a.nk + BC - 1) / BC;
    for (int jb = 0; jb < nkb; ++jb) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < IO::PASSES; ++p) IO::store(Ks, tid, p, rk[p]);
        __syncthreads();
        if (jb + 1 < nkb) {
#pragma unroll
            for (int p = 0; p < IO::PASSES; ++p) rk[p] = IO::load(kz, a.ldk, (jb + 1) * BC, a.nk, tid, p);
        }
        const f32x16 st = rows_dot_rows<DHP>(Ks, 32 * kj, Qs, 32 * qi, lane);      // [key][query]: lane = query
        float mx = -INFINITY;
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = jb * BC + 32 * kj + mfma32_row(r, lane);
            sv[r] = key < a.nk ? st[r] * a.scale : -INFINITY;
            mx = fmaxf(mx, sv[r]);
        }
        const float mn = fmaxf(m, mx);
        if (mn > -INFINITY) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += __expf(sv[r] - mn);
            l = l * __expf(m - mn) + sum;
            m = mn;
        }
    }
    // the two lane halves hold different keys of the same query, so do the two key-half waves
    {
        const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
        const float mn = fmaxf(m, mo);
        l = (m > -INFINITY ? l * __expf(m - mn) : 0.f) + (mo > -INFINITY ? lo * __expf(mo - mn) : 0.f);
        m = mn;
    }
    if (lane < 32) { Ms[kj][32 * qi + lane] = m; Ss[kj][32 * qi + lane] = l; }
    __syncthreads();
    if (tid < BR && i0 + tid < a.nq) {
        const float m0 = Ms[0][tid], m1 = Ms[1][tid];
        const float mn = fmaxf(m0, m1);
        const float ls = (m0 > -INFINITY ? Ss[0][tid] * __expf(m0 - mn) : 0.f) + (m1 > -INFINITY ? Ss[1][tid] * __expf(m1 - mn) : 0.f);
        a.lse_out[((int64_t)b * a.H + h) * a.nq + i0 + tid] = mn + __logf(ls);
    }
}

// ------------------------------------------------------------------------------------------------------------------
template <int DH>
#ifndef OG_ATTN_BWD_WGS
#define OG_ATTN_BWD_WGS 2     // workgroups per CU the kernel is built for (1 = rounds 3-5: 105 KB of LDS, ~300 registers)
#endif
__global__ __launch_bounds__(256, OG_ATTN_BWD_WGS) void attention_bwd_kernel(AttnTrainArgs a) {
    constexpr int DHP = DH < 32 ? 32 : DH;
    constexpr int NB = DHP / 32;                  // 32-column blocks of a head
    using IO = TileIO<DH, DHP>;
    constexpr int LD = IO::LD;
    constexpr int LDT = 32 + 4;                   // per-wave 32 x 32 transpose tile
    __shared__ __attribute__((aligned(16))) float Ks[BC * LD];
    __shared__ __attribute__((aligned(16))) float Vs[BC * LD];
    __shared__ float Ls[BR], Ds[BR];
    // Round 6: two workgroups per CU (the kernel is one wave per SIMD with three barriers per block: the second workgroup is what fills the
    // matrix pipe under them -- the lesson of the fp32 GEMM's tile work).  By LDS that needs <= 80 KB; at dh = 64 the Q and dO tiles, the per-wave
    // transpose tiles (dS with lane = query) and the hand-over area of the partial sums added up to 105 KB.  The last two now live INSIDE the Q | dO
    // area, which is dead by then: every wave has its B operands of the step (bo, bq) in registers before the first tile write, and a barrier
    // separates the last read of Q / dO by any wave from the first write over them.  Hand-over area = the first 4096 floats (inside Q), transpose
    // tiles = the last 4608 (the end of Q and all of dO): disjoint, so a wave may hand its dQ block over while another still reads its tile.  70 KB.
    constexpr bool ALIAS = DH == 64;
    constexpr int TSZ = 4 * 32 * LDT, RSZ = 2 * 32 * DHP;
    static_assert(!ALIAS || (RSZ + TSZ <= 2 * BR * LD), "hand-over area and transpose tiles side by side inside Q | dO");
    __shared__ __attribute__((aligned(16))) float QO[2 * BR * LD + (ALIAS ? 0 : TSZ + RSZ)];
    float* const Qs = QO;
    float* const Os = QO + BR * LD;                                     // dO
    float* const Ts = ALIAS ? QO + 2 * BR * LD - TSZ : QO + 2 * BR * LD;          // dS, lane = query, one tile per wave
    float* const Rs = ALIAS ? QO : QO + 2 * BR * LD + TSZ;              // partial sums handed from one wave of a pair to the other
    const int nkb = (a.nk + BC - 1) / BC;
    const int z = blockIdx.x / nkb, jb = blockIdx.x - z * nkb;
    const int b = z / a.H, h = z - b * a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qi = wave >> 1, kj = wave & 1;
    const int j0 = jb * BC;
    const float* qz = a.q + (int64_t)b * a.nq * a.ldq + h * DH;
    const float* doz = a.dout + (int64_t)b * a.nq * a.D + h * DH;
    const float* kz = a.k + (int64_t)b * a.nk * a.ldk + h * DH;
    const float* vz = a.v + (int64_t)b * a.nk * a.ldv + h * DH;
    const float* lz = a.lse + ((int64_t)b * a.H + h) * a.nq;
    const float* dz = a.delta + (int64_t)b * a.nq * a.H + h;            // [B][nq][H]

#pragma unroll
    for (int p = 0; p < IO::PASSES; ++p) {
        IO::store(Ks, tid, p, IO::load(kz, a.ldk, j0, a.nk, tid, p));
        IO::store(Vs, tid, p, IO::load(vz, a.ldv, j0, a.nk, tid, p));
    }
    f32x4 rq[IO::PASSES], ro[IO::PASSES];
    float rl = 0.f, rd = 0.f;
    auto prefetch = [&](int i0) {
#pragma unroll
        for (int p = 0; p < IO::PASSES; ++p) {
            rq[p] = IO::load(qz, a.ldq, i0, a.nq, tid, p);
            ro[p] = IO::load(doz, a.D, i0, a.nq, tid, p);
        }
        if (tid < BR) {
            const bool ok = i0 + tid < a.nq;
            rl = ok ? lz[i0 + tid] : 0.f;
            rd = ok ? dz[(int64_t)(i0 + tid) * a.H] : 0.f;
        }
    };
    prefetch(0);

    f32x16 dv[NB], dk[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { dv[nb] = zero16(); dk[nb] = zero16(); }
    const bool key_ok = j0 + 32 * kj + l31 < a.nk;
    float* T = Ts + wave * 32 * LDT;

    const int nqb = (a.nq + BR - 1) / BR;
    for (int ib = 0; ib < nqb; ++ib) {
        const int i0 = ib * BR;
        __syncthreads();                          // the previous step is done with Qs / Os / Rs
#pragma unroll
        for (int p = 0; p < IO::PASSES; ++p) { IO::store(Qs, tid, p, rq[p]); IO::store(Os, tid, p, ro[p]); }
        if (tid < BR) { Ls[tid] = rl; Ds[tid] = rd; }
        __syncthreads();
        if (ib + 1 < nqb) prefetch(i0 + BR);

        // S and dP blocks: rows = queries 32 qi.., lane = key 32 kj + l31
        const f32x16 s = rows_dot_rows<DHP>(Qs, 32 * qi, Ks, 32 * kj, lane);
        const f32x16 dp = rows_dot_rows<DHP>(Os, 32 * qi, Vs, 32 * kj, lane);
        // (operands of the next MFMA block are requested from LDS before the arithmetic that precedes it: see rows_dot_rows)
        float p[16], ds[16], lr[16], dr[16], bo[16][NB], bq[16][NB];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * qi + mfma32_row(r, lane);
            lr[r] = Ls[row];
            dr[r] = Ds[row];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * qi + mfma32_row(r, lane);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bo[r][nb] = Os[row * LD + nb * 32 + l31];
                bq[r][nb] = Qs[row * LD + nb * 32 + l31];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * qi + mfma32_row(r, lane);
            const bool ok = key_ok && (i0 + row < a.nq);
            const float e = __expf(s[r] * a.scale - lr[r]);       // computed for every lane, selected afterwards: no branch per register
            const float pv = ok ? e : 0.f;
            p[r] = pv;
            ds[r] = pv * (dp[r] - dr[r]) * a.scale;
        }
        // dV += P^T dO, dK += dS^T Q: accumulator register r of lane half `hi` is query row rho = mfma32_row(r, lane); chain step r
        // contracts the two rows rho(r, 0), rho(r, 1), the B operand reads the same rows of dO / Q
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                dv[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[r], bo[r][nb], dv[nb], 0, 0, 0);
                dk[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], bq[r][nb], dk[nb], 0, 0, 0);
            }
        }
        // dQ block = dS K: dS with lane = query through this wave's LDS tile (dh = 64: inside the Q | dO area: every wave is past its reads of both)
        if constexpr (ALIAS) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) T[mfma32_row(r, lane) * LDT + l31] = ds[r];
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the tile is read back by the same wave
        __builtin_amdgcn_wave_barrier();
        f32x16 dq[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dq[nb] = zero16();
        f32x4 av[4];
        float kb[4][4][NB];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av[kk] = *reinterpret_cast<const f32x4*>(&T[l31 * LDT + kk * 8 + hi * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 32 * kj + kk * 8 + hi * 4 + e;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) kb[kk][e][nb] = Ks[key * LD + nb * 32 + l31];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) dq[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][e], kb[kk][e][nb], dq[nb], 0, 0, 0);
        // the two key halves of a query half add up: wave kj = 1 hands its block over, wave kj = 0 writes the partial of this key block
        float* R = Rs + qi * 32 * DHP;
        if (kj == 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[mfma32_row(r, lane) * DHP + nb * 32 + l31] = dq[nb][r];
        }
        __syncthreads();
        if (kj == 0) {
            float* out = a.dq_part + (((int64_t)jb * a.B + b) * a.nq) * a.D + h * DH;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = nb * 32 + l31;
                if (col < DH) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mfma32_row(r, lane);
                        const int row = i0 + 32 * qi + rr;
                        if (row < a.nq) out[(int64_t)row * a.D + col] = dq[nb][r] + R[rr * DHP + col];
                    }
                }
            }
        }
    }
    // dV, dK of this key block: the two query halves add up (wave qi = 1 -> wave qi = 0)
    auto flush = [&](f32x16 (&acc)[NB], float* dst, int64_t ldd) {
        __syncthreads();
        float* R = Rs + kj * 32 * DHP;
        if (qi == 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[mfma32_row(r, lane) * DHP + nb * 32 + l31] = acc[nb][r];
        }
        __syncthreads();
        if (qi == 0) {
            float* out = dst + (int64_t)b * a.nk * ldd + h * DH;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = nb * 32 + l31;
                if (col < DH) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mfma32_row(r, lane);
                        const int row = j0 + 32 * kj + rr;
                        if (row < a.nk) out[(int64_t)row * ldd + col] = acc[nb][r] + R[rr * DHP + col];
                    }
                }
            }
        }
    };
    flush(dv, a.dv, a.lddv);
    flush(dk, a.dk, a.lddk);
}

int check(const AttnTrainArgs& a, int dh) {
    if (!a.q || !a.k || a.B <= 0 || a.nq <= 0 || a.nk <= 0 || a.H <= 0) return OG_E_INVALID;
    if (dh != 16 && dh != 32 && dh != 64) return OG_E_SHAPE;
    if (a.D != a.H * dh) return OG_E_SHAPE;
    if (((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15)) return OG_E_ALIGN;
    if ((a.ldq & 3) || (a.ldk & 3) || a.ldq < a.D || a.ldk < a.D) return OG_E_ALIGN;
    return 0;
}

// delta[row][h] = sum_c dout[row][h dh + c] * out[row][h dh + c]: the row term of the softmax backward (one thread per (row, head))
__global__ __launch_bounds__(256) void attention_delta_kernel(const float* __restrict__ dout, const float* __restrict__ out, int64_t rows, int H,
                                                              int dh, float* __restrict__ delta) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * H) return;
    const float* a = dout + i * dh;
    const float* b = out + i * dh;
    float s = 0.f;
    for (int c = 0; c < dh; c += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + c), y = *reinterpret_cast<const f32x4*>(b + c);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    delta[i] = s;
}

}  // namespace

extern "C" int og_attention_delta(const float* dout, const float* out, int64_t rows, int32_t num_heads, int32_t dh, float* delta, void* stream) {
    og_clear_status();
    if (!dout || !out || !delta || rows <= 0 || num_heads <= 0 || dh <= 0) return OG_E_INVALID;
    if ((dh & 3) || ((uintptr_t)dout & 15) || ((uintptr_t)out & 15)) return OG_E_ALIGN;
    const int64_t n = rows * num_heads;
    if ((n + 255) / 256 > 0x7fffffffLL) return OG_E_SHAPE;
    hipLaunchKernelGGL(attention_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, out, rows, num_heads, dh,
                       delta);
    return og_launch_status();
}

extern "C" int og_attention_train_lse(const float* q, const float* k, int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh,
                                      float scale, float* lse, void* stream) {
    og_clear_status();
    AttnTrainArgs a{};
    a.q = q; a.k = k; a.lse_out = lse;
    a.B = batch; a.nq = nq; a.nk = nk; a.H = num_heads; a.D = num_heads * dh; a.scale = scale;
    a.ldq = a.ldk = a.D;
    if (int rc = check(a, dh)) return rc;
    if (!lse) return OG_E_INVALID;
    const int64_t blocks = (int64_t)batch * num_heads * ((nq + BR - 1) / BR);
    if (blocks > 0x7fffffffLL) return OG_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dh == 64) hipLaunchKernelGGL(attention_lse_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (dh == 32) hipLaunchKernelGGL(attention_lse_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attention_lse_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return og_launch_status();
}

extern "C" int og_attention_backward_parts(int32_t nk) { return nk > 0 ? (nk + BC - 1) / BC : 0; }

extern "C" int og_attention_backward_ld(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                                        const float* lse, const float* delta, int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh,
                                        float scale, float* dq_part, float* dk, int64_t lddk, float* dv, int64_t lddv, void* stream) {
    og_clear_status();
    AttnTrainArgs a{};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.delta = delta;
    a.dq_part = dq_part; a.dk = dk; a.dv = dv;
    a.B = batch; a.nq = nq; a.nk = nk; a.H = num_heads; a.D = num_heads * dh; a.scale = scale;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddk = lddk; a.lddv = lddv;
    if (int rc = check(a, dh)) return rc;
    if (!v || !dout || !lse || !delta || !dq_part || !dk || !dv) return OG_E_INVALID;
    if (((uintptr_t)v & 15) || ((uintptr_t)dout & 15) || (ldv & 3) || ldv < a.D || lddk < a.D || lddv < a.D) return OG_E_ALIGN;
    const int64_t blocks = (int64_t)batch * num_heads * ((nk + BC - 1) / BC);
    if (blocks > 0x7fffffffLL) return OG_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dh == 64) hipLaunchKernelGGL(attention_bwd_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (dh == 32) hipLaunchKernelGGL(attention_bwd_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attention_bwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return og_launch_status();
}

extern "C" int og_attention_backward(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                                     const float* delta, int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh,
                                     float scale, float* dq_part, float* dk, float* dv, void* stream) {
    const int64_t D = (int64_t)num_heads * dh;
    return og_attention_backward_ld(q, D, k, D, v, D, dout, lse, delta, batch, nq, nk, num_heads, dh, scale, dq_part, dk, D, dv, D, stream);
}

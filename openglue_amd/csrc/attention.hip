// Flash-style multi-head softmax attention for keypoint sets on gfx950 matrix cores.
//
// Replaces MultiheadAttention's head split + softmax_attention (reference attention_gnn.py:24-29,
// attention.py:8-19): per head, O = softmax_keys(Q Kᵀ / sqrt(d)) V.  The reference materialises the
// [B,H,N,N] attention matrix; here a workgroup owns 128 queries of one (problem, head), streams
// 64-key tiles of K and V through LDS and keeps the running max / sum / output in registers.
//
// Precision (DESIGN.md "attention numerics"): the parity bar is 1e-3 on the final log-scores and
// plain f16 operands miss it on small problems (logit error dominates).  Q, K and V are therefore
// split x = hi + lo with hi, lo both f16 (og_common.h: lo at its true scale, subnormals are honoured by the
// matrix cores); QKᵀ = Qh·Kh + Qh·Kl + Ql·Kh and PV = Ph·Vh + Ph·Vl + Pl·Vh (3 MFMAs each into ONE f32
// accumulator): every product carries ~22 mantissa bits.  (Rounding P alone to f16 already costs 1.2e-3
// on the `flags` golden case.)
//
// MFMA bookkeeping (v_mfma_f32_32x32x16_f16, lane l holds 8 k-values of row/col l&31, k-group l>>5):
//   Sᵀ[key][query] = K · Qᵀ   (A = K tile, B = Qᵀ)  -> lane owns ONE query column (l&31) and 16 keys
//                               per 32-key block: softmax statistics are lane-local + one xor-32 exchange.
//   Oᵀ[dv][query]  = Vᵀ · Pᵀ   (A = Vᵀ, B = Pᵀ)    -> same query column per lane, so the exp'd Sᵀ
//                               registers ARE the B operand: register 8t+e of key block kb is key
//                               kb*32 + 16t + 8(e>>2) + 4(l>>5) + (e&3); Vᵀ is staged in LDS as [dv][key]
//                               so those keys are two 8-byte reads.  (The k order inside an MFMA is free
//                               as long as A and B agree.)
// Q is expected PRE-SCALED by d^-1/2 * log2(e) (folded into the packed q-projection weights): the
// softmax then runs in the base-2 domain, p = 2^(s - m) is one v_exp_f32 with no extra multiply.
// The running max is only advanced (and O, l rescaled) when some row's max grew by more than 2^11
// in the current tile ("deferred rescale"): p stays <= 2^11, exactly representable in the f16 hi plane,
// and the 64-register rescale of O leaves the common path.
// I/O: q, k, v arrive as f16 (hi, lo) PLANES written by the producing GEMM's epilogue and O leaves as planes
// (it is the A operand of the fc.0 GEMM), so no conversion sits on the load path of either kernel.
#include "og_common.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int Q_TILE = 128;
constexpr float RESCALE_THR = 11.f;          // base-2 exponent headroom before the running max is advanced

template <int DH>
__global__ __launch_bounds__(256, 2) void attention_kernel(AttnArgs a, RaggedDesc rd) {
    constexpr int DHP = DH < 32 ? 32 : DH;        // Vᵀ rows padded to a full 32-row MFMA tile
    constexpr int NDV = DHP / 32;                 // output row blocks
    constexpr int NCH = DH / 16;                  // 16-wide k chunks of the QKᵀ contraction
    constexpr int KW = DH + 8;                    // K LDS row (halves): 16 B pad -> conflict-free b128
    constexpr int VW = KV_TILE + 4;               // Vᵀ LDS row (halves): 8 B pad -> conflict-free b64
    constexpr int D4 = DH / 4;                    // 4-wide dv groups per row
    constexpr int KSZ = KV_TILE * KW;             // halves per K plane buffer
    constexpr int VSZ = DHP * VW;                 // halves per Vᵀ plane buffer

    // Double-buffered tiles: K_t / V_t live in buffer t&1.  One array (a second __shared__ object makes
    // hipcc serialise LDS traffic with outstanding global loads).
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (2 * KSZ + 2 * VSZ)];
    auto Kh = [&](int b) { return smem + b * (2 * KSZ + 2 * VSZ); };
    auto Kl = [&](int b) { return smem + b * (2 * KSZ + 2 * VSZ) + KSZ; };
    auto Vh = [&](int b) { return smem + b * (2 * KSZ + 2 * VSZ) + 2 * KSZ; };
    auto Vl = [&](int b) { return smem + b * (2 * KSZ + 2 * VSZ) + 2 * KSZ + VSZ; };

    // XCD-aware block mapping: workgroup b is dispatched to XCD b % 8 and every XCD has its own L2.  All
    // query tiles of one (problem, head) share the same K/V (512 KB at 1024 keys), so they are placed on ONE
    // XCD; with the natural (x = tile) order the eight tiles land on eight XCDs and K/V is fetched from
    // HBM / Infinity Cache eight times (measured: 743 MB per launch instead of ~250 MB).
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int grp = (local / a.qtiles) * 8 + xcd;          // (problem, head) group
    if (grp >= a.nz * a.num_heads) return;
    const int z = grp / a.num_heads, h = grp - z * a.num_heads;
    const int gsel = z < a.split ? 0 : 1;
    const int zz = gsel ? z - a.split : z;
    int nq = a.nq[gsel], nk = a.nk[gsel];
    int64_t q_row0 = a.q_base[gsel] + (int64_t)zz * a.q_step[gsel];
    int64_t kv_row0 = a.kv_base[gsel] + (int64_t)zz * a.kv_step[gsel];
    if (rd.B > 0) {          // ragged batch: per-pair row ranges of the packed token matrix
        const int T0 = rd.off0[rd.B];
        const int b = z < rd.B ? z : z - rd.B;
        const int r0 = rd.off0[b], m_b = rd.off0[b + 1] - r0;
        const int r1 = T0 + rd.off1[b], n_b = rd.off1[b + 1] - rd.off1[b];
        const bool q_is0 = a.rag_mode == 1 ? z < rd.B : a.rag_mode == 2;
        const bool kv_is0 = a.rag_mode == 1 ? q_is0 : !q_is0;
        q_row0 = q_is0 ? r0 : r1; nq = q_is0 ? m_b : n_b;
        kv_row0 = kv_is0 ? r0 : r1; nk = kv_is0 ? m_b : n_b;
    }
    const int q0 = (local % a.qtiles) * Q_TILE;
    if (q0 >= nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    if (DH < 32) {   // zero the padding rows of Vᵀ once (both buffers); staging never touches them
        for (int i = tid; i < (DHP - DH) * VW; i += 256)
#pragma unroll
            for (int b = 0; b < 2; ++b) { Vh(b)[DH * VW + i] = (_Float16)0.f; Vl(b)[DH * VW + i] = (_Float16)0.f; }
    }

    // ---- Q fragments (B operand): lane (query l31, k-group hi) holds Q[q][16c + 8hi + e] ----
    f16x8 qh[NCH], ql[NCH];
    {
        int qi = q0 + wave * 32 + l31;
        if (qi >= nq) qi = nq - 1;     // clamp: computed but never stored
        const int64_t qo = (q_row0 + qi) * a.ldq + h * DH + 8 * hi;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            qh[c] = *reinterpret_cast<const f16x8*>(a.qh + qo + 16 * c);
            ql[c] = *reinterpret_cast<const f16x8*>(a.ql + qo + 16 * c);
        }
    }

    f32x16 oacc[NDV];
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = OG_NEG_INF, l_run = 0.f;

    // ---- staging: K as 16-byte chunks, V as 4 keys x 4 dv register transposes; a tile travels
    //      global -> registers (issued one tile ahead) -> LDS ----
    constexpr int C8 = DH / 8;                       // chunks per K row
    constexpr int K_KEYS_PER_PASS = 256 / C8;
    constexpr int K_PASSES = (KV_TILE + K_KEYS_PER_PASS - 1) / K_KEYS_PER_PASS;
    const int k_c8 = tid % C8, k_key = tid / C8;
    const int v_dg = tid % D4, v_kg = tid / D4;
    const bool v_active = v_kg < KV_TILE / 4;
    f16x8 rkh[K_PASSES], rkl[K_PASSES];
    f16x4 rvh[4], rvl[4];
    auto load_tile = [&](int kt) {
        const int key0 = kt * KV_TILE;
#pragma unroll
        for (int p = 0; p < K_PASSES; ++p) {
            const int key = k_key + p * K_KEYS_PER_PASS;
            if (key < KV_TILE) {
                int gk = key0 + key; if (gk >= nk) gk = nk - 1;      // clamp (masked in the softmax)
                const int64_t go = (kv_row0 + gk) * a.ldk + h * DH + 8 * k_c8;
                rkh[p] = *reinterpret_cast<const f16x8*>(a.kh + go);
                rkl[p] = *reinterpret_cast<const f16x8*>(a.kl + go);
            }
        }
        if (v_active) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int gk = key0 + 4 * v_kg + kk; if (gk >= nk) gk = nk - 1;
                const int64_t go = (kv_row0 + gk) * a.ldv + h * DH + 4 * v_dg;
                rvh[kk] = *reinterpret_cast<const f16x4*>(a.vh + go);
                rvl[kk] = *reinterpret_cast<const f16x4*>(a.vl + go);
            }
        }
    };
    auto store_tile = [&](int b) {
#pragma unroll
        for (int p = 0; p < K_PASSES; ++p) {
            const int key = k_key + p * K_KEYS_PER_PASS;
            if (key < KV_TILE) {
                *reinterpret_cast<f16x8*>(Kh(b) + key * KW + 8 * k_c8) = rkh[p];
                *reinterpret_cast<f16x8*>(Kl(b) + key * KW + 8 * k_c8) = rkl[p];
            }
        }
        if (v_active) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // dv = 4*v_dg + e : 4 consecutive keys
                f16x4 th, tl;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { th[kk] = rvh[kk][e]; tl[kk] = rvl[kk][e]; }
                *reinterpret_cast<f16x4*>(Vh(b) + (4 * v_dg + e) * VW + 4 * v_kg) = th;
                *reinterpret_cast<f16x4*>(Vl(b) + (4 * v_dg + e) * VW + 4 * v_kg) = tl;
            }
        }
    };
    // Oᵀ += Vᵀ_b Pᵀ  (P fragments of the tile staged in buffer b)
    f16x8 pf[2][2], pl[2][2];
    auto pv = [&](int b) {
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int off = (d * 32 + l31) * VW + kb * 32 + 16 * t + 4 * hi;
                    const f16x4 h0 = *reinterpret_cast<const f16x4*>(Vh(b) + off);
                    const f16x4 h1 = *reinterpret_cast<const f16x4*>(Vh(b) + off + 8);
                    const f16x4 l0 = *reinterpret_cast<const f16x4*>(Vl(b) + off);
                    const f16x4 l1 = *reinterpret_cast<const f16x4*>(Vl(b) + off + 8);
                    f16x8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vh[e] = h0[e]; vh[4 + e] = h1[e]; vl[e] = l0[e]; vl[4 + e] = l1[e]; }
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, pf[kb][t], oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kb][t], oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pf[kb][t], oacc[d], 0, 0, 0);
                }
    };

    const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    // Software pipeline per tile t (buffer t&1):  issue the global loads of tile t+1 | QKᵀ(t) on the matrix
    // pipe | PV(t-1) on the matrix pipe while the VALU runs softmax(t) | barrier | registers -> LDS | barrier
    for (int kt = 0; kt < ntiles; ++kt) {
        const int b = kt & 1;
        const int key0 = kt * KV_TILE;
        if (kt + 1 < ntiles) load_tile(kt + 1);

        // ---- Sᵀ = K Qᵀ for the two 32-key blocks ----
        float s[2][16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(Kh(b) + (kb * 32 + l31) * KW + 16 * c + 8 * hi);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(Kl(b) + (kb * 32 + l31) * KW + 16 * c + 8 * hi);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], sacc, 0, 0, 0);
            }
            if (key0 + KV_TILE > nk) {                  // only the last tile can hold padded keys (block-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + mfma32_row(r, lane);
                    s[kb][r] = key < nk ? sacc[r] : OG_NEG_INF;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = sacc[r];
            }
        }

        // ---- PV of the PREVIOUS tile: independent of Sᵀ(t), keeps the matrix pipe busy under the softmax ----
        if (kt > 0) pv(b ^ 1);

        // ---- online softmax over keys, base 2 (this lane: 32 of the tile's 64 keys of ONE query) ----
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);            // finite: every tile holds >= 1 valid key
        if (__any(m_new - m_run > RESCALE_THR)) {        // wave-uniform; first tile: inf > THR
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // rows that did not grow: 2^0 = 1
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;   // after PV(t-1): O is complete up to t-1
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_run);    // <= 2^RESCALE_THR
                psum += p;
                _Float16 th, tl;
                og_split(p, th, tl);
                pf[kb][r >> 3][r & 7] = th;
                pl[kb][r >> 3][r & 7] = tl;
            }
        l_run += psum;

        __syncthreads();                                  // every wave is done with K(t) and V(t-1)
        if (kt + 1 < ntiles) {
            store_tile(b ^ 1);                            // K(t+1), V(t+1) replace K(t-1), V(t-1)
            __syncthreads();
        }
    }
    pv((ntiles - 1) & 1);                                 // the last tile's PV

    // ---- normalise and store O[q][h*DH + dv] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qi = q0 + wave * 32 + l31;
    if (qi < nq) {
        const int64_t orow = (q_row0 + qi) * a.ldo;
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = d * 32 + 8 * g4 + 4 * hi;
                if (dv < DH) {
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float o = oacc[d][4 * g4 + e] * inv;
                        asm("" : "+v"(o));          // one materialised product for both halves of og_split (og_common.h)
                        _Float16 th, tl;
                        og_split(o, th, tl);
                        vh[e] = th; vl[e] = tl;
                    }
                    const int64_t oo = orow + (a.o_hl ? og_hl_col(h * DH + dv) : (int64_t)(h * DH + dv));
                    *reinterpret_cast<f16x4*>(a.oh + oo) = vh;
                    *reinterpret_cast<f16x4*>(a.ol + oo) = vl;
                }
            }
    }
}

}  // namespace

int og_launch_attention(const AttnArgs& a, hipStream_t stream) {
    if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.oh || !a.ol || a.nz <= 0 || a.num_heads <= 0) return OG_E_INVALID;
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 3)) return OG_E_ALIGN;
    const void* ps[8] = {a.qh, a.ql, a.kh, a.kl, a.vh, a.vl, a.oh, a.ol};
    for (int i = 0; i < 8; ++i)
        if ((uintptr_t)ps[i] & 15) return OG_E_ALIGN;
    int nqmax = 0;
    for (int g = 0; g < 2; ++g) {
        const bool used = g == 0 ? a.split > 0 : a.split < a.nz;
        if (!used) continue;
        if (a.nq[g] <= 0 || a.nk[g] <= 0) return OG_E_INVALID;
        if (a.nq[g] > nqmax) nqmax = a.nq[g];
    }
    AttnArgs a2 = a;
    a2.qtiles = (nqmax + Q_TILE - 1) / Q_TILE;
    RaggedDesc rd;
    rd.B = 0;
    if (a.rag) rd = *a.rag;
    a2.rag = nullptr;
    const int groups8 = (a.nz * a.num_heads + 7) / 8 * 8;
    dim3 grid(groups8 * a2.qtiles), block(256);
    switch (a.dh) {
        case 16: hipLaunchKernelGGL(attention_kernel<16>, grid, block, 0, stream, a2, rd); break;
        case 32: hipLaunchKernelGGL(attention_kernel<32>, grid, block, 0, stream, a2, rd); break;
        case 64: hipLaunchKernelGGL(attention_kernel<64>, grid, block, 0, stream, a2, rd); break;
        default: return OG_E_SHAPE;
    }
    return og_launch_status();
}

extern "C" int og_attention(const void* qh, const void* ql, int64_t ldq, const void* kh, const void* kl, int64_t ldk,
                            const void* vh, const void* vl, int64_t ldv, void* oh, void* ol, int64_t ldo, int32_t batch,
                            int32_t nq, int32_t nk, int32_t num_heads, int32_t dh, void* stream) {
    og_clear_status();
    AttnArgs a{};
    a.qh = (const _Float16*)qh; a.ql = (const _Float16*)ql; a.ldq = ldq;
    a.kh = (const _Float16*)kh; a.kl = (const _Float16*)kl; a.ldk = ldk;
    a.vh = (const _Float16*)vh; a.vl = (const _Float16*)vl; a.ldv = ldv;
    a.oh = (_Float16*)oh; a.ol = (_Float16*)ol; a.ldo = ldo; a.o_hl = 0;
    a.nz = batch; a.num_heads = num_heads; a.dh = dh; a.split = batch;
    a.q_base[0] = 0; a.q_step[0] = nq; a.kv_base[0] = 0; a.kv_step[0] = nk;
    a.nq[0] = nq; a.nk[0] = nk;
    a.q_base[1] = a.q_step[1] = a.kv_base[1] = a.kv_step[1] = 0; a.nq[1] = a.nk[1] = 0;
    return og_launch_attention(a, (hipStream_t)stream);
}

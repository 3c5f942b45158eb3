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
//                               kb*32 + 16t + 8(e>>2) + 4(l>>5) + (e&3); V is staged in LDS row-major [key][dv]
//                               like K and the A fragments are two ds_read_b64_tr_b16 (hardware 4x16 transpose)
//                               each.  (The k order inside an MFMA is free as long as A and B agree.)
// Q is expected PRE-SCALED by d^-1/2 * log2(e) (folded into the packed q-projection weights): the
// softmax then runs in the base-2 domain, p = 2^(s - m) is one v_exp_f32 with no extra multiply.
// The running max is only advanced (and O, l rescaled) when it has to ("deferred rescale"): the QKᵀ accumulator is initialised with
// -m_run, so in the common path the exponent arguments s - m_run come straight from the matrix pipe (no per-element subtract) and the
// rescale of O leaves the common path.  attention_kernel (register-staged, dh = 16) computes the tile maximum and advances when a row's
// max grew by more than 2^11; attention_dma_kernel (dh = 64 / 32, the hot one) does not even compute the maximum per tile: it reads the
// need off the row sum it forms anyway (a lane whose 32 exponentials add up to <= 2^15 holds no p above 2^15: inside f16 range).
// This file is compiled WITHOUT packed-fp32 instructions (openglue_amd/build.py: a v_pk_*_f32 does not issue while the other wave of
// the SIMD keeps the matrix pipe busy).  Launch forms of attention_dma_kernel: 4 waves / two workgroups per CU (batches); KS = 2, an
// 8-wave workgroup whose halves split the key range (<= one workgroup per CU); GS = 2 / 4, the key range of a query tile split over
// workgroups that meet in scratch (one or two pairs) -- og_launch_attention picks.
// I/O: q, k, v arrive as f16 (hi, lo) PLANES written by the producing GEMM's epilogue and O leaves as planes
// (it is the A operand of the fc.0 GEMM), so no conversion sits on the load path of either kernel.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "og_common.h"

namespace {

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4 og_lds_s16x4;
typedef __attribute__((address_space(3))) void og_lds_void;
typedef __attribute__((address_space(1))) const void og_glb_void;

constexpr int KV_TILE = 64;
constexpr int Q_TILE = 128;
constexpr float RESCALE_THR = 11.f;          // base-2 exponent headroom before the running max is advanced (register-staged kernel)
#ifndef OG_ATTN_SUMLIMIT
#define OG_ATTN_SUMLIMIT 32768.f
#endif
constexpr float SUM_LIMIT = OG_ATTN_SUMLIMIT;         // attention_dma_kernel: the running max moves when the 32 exponentials of a lane in one tile add up to more than 2^15

// Experiment builds only (scripts/build_ablation.sh attn_trace -DOG_ATTN_TRACE=1): per-segment shader-cycle stamps of
// all four waves of two workgroups, read back by og_debug_attn_trace().  The sched_barriers around the stamps
// perturb the schedule; compare the traced build's kernel time with the normal one before trusting a breakdown.
#ifndef OG_ATTN_PKSUM
#define OG_ATTN_PKSUM 0       // experiments: 1 = the row sum as packed-fp32 adds (rounds 1-3)
#endif
#ifndef OG_ATTN_MAXFIRST
#define OG_ATTN_MAXFIRST 0    // experiments: 1 = tile maximum in front of every tile's exponentials (rounds 1-3)
#endif
#ifndef OG_ATTN_TRACE
#define OG_ATTN_TRACE 0
#endif
#ifndef OG_ATTN_DBG
#define OG_ATTN_DBG 0      // debugging: 1 = P split in C++ instead of the 3-instruction asm
#endif
#if OG_ATTN_TRACE
__device__ unsigned og_attn_trace_buf[2][4][16][8];
#define OG_TP(i)                                                                                         \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (tsel >= 0) tp[i] = (unsigned)__builtin_amdgcn_s_memtime();   /* SMEM: costs an lgkmcnt(0) at the stamp */        \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define OG_TP(i) do {} while (0)
#endif


// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>); the index feeds asm immediates.
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// LDS fragment reads as inline asm with an immediate offset: the compiler's waitcnt insertion treats an LDS read it
// knows about as a possible alias of every LDS-DMA in flight and drains vmcnt(0) in front of it (which would serialise
// the next tile's DMA with this tile's PV); the waits for these reads are counted by hand (LDS returns in order).
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64(s16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int OFF>
__device__ __forceinline__ void lds_read_tr8_b64(i32x2& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

// pipelined loop: K fragment read order -- pieces 0, 1 = hi of key blocks 0, 1 (double-buffered by chunk), 2, 3 = lo (ONE buffer: the pass-0 MFMAs alone read it)
// -> read_k1's piece index (key block j >> 1, plane j & 1)
__device__ constexpr int og_korder(int q) { return q == 0 ? 0 : q == 1 ? 2 : q == 2 ? 1 : 3; }

template <int DH, class RD>
__global__ __launch_bounds__(256, DH > 64 ? 1 : 2) void attention_kernel(AttnArgs a, RD rd) {      // dh = 128 (round 6: two heads at 256-d): 136 KB of LDS, one workgroup per CU
    constexpr int DHP = DH < 32 ? 32 : DH;        // Vᵀ rows padded to a full 32-row MFMA tile
    constexpr int NDV = DHP / 32;                 // output row blocks
    constexpr int NCH = DH / 16;                  // 16-wide k chunks of the QKᵀ contraction
    constexpr int KW = DH + 8;                    // K LDS row (halves): 16 B pad -> conflict-free b128
    constexpr int VP = DHP + 8;                   // V LDS row (halves), row-major [key][dv] like K: conflict-free b128 stores and
                                                  // conflict-free ds_read_b64_tr_b16 gathers (4 rows x 32 B per 16-lane group)
    constexpr int KSZ = KV_TILE * KW;             // halves per K plane buffer
    constexpr int VSZ = KV_TILE * VP;             // halves per V plane buffer

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

    if constexpr (DH < 32) {   // zero the padding columns DH..31 of V once (both buffers); staging never touches them
        for (int i = tid; i < KV_TILE * (DHP - DH); i += 256) {
            const int o = (i / (DHP - DH)) * VP + DH + i % (DHP - DH);
#pragma unroll
            for (int b = 0; b < 2; ++b) { Vh(b)[o] = (_Float16)0.f; Vl(b)[o] = (_Float16)0.f; }
        }
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
    float m_run = 0.f, l_run = 0.f;              // m_run: any finite start; the first tile replaces it (kt == 0 below)

    // ---- staging: K and V as 16-byte chunks; a tile travels
    //      (V is staged row-major exactly like K and transposed by the LDS read, ds_read_b64_tr_b16)
    //      global -> registers (issued one tile ahead) -> LDS.  Addresses: one per-lane byte offset computed once,
    //      the tile base is wave-uniform (scalar) arithmetic; only the last, partial tile clamps its rows. ----
    constexpr int C8 = DH / 8;                       // chunks per K row
    constexpr int K_KEYS_PER_PASS = 256 / C8;
    constexpr int K_PASSES = (KV_TILE + K_KEYS_PER_PASS - 1) / K_KEYS_PER_PASS;
    const int k_c8 = tid % C8, k_key = tid / C8;
    f16x8 rkh[K_PASSES], rkl[K_PASSES], rvh[K_PASSES], rvl[K_PASSES];
    const unsigned k_off = (unsigned)(k_key * (int)a.ldk + 8 * k_c8) * 2u;                  // bytes inside a tile
    const unsigned v_off = (unsigned)(k_key * (int)a.ldv + 8 * k_c8) * 2u;
    const int64_t k_tile0 = (kv_row0 * a.ldk + h * DH) * 2, v_tile0 = (kv_row0 * a.ldv + h * DH) * 2;   // bytes, uniform
    auto load_tile = [&](int kt) {
        const int key0 = kt * KV_TILE;
        const char* kbh = reinterpret_cast<const char*>(a.kh) + k_tile0 + (int64_t)key0 * a.ldk * 2;
        const char* kbl = reinterpret_cast<const char*>(a.kl) + k_tile0 + (int64_t)key0 * a.ldk * 2;
        const char* vbh = reinterpret_cast<const char*>(a.vh) + v_tile0 + (int64_t)key0 * a.ldv * 2;
        const char* vbl = reinterpret_cast<const char*>(a.vl) + v_tile0 + (int64_t)key0 * a.ldv * 2;
        if (key0 + KV_TILE <= nk) {                  // full tile (block-uniform): no clamping
#pragma unroll
            for (int p = 0; p < K_PASSES; ++p) {
                if (k_key + p * K_KEYS_PER_PASS < KV_TILE) {
                    const unsigned o = k_off + (unsigned)(p * K_KEYS_PER_PASS * (int)a.ldk) * 2u;
                    rkh[p] = *reinterpret_cast<const f16x8*>(kbh + o);
                    rkl[p] = *reinterpret_cast<const f16x8*>(kbl + o);
                    const unsigned ov = v_off + (unsigned)(p * K_KEYS_PER_PASS * (int)a.ldv) * 2u;
                    rvh[p] = *reinterpret_cast<const f16x8*>(vbh + ov);
                    rvl[p] = *reinterpret_cast<const f16x8*>(vbl + ov);
                }
            }
        } else {                                     // last, partial tile: rows past nk are clamped (masked in the softmax)
#pragma unroll
            for (int p = 0; p < K_PASSES; ++p) {
                const int key = k_key + p * K_KEYS_PER_PASS;
                if (key < KV_TILE) {
                    int gk = key0 + key; if (gk >= nk) gk = nk - 1;
                    const int64_t go = (kv_row0 + gk) * a.ldk + h * DH + 8 * k_c8;
                    rkh[p] = *reinterpret_cast<const f16x8*>(a.kh + go);
                    rkl[p] = *reinterpret_cast<const f16x8*>(a.kl + go);
                    const int64_t gv = (kv_row0 + gk) * a.ldv + h * DH + 8 * k_c8;
                    rvh[p] = *reinterpret_cast<const f16x8*>(a.vh + gv);
                    rvl[p] = *reinterpret_cast<const f16x8*>(a.vl + gv);
                }
            }
        }
    };
    // LDS offsets of this lane (halves), identical for both buffers; the buffer is a compile-time constant at every use
    // (the tile loop is unrolled by two) so all LDS addresses are one VGPR + an immediate.
    const int ks_off = k_key * KW + 8 * k_c8;            // K staging store
    const int vs_off = k_key * VP + 8 * k_c8;            // V staging store
    const int kf_off = l31 * KW + 8 * hi;                // K fragment read (key block kb, chunk c: + kb*32*KW + 16c)
    // V fragment gather: each 16-lane group fetches one [4 keys][16 dv] block, lane i of the group supplies the address
    // of the 8-byte chunk (key i>>2, dv 4(i&3)..) and receives column dv i of the 4 keys (scripts/probes/ds_read_tr.hip).
    // Groups: lanes 0-15 dv 0-15, 16-31 dv 16-31 (k-half hi = 0), 32-63 the same for hi = 1 (keys + 4).
    const int vf_off = (4 * hi + ((lane & 15) >> 2)) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto store_tile = [&](auto BUF) {
        constexpr int b = decltype(BUF)::value;
#pragma unroll
        for (int p = 0; p < K_PASSES; ++p) {
            if (k_key + p * K_KEYS_PER_PASS < KV_TILE) {
                *reinterpret_cast<f16x8*>(Kh(b) + ks_off + p * K_KEYS_PER_PASS * KW) = rkh[p];
                *reinterpret_cast<f16x8*>(Kl(b) + ks_off + p * K_KEYS_PER_PASS * KW) = rkl[p];
            }
        }
#pragma unroll
        for (int p = 0; p < K_PASSES; ++p) {
            if (k_key + p * K_KEYS_PER_PASS < KV_TILE) {
                *reinterpret_cast<f16x8*>(Vh(b) + vs_off + p * K_KEYS_PER_PASS * VP) = rvh[p];
                *reinterpret_cast<f16x8*>(Vl(b) + vs_off + p * K_KEYS_PER_PASS * VP) = rvl[p];
            }
        }
    };
    // Oᵀ += Vᵀ_b Pᵀ  (P fragments of the tile staged in buffer b)
    f16x8 pf[2][2], pl[2][2];
    auto pv = [&](auto BUF) {
        constexpr int b = decltype(BUF)::value;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // A operand element e of lane (dv, hi): key kb*32 + 16t + 8(e>>2) + 4hi + (e&3) -> two transposing reads
                f16x8 vh[NDV], vl[NDV];
#pragma unroll
                for (int d = 0; d < NDV; ++d) {
                    const int off = vf_off + (kb * 32 + 16 * t) * VP + d * 32;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((og_lds_s16x4*)(Vh(b) + off));
                    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((og_lds_s16x4*)(Vh(b) + off + 8 * VP));
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((og_lds_s16x4*)(Vl(b) + off));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((og_lds_s16x4*)(Vl(b) + off + 8 * VP));
                    vh[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    vl[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                // pass-major over the dv blocks: consecutive MFMAs write different accumulators (when NDV = 2)
#pragma unroll
                for (int d = 0; d < NDV; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[d], pf[kb][t], oacc[d], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < NDV; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], pl[kb][t], oacc[d], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < NDV; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], pf[kb][t], oacc[d], 0, 0, 0);
            }
    };
    const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
#if OG_ATTN_TRACE
    const int tsel = blockIdx.x == 8 * 40 ? 0 : blockIdx.x == 8 * 41 + 3 ? 1 : -1;     // two workgroups somewhere in the middle
    unsigned tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    load_tile(0);
    store_tile(std::integral_constant<int, 0>{});
    __syncthreads();
    // Software pipeline per tile t (buffer t&1):  issue the global loads of tile t+1 | QKᵀ(t) on the matrix
    // pipe | PV(t-1) on the matrix pipe while the VALU runs softmax(t) | barrier | registers -> LDS | barrier
    auto tile_step = [&](int kt, auto BUF) {
        constexpr int b = decltype(BUF)::value;
        const int key0 = kt * KV_TILE;
        OG_TP(0);
        if (kt + 1 < ntiles) load_tile(kt + 1);
        OG_TP(1);

        // ---- S' = K Qᵀ - m_run for the two 32-key blocks: the accumulator starts at -m_run, so the exponent
        //      arguments of the common (no-rescale) path come straight out of the matrix pipe ----
        float s[2][16];
        {
            // the two 32-key blocks are independent accumulator chains: interleaved, so consecutive MFMAs never depend
            // on each other
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = -m_run;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                f16x8 kh[2], kl[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    kh[kb] = *reinterpret_cast<const f16x8*>(Kh(b) + kf_off + kb * 32 * KW + 16 * c);
                    kl[kb] = *reinterpret_cast<const f16x8*>(Kl(b) + kf_off + kb * 32 * KW + 16 * c);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[kb], qh[c], sacc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kb], ql[c], sacc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kb], qh[c], sacc[kb], 0, 0, 0);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (key0 + KV_TILE > nk) {                  // only the last tile can hold padded keys (block-uniform)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kb * 32 + mfma32_row(r, lane);
                        s[kb][r] = key < nk ? sacc[kb][r] : OG_NEG_INF;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] = sacc[kb][r];
                }
            }
        }

        OG_TP(2);
        // ---- PV of the PREVIOUS tile: independent of S(t), keeps the matrix pipe busy under the softmax ----
        if (kt > 0) pv(std::integral_constant<int, b ^ 1>{});
        OG_TP(3);

        // ---- online softmax over keys, base 2 (this lane: 32 of the tile's 64 keys of ONE query) ----
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));          // max of s - m_run over the tile; finite: every tile holds >= 1 valid key
        if (kt == 0 || __any(mt > RESCALE_THR)) {        // wave-uniform; rare after the first tile
            const float delta = kt == 0 ? mt : fmaxf(mt, 0.f);       // new running max = m_run + delta
            m_run += delta;
            if (kt > 0) {                                // after PV(t-1): O and l are complete up to t-1 (first tile: both still 0)
                const float alpha = __builtin_amdgcn_exp2f(-delta);  // rows that did not grow: 2^0 = 1
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDV; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float p0 = __builtin_amdgcn_exp2f(s[kb][r]);       // <= 2^RESCALE_THR
                const float p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                const float p2 = __builtin_amdgcn_exp2f(s[kb][r + 2]);
                const float p3 = __builtin_amdgcn_exp2f(s[kb][r + 3]);
                psum += (p0 + p1) + (p2 + p3);
                unsigned ha, la, hb, lb;
#if OG_ATTN_DBG & 1
                auto pack = [](_Float16 x, _Float16 y) { return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16); };
                { _Float16 h0_, l0_, h1_, l1_; og_split(p0, h0_, l0_); og_split(p1, h1_, l1_); ha = pack(h0_, h1_); la = pack(l0_, l1_);
                  og_split(p2, h0_, l0_); og_split(p3, h1_, l1_); hb = pack(h0_, h1_); lb = pack(l0_, l1_); }
#else
                og_split4(p0, p1, p2, p3, ha, la, hb, lb);      // og_common.h: 3 instructions per pair, hazard-safe
#endif
                unsigned* pfw = reinterpret_cast<unsigned*>(&pf[kb][r >> 3]);
                unsigned* plw = reinterpret_cast<unsigned*>(&pl[kb][r >> 3]);
                pfw[(r & 7) >> 1] = ha; pfw[((r & 7) >> 1) + 1] = hb;
                plw[(r & 7) >> 1] = la; plw[((r & 7) >> 1) + 1] = lb;
            }
        l_run += psum;

        OG_TP(4);
        __syncthreads();                                  // every wave is done with K(t) and V(t-1)
        OG_TP(5);
        if (kt + 1 < ntiles) {
            store_tile(std::integral_constant<int, b ^ 1>{});   // K(t+1), V(t+1) replace K(t-1), V(t-1)
            OG_TP(6);
            __syncthreads();
        }
        OG_TP(7);
#if OG_ATTN_TRACE
        if (tsel >= 0 && lane == 0 && kt < 16)
#pragma unroll
            for (int i = 0; i < 8; ++i) og_attn_trace_buf[tsel][wave][kt][i] = tp[i];
#endif
    };
    for (int kt = 0; kt < ntiles; kt += 2) {
        tile_step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < ntiles) tile_step(kt + 1, std::integral_constant<int, 1>{});
    }
    if ((ntiles - 1) & 1) pv(std::integral_constant<int, 1>{});   // the last tile's PV
    else pv(std::integral_constant<int, 0>{});

    // ---- normalise and store O[q][h*DH + dv] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qi = q0 + wave * 32 + l31;
    // row log-sum-exp of the (scaled) scores in natural units, for a backward pass that recomputes P (uniform calls only):
    // the kernel works in base 2 (q carries log2 e): L = ln 2 (m_run + log2 l)
    if (a.lse && hi == 0 && qi < nq) a.lse[((int64_t)z * a.num_heads + h) * nq + qi] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
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


// ---------------------------------------------------------------------------------------------------------------
// dh = 64 (BASELINE configs 1-3, 5) and dh = 32 (config 4): a head row of K or V is exactly one 128-byte line (or half of one), so the
// tiles travel global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, 8 rows x 128 B per wave instruction) with no
// staging registers, no ds_write and no second barrier.  LDS rows are unpadded; bank conflicts are removed by an XOR
// of the 16-byte chunk index applied on the SOURCE address of the DMA (the LDS side of a DMA is always contiguous)
// and again on the fragment reads:
//   K (ds_read_b128, 16 lanes = 16 keys per pass, all reading the same logical chunk):  chunk ^ ((key >> 1) & 7)
//   V (ds_read_b64_tr_b16, a pass = [4 keys][32 dv] blocks):                             chunk ^ (((key >> 1) & 1) << 2)
// Per tile t (buffer t & 1):  issue the DMA of tile t+1 into the other buffer | QK^T(t) | softmax(t) | PV(t) |
// s_waitcnt vmcnt(0) + ONE barrier (tile t+1 landed, everybody is done with tile t).  The matrix pipe of a SIMD is kept
// busy by the second workgroup of the CU (two independent 4-wave workgroups, 64 KB of LDS each): inside one in-order
// wave a lagged PV(t-1) never overlapped softmax(t) anyway, the MFMA issue stalls the wave for the length of the burst.
// Measured alternative (git history): one 8-wave workgroup whose two halves alternate matrix and softmax
// phases between barriers, sharing the K/V ring -- 30 % slower: on this part the MFMA and VALU issue of the two waves of a
// SIMD add up (tile period ~ 2 x (1536 MFMA + ~1200 VALU cycles)) whatever the phase alignment.
// Timing experiment only (-DOG_ATTN_ABL16=1, results WRONG): every 32x32x16 MFMA of attention_dma_kernel issued as two 16x16x32 MFMAs on the first
// eight accumulator registers -- the same flops and matrix-pipe cycles in the form that costs less energy (scripts/probes/mfma_energy.hip).
#ifndef OG_ATTN_ABL16
#define OG_ATTN_ABL16 0
#endif
#ifndef OG_PIPE_ABL
#define OG_PIPE_ABL 0          // timing experiments on the pipelined loop (results WRONG): 1 = no DMA inside the steps, 2 = no barrier between the steps
#endif
__device__ __forceinline__ f32x16 og_attn_mfma(f16x8 a, f16x8 b, f32x16 c) {
#if OG_ATTN_ABL16
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ c0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), c1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// KS = 2 (few workgroups: one or a few image pairs, og_launch_attention): an 8-wave workgroup whose two halves take the two halves of the KEY
// range of the same 128 queries, each with its own K/V ring, and merge their (O, m, l) through LDS at the end -- a workgroup is alone on
// its CU then, and its latency is the number of key tiles a wave walks through (16 at 1024 keys: 26 us per launch at any small batch).
// GS = 2 / 4 (ONE or two pairs: fewer workgroups than a quarter / half of the CUs): the key tiles of a query tile are dealt to GS WORKGROUPS
// (same XCD); each writes its unnormalised (O, m, l) to scratch, and the workgroup that arrives last (one counter per query tile) merges the
// GS partial results in index order -- deterministic -- and runs the normal epilogue.
// MX = 1 (round 6): the two P V CROSS products on the block-scaled 8-bit matrix instruction.  The plane `vl` then holds, per key and head, the row
// [e4m3(V 2^-sv) x DH | e4m3((V - Vh) 2^(11 - sv)) x DH] (one byte per element: the same bytes as the f16 lo row) written by the projection epilogue;
// P8h = e4m3(P), P8l = e4m3((P - Ph) 2^11) are made next to the f16 hi parts.  Per 64-key tile and dv block, Ph.Vl + Pl.Vh = 2^(sv - 11) (P8h.V8lo + P8l.V8hi):
// two v_mfma_scale_f32_32x32x64_f8f6f4 (k = the lane's own 32 keys of the tile, constant E8M0 scales) instead of eight 32x32x16 f16 MFMAs.
// PIPE = 1 (round 6): the tile loop software-pipelined inside every wave -- one stream of 48 MFMAs per step, PV(t-1) then QK^T(t+1), with softmax(t)
// dealt out behind them in pieces of <= 7 vector instructions (comment at the loop).
template <int DH, class RD, int KS = 1, int GS = 1, int MX = 0, int PIPE = 0>
#ifndef OG_ATTN_WG32
#define OG_ATTN_WG32 2        // workgroups per CU the dh = 32 instantiation is compiled for (experiment: 3)
#endif
__global__ __launch_bounds__(PIPE == 2 ? 512 : 256 * KS, PIPE == 2 ? 2 : KS == 2 ? 1 : DH == 32 ? OG_ATTN_WG32 : 2) void attention_dma_kernel(AttnArgs a, RD rd) {      // (dh = 32 would fit three workgroups per CU: measured 4 % slower)
    static_assert(DH == 64 || DH == 32, "head rows of 128 or 64 bytes");
    constexpr int NDV = DH / 32, NCH = DH / 16;
    constexpr int ROWB = DH * 2;                    // bytes of a head row of one plane: a full 128-byte line (dh = 64) or half of one
    constexpr int RPI = 1024 / ROWB;                // rows per DMA instruction (8 or 16), LPR lanes per row
    constexpr int LPR = ROWB / 16;
    // PIPE = 2: the pipelined loop in an EIGHT-wave workgroup of 256 queries (one per CU: the two waves of a SIMD share the K / V tiles, so a tile's DMA and
    // its 32 instructions serve twice the queries); every wave fills RW = 8 rows of each plane instead of 16
    constexpr int NWAVES = PIPE == 2 ? 8 : 4;
    constexpr int QT = PIPE == 2 ? 2 * Q_TILE : Q_TILE;
    constexpr int RW = KV_TILE / NWAVES;
    static_assert(PIPE != 2 || DH == 64, "the eight-wave form needs RW >= the rows of one DMA instruction");
    constexpr int NPI = RW / RPI;                   // DMA pieces per wave and plane: the wave fills rows [RW w, RW w + RW)
    constexpr int PLANE = KV_TILE * ROWB;           // bytes: 64 keys x one head row
    constexpr int BUFB = 4 * PLANE;                 // Kh | Kl | Vh | Vl
    static_assert(KS == 1 || KS == 2, "key split");
    // phase form: two buffers of Kh | Kl | Vh | Vl; pipelined form: a K ring of two slots (Kh | Kl) and a V ring of THREE (Vh | Vl): 80 KB, two workgroups = the CU's 160 KB
    __shared__ __attribute__((aligned(1024))) char smem_all[PIPE ? 10 * PLANE : KS * 2 * BUFB];

    static_assert(GS == 1 || KS == 1, "the workgroup-level key split is built for 4-wave workgroups (dh = 64 and, round 5, dh = 32: the 128-d family's single pairs)");
    const int id = blockIdx.x;
    int grp, qt, part;
    if (GS > 1 && a.gs_scatter) {      // test knob (OG_ATTN_GS_SCATTER=1): the parts of a query tile are CONSECUTIVE workgroups, i.e. on different XCDs under the
        part = id % GS;                // round-robin dispatch -- the hand-over below has to be right wherever the parts run
        const int rest = id / GS;
        grp = ((rest >> 3) / a.qtiles) * 8 + (rest & 7);
        qt = (rest >> 3) % a.qtiles;
    } else {
        const int xcd = id & 7, local = id >> 3;
        grp = (local / (a.qtiles * GS)) * 8 + xcd;         // (problem, head) group: all its query tiles (and key parts) on one XCD -- for SPEED only
        const int qt_part = local % (a.qtiles * GS);
        qt = qt_part / GS; part = qt_part % GS;
    }
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
    const int q0 = qt * QT;
    if (q0 >= nq) return;
#ifndef OG_ATTN_PRIO
#define OG_ATTN_PRIO 0        // experiment (round 6): static wave priority 1 for every other workgroup -- bit (OG_ATTN_PRIO - 1) of its index inside the XCD --
#endif                        // so that the two workgroups sharing a CU do not arbitrate as equals (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#if OG_ATTN_PRIO
    if (((id >> 3) >> (OG_ATTN_PRIO - 1)) & 1) __builtin_amdgcn_s_setprio(1);
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = KS == 2 ? wave_all >> 2 : 0;        // which half of the key range
    const int wave = KS == 2 ? wave_all & 3 : wave_all;  // 32-query block inside the 128-query tile
    const int l31 = lane & 31, hi = lane >> 5;
    char* const smem = smem_all + half * 2 * BUFB;
    // key tiles of this half: [t_begin, t_begin + ntiles) of the problem's ceil(nk / 64); everything below sees only "its" keys
    const int nk_all = nk;
    int ntiles_other = 0;                                 // KS = 2: tiles of the other half (the halves must meet at the same barriers)
    if constexpr (GS > 1) {
        const int nt_all = (nk_all + KV_TILE - 1) / KV_TILE, per = (nt_all + GS - 1) / GS;
        const int t_begin = part * per < nt_all ? part * per : nt_all, t_end = t_begin + per < nt_all ? t_begin + per : nt_all;
        kv_row0 += (int64_t)t_begin * KV_TILE;
        const int k_end = t_end * KV_TILE < nk_all ? t_end * KV_TILE : nk_all;
        nk = k_end - t_begin * KV_TILE;                   // 0: a part past the last tile
        if (nk < 0) nk = 0;
    }
    if constexpr (KS == 2) {
        const int nt_all = (nk_all + KV_TILE - 1) / KV_TILE, t_half = (nt_all + 1) / 2;
        const int t_begin = half ? t_half : 0, t_end = half ? nt_all : t_half;
        ntiles_other = half ? t_half : nt_all - t_half;
        kv_row0 += (int64_t)t_begin * KV_TILE;
        const int k_end = t_end * KV_TILE < nk_all ? t_end * KV_TILE : nk_all;
        nk = k_end - t_begin * KV_TILE;                   // <= 0: the second half of a one-tile problem
        if (nk < 0) nk = 0;
    }

    // ---- DMA pieces: wave w fills rows [16w, 16w+16) of each of the four planes, NPI pieces of RPI rows each.  Chunk swizzles
    //      (on the LDS row r): dh = 64: K chunk ^ ((r >> 1) & 7), V chunk ^ (((r >> 1) & 1) << 2); dh = 32 (64-byte rows, four
    //      chunks): K chunk ^ ((r >> 2) & 3), V none (a [4 keys][32 dv] transposing pass already covers all banks) ----
    const int rl = lane / LPR, pc = lane % LPR;
    const int ldkb = (int)a.ldk * 2, ldvb = (int)a.ldv * 2;          // row strides in bytes
    [[maybe_unused]] const unsigned lds0_dma = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    [[maybe_unused]] __attribute__((address_space(3))) char* const smem_lds = (__attribute__((address_space(3))) char*)smem;
    unsigned ksw[2], vsw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * RPI + rl + (PIPE == 2 ? RW * wave : 0);     // + 16 w: a multiple of 16, invisible to either swizzle (PIPE = 2: + 8 w, visible to K's)
        ksw[i] = (unsigned)(pc ^ (DH == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3))) * 16u;
    }
    vsw = (unsigned)(pc ^ (DH == 64 ? (((rl >> 1) & 1) << 2) : 0)) * 16u;
    // MX: the 8-bit plane is read by ds_read_b64_tr_b8 -- 32 lanes = 8 key rows {0..3, 8..11} (+ 4 hi) x 32 bytes -- so its chunk swizzle is
    // dh = 64: chunk ^ 2 (((r >> 1) & 1) | (((r >> 3) & 1) << 1)), dh = 32 (64-byte rows): chunk ^ (((r >> 3) & 1) << 1)
    [[maybe_unused]] unsigned vsw8[2] = {0u, 0u};
    if constexpr (MX) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * RPI + rl;
            vsw8[i] = (unsigned)(pc ^ (DH == 64 ? ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1) : (((r >> 3) & 1) << 1))) * 16u;
        }
    }
    const int64_t k_tile0 = (kv_row0 * a.ldk + h * DH) * 2, v_tile0 = (kv_row0 * a.ldv + h * DH) * 2;   // bytes, uniform
    // one tile = 8 DMA instructions per wave, issued in pairs (plane pair pp: 0 = K hi/lo, 1 = V hi/lo of piece i) so that
    // the main loop can spread them under its MFMA bursts: the vector-memory path takes 64 B/clk per CU, a burst of 8 per
    // wave right after the barrier stalls every wave of the workgroup for ~700 cycles (scripts/trace_attention.py)
#ifndef OG_ATTN_ASMDMA
#define OG_ATTN_ASMDMA 0      // 1 (experiment, round 3): scalar plane base + 32-bit lane offset in one asm block per (hi, lo) pair -- measured no gain (252 vs 249 us)
#endif
    // (scalar plane base + 32-bit lane offset) addressing, one asm block per (hi, lo) pair: the lane part is loop invariant except in
    // the last tile (row clamp); as builtin calls every piece carried 64-bit per-lane address arithmetic on the vector ALU -- which
    // shares its issue with the matrix pipe.
    unsigned koffs[NPI], voffs[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int r = wave * RW + i * RPI + rl;
        koffs[i] = (unsigned)(r * ldkb) + ksw[i];
        voffs[i] = (unsigned)(r * ldvb) + vsw;
        asm volatile("" : "+v"(koffs[i]), "+v"(voffs[i]));
    }
    static_assert(!MX || !OG_ATTN_ASMDMA, "the MX form uses the builtin DMA path");
    [[maybe_unused]] auto sptr = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    auto issue_pair = [&](int kt, auto BUF, auto I, auto PP) {
        constexpr int b = decltype(BUF)::value, i = decltype(I)::value, pp = decltype(PP)::value;
        const int key0 = kt * KV_TILE;
        const int last = nk - 1 - key0;                  // rows past the last key are clamped (masked in the softmax)
#if OG_ATTN_ASMDMA
        unsigned off = pp == 0 ? koffs[i] : voffs[i];
        if (last < KV_TILE - 1) {                        // block-uniform: only the last tile of a problem
            int r = wave * 16 + i * RPI + rl;
            r = r < last ? r : last;
            off = pp == 0 ? (unsigned)(r * ldkb) + ksw[i] : (unsigned)(r * ldvb) + vsw;
        }
        const int64_t o = pp == 0 ? k_tile0 + (int64_t)key0 * ldkb : v_tile0 + (int64_t)key0 * ldvb;       // uniform
        const char* bh = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kh : a.vh) + o);
        const char* bl = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kl : a.vl) + o);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0_dma + b * BUFB + (wave * 16 + i * RPI) * ROWB + pp * 2 * PLANE);
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "s_add_u32 m0, m0, %4\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2"
                     :: "v"(off), "s"(bh), "s"(bl), "s"(m0v), "n"(PLANE) : "memory");
#else
        int r = wave * 16 + i * RPI + rl;
        r = r < last ? r : last;
        // (an LDS-typed base cast once at kernel entry: a generic -> LDS cast at a call site the optimiser cannot see through emits an illegal
        //  V_CMP against src_shared_base on this compiler)
        __attribute__((address_space(3))) char* const dst = smem_lds + b * BUFB + (wave * 16 + i * RPI) * ROWB + pp * 2 * PLANE;
        if constexpr (pp == 0) {
            const int64_t o = k_tile0 + (int64_t)key0 * ldkb + (unsigned)(r * ldkb) + ksw[i];
            __builtin_amdgcn_global_load_lds((og_glb_void*)(reinterpret_cast<const char*>(a.kh) + o), (og_lds_void*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((og_glb_void*)(reinterpret_cast<const char*>(a.kl) + o), (og_lds_void*)(dst + PLANE), 16, 0, 0);
        } else {
            const int64_t o = v_tile0 + (int64_t)key0 * ldvb + (unsigned)(r * ldvb) + vsw;
            __builtin_amdgcn_global_load_lds((og_glb_void*)(reinterpret_cast<const char*>(a.vh) + o), (og_lds_void*)(dst), 16, 0, 0);
            if constexpr (MX) {
                const int64_t o8 = v_tile0 + (int64_t)key0 * ldvb + (unsigned)(r * ldvb) + vsw8[i & 1];
                __builtin_amdgcn_global_load_lds((og_glb_void*)(reinterpret_cast<const char*>(a.vl) + o8), (og_lds_void*)(dst + PLANE), 16, 0, 0);
            } else
            __builtin_amdgcn_global_load_lds((og_glb_void*)(reinterpret_cast<const char*>(a.vl) + o), (og_lds_void*)(dst + PLANE), 16, 0, 0);
        }
#endif
    };
    auto issue_tile = [&](int kt, auto BUF) {
        static_for<2 * NPI>([&](auto J) {
            constexpr int j = decltype(J)::value;
            issue_pair(kt, BUF, std::integral_constant<int, (j >> 1)>{}, std::integral_constant<int, (j & 1)>{});
        });
    };
    const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
    // pipelined form: one (hi, lo) pair of DMA instructions of tile kt, piece i, into the ring slot at byte offset slot_off (run-time: K slot (kt & 1) * 2 PLANE,
    // V slot 4 PLANE + (kt % 3) * 2 PLANE)
    [[maybe_unused]] auto pipe_dma = [&](int kt, auto I, auto PP, int slot_off) {
        // scalar plane bases + ONE 32-bit lane offset (koffs / voffs, four registers in all), both instructions in one asm block: with per-lane 64-bit addresses
        // the compiler kept ~10 lane constants for them, spilled some around the loop, and every reload's vmcnt(0) drained the DMA queue (the guide's pitfall)
        constexpr int i = decltype(I)::value, pp = decltype(PP)::value;
        const int key0 = kt * KV_TILE;
        const int last = nk - 1 - key0;
        unsigned off = pp == 0 ? koffs[i] : voffs[i];
        if (last < KV_TILE - 1) {                        // wave-uniform: only the last tile of a problem clamps its rows
            int r = wave * RW + i * RPI + rl;
            r = r < last ? r : last;
            off = pp == 0 ? (unsigned)(r * ldkb) + ksw[i] : (unsigned)(r * ldvb) + vsw;
        }
        const int64_t o = pp == 0 ? k_tile0 + (int64_t)key0 * ldkb : v_tile0 + (int64_t)key0 * ldvb;       // uniform
        const char* bh = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kh : a.vh) + o);
        const char* bl = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kl : a.vl) + o);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0_dma + (unsigned)slot_off + (unsigned)((wave * RW + i * RPI) * ROWB));
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "s_add_u32 m0, m0, %4\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2"
                     :: "v"(off), "s"(bh), "s"(bl), "s"(m0v), "n"(PLANE) : "memory");
    };
    if constexpr (PIPE) {        // the pipelined loop computes QK^T one tile ahead and reads V one step after its barrier: K(0), V(0), K(1), V(1) up front
        static_assert(KS == 1 && GS == 1 && MX == 0, "the pipelined loop is built for the batch form");
        static_assert(NPI >= 1, "rows per wave");
        static_for<NPI>([&](auto I) {
            pipe_dma(0, I, std::integral_constant<int, 0>{}, 0);
            pipe_dma(0, I, std::integral_constant<int, 1>{}, 4 * PLANE);
            if (ntiles > 1) {
                pipe_dma(1, I, std::integral_constant<int, 0>{}, 2 * PLANE);
                pipe_dma(1, I, std::integral_constant<int, 1>{}, 6 * PLANE);
            }
        });
    } else if (ntiles > 0) issue_tile(0, std::integral_constant<int, 0>{});

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
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm;                                   // -m_run in every element: the C operand that opens each QK^T chain
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    // fragment addresses of this lane (bytes inside a plane); everything else is an immediate
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned kf[NCH], va[NDV];
#pragma unroll
    for (int c = 0; c < NCH; ++c) kf[c] = lds0 + l31 * ROWB + (((2 * c + hi) ^ (DH == 64 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3))) * 16);
    {
        const int vrow = (4 * hi + ((lane & 15) >> 2)) * ROWB + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
#pragma unroll
        for (int d = 0; d < NDV; ++d) va[d] = lds0 + vrow + (DH == 64 ? 64 * (d ^ ((lane >> 3) & 1)) : 0);
    }
    // MX: A operands of the 8-bit products.  One ds_read_b64_tr_b8 gives lane (dv, hi) the bytes of 8 keys {0..3, 8..11} + 4 hi + 16 n (+ 32 for n >= 2):
    // lane i of a 16-lane group supplies the 8-byte chunk (key row i >> 1, dv 8 (i & 1) ..) and receives column i (scripts/probes/mx_pv.hip).
    // va8[part][d]: part 0 = the hi bytes (chunks 0 ..), 1 = the lo bytes (chunks DH / 16 ..) of dv block d; the read index n is an immediate.
    [[maybe_unused]] unsigned va8[2][NDV];
    if constexpr (MX) {
        const int i16 = lane & 15, ri = i16 >> 1, gi = (lane >> 4) & 1;
        const int row = 4 * hi + (ri & 3) + 8 * (ri >> 2);
        const int f = DH == 64 ? ((((ri >> 1) & 1) | ((ri >> 2) << 1)) << 1) : ((ri >> 2) << 1);
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int d = 0; d < NDV; ++d) va8[part][d] = lds0 + row * ROWB + ((((part * DH + d * 32) / 16) ^ f) + gi) * 16 + 8 * (i16 & 1);
    }
    // fragment registers, double-buffered by hand: K [buf][key block], V [buf][dv block] as two transposed halves
    f16x8 kh[2][2], kl[2][2];
    s16x4 vh0[2][NDV], vh1[2][NDV], vl0[2][MX ? 1 : NDV], vl1[2][MX ? 1 : NDV];
    [[maybe_unused]] i32x2 v8[2][4];                     // MX: [buffer][read n]: 8 key bytes each
    [[maybe_unused]] const float mx_rscale = 0x1p-11f;
    [[maybe_unused]] int mx_sa = a.mx_scale_a;            // E8M0 bytes of the A-side block scale: 127 + sv - 11 (byte 0)
    asm volatile("" : "+v"(mx_sa));

#if OG_ATTN_TRACE
    const int tsel = blockIdx.x == 8 * 40 ? 0 : blockIdx.x == 8 * 41 + 3 ? 1 : -1;     // two workgroups somewhere in the middle
    unsigned tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // MASKED: this tile may hold padded keys (only the last tile of a problem can).  A compile-time flag: with the mask as a
    // run-time branch the two sides of it merged the 32 score registers through register copies in EVERY tile.
    auto tile_step = [&](int kt, auto BUF, auto MASKED) {
        constexpr int b = decltype(BUF)::value;
        constexpr bool masked = decltype(MASKED)::value;
        const int key0 = kt * KV_TILE;
        OG_TP(0);
        const bool more = kt + 1 < ntiles;
        OG_TP(1);

        // One LDS read per call, so that the reads of the NEXT chunk / group are slotted between the MFMAs of the current one:
        // a wave issues at most one ds_read_b128 per ~26 cycles / one ds_read_b64_tr_b16 per ~14 (scripts/probes/
        // lds_read_patterns.hip); as a burst in front of the group's MFMAs they delay the matrix pipe by their whole issue time.
        auto read_k1 = [&](auto C, auto J) {             // piece j of chunk c: key block j >> 1, plane j & 1 -> buffer c & 1
            constexpr int c = decltype(C)::value, j = decltype(J)::value, kb = j >> 1;
            if constexpr ((j & 1) == 0) lds_read_b128<b * BUFB + kb * 32 * ROWB>(kh[c & 1][kb], kf[c]);
            else lds_read_b128<b * BUFB + PLANE + kb * 32 * ROWB>(kl[c & 1][kb], kf[c]);
        };
        auto read_v1 = [&](auto G, auto J) {             // piece j of group g = 2kb + t: dv block j >> 2, (plane, half) j & 3
            constexpr int g = decltype(G)::value, j = decltype(J)::value, d = j >> 2, off = b * BUFB + 2 * PLANE + g * 16 * ROWB;
            if constexpr ((j & 3) == 0) lds_read_tr16_b64<off>(vh0[g & 1][d], va[d]);
            else if constexpr ((j & 3) == 1) lds_read_tr16_b64<off + 8 * ROWB>(vh1[g & 1][d], va[d]);
            else if constexpr ((j & 3) == 2) lds_read_tr16_b64<off + PLANE>(vl0[g & 1][d], va[d]);
            else lds_read_tr16_b64<off + PLANE + 8 * ROWB>(vl1[g & 1][d], va[d]);
        };
        // MX: the 4 NDV reads a group g = (key block, half) needs: the f16 hi fragments (2 NDV) and its share of the 8-bit A operands -- NDV = 2: all four
        // reads of product g; NDV = 1: reads 2 (g & 1), + 1 of product g >> 1 (its MFMA follows the odd group)
        [[maybe_unused]] auto read_vm = [&](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value;
            if constexpr (j < 2 * NDV) {
                constexpr int d = j >> 1, off = b * BUFB + 2 * PLANE + g * 16 * ROWB;
                if constexpr ((j & 1) == 0) lds_read_tr16_b64<off>(vh0[g & 1][d], va[d]);
                else lds_read_tr16_b64<off + 8 * ROWB>(vh1[g & 1][d], va[d]);
            } else {
                constexpr int k = j - 2 * NDV, jp = NDV == 2 ? g : g >> 1, n = NDV == 2 ? k : 2 * (g & 1) + k;
                constexpr int d = jp % NDV, part = 1 - jp / NDV;
                constexpr int off = b * BUFB + 3 * PLANE + ((n >> 1) * 32 + 16 * (n & 1)) * ROWB;
                lds_read_tr8_b64<off>(v8[jp & 1][n], va8[part][d]);
            }
        };
        auto fence = [] { __builtin_amdgcn_sched_barrier(0); };

        // ---- S' = K Q^T - m_run for the two 32-key blocks (independent accumulator chains, interleaved) ----
        float s[2][16];
        {
            f32x16 sacc[2];
            fence();
            static_for<4>([&](auto J) { read_k1(std::integral_constant<int, 0>{}, J); });
            static_for<NCH>([&](auto C) {
                constexpr int c = decltype(C)::value;
                constexpr int cb = c & 1;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh[cb][0]), "+v"(kh[cb][1]), "+v"(kl[cb][0]), "+v"(kl[cb][1]) :: "memory");
                fence();
                static_for<6>([&](auto M) {
                    constexpr int m = decltype(M)::value, kb = m & 1, pass = m >> 1;
                    // the first MFMA of a chain takes -m_run (negm) as its C operand: no accumulator initialisation
                    if constexpr (pass == 0) sacc[kb] = og_attn_mfma(kl[cb][kb], qh[c], c == 0 ? negm : sacc[kb]);
                    else if constexpr (pass == 1) sacc[kb] = og_attn_mfma(kh[cb][kb], ql[c], sacc[kb]);
                    else sacc[kb] = og_attn_mfma(kh[cb][kb], qh[c], sacc[kb]);
                    fence();
                    if constexpr (m < 4) {
                        if constexpr (c + 1 < NCH) {
                            read_k1(std::integral_constant<int, c + 1>{}, std::integral_constant<int, m>{});
                        } else if constexpr (2 * m < 4 * NDV) {          // last chunk: the first V fragments, they fly under the softmax
                            if constexpr (MX) {
                                read_vm(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * m>{});
                                read_vm(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * m + 1>{});
                            } else {
                                read_v1(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * m>{});
                                read_v1(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * m + 1>{});
                            }
                        }
                        fence();
                    }
                    // the DMA of tile t+1 in the shadow of the matrix pipe (its buffer was last read before the barrier), STAGGERED:
                    // wave w issues its instructions (eight at dh = 64) during chunk w % NCH.  The vector-memory path serves ~18 cycles per 1 KB
                    // instruction per CU; when all waves of the workgroup issue at the same point each instruction queues behind
                    // the others' (~125 cycles each in the trace), one wave at a time pays only its own service time.
                    if constexpr (m == 4) {
                        if (more && wave % NCH == c) issue_tile(kt + 1, std::integral_constant<int, b ^ 1>{});
                        fence();
                    }
                });
            });
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (masked && key0 + KV_TILE > nk) {        // only the last tile can hold padded keys (block-uniform)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kb * 32 + mfma32_row(r, lane);
                        s[kb][r] = key < nk ? sacc[kb][r] : OG_NEG_INF;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] = sacc[kb][r];
                }
            }
        }

        OG_TP(2);
        // ---- online softmax over keys, base 2 (this lane: 32 of the tile's 64 keys of ONE query) ----
        // The scores arrive as s - m_run, so the common path is: 32 exponentials, their (hi, lo) split and the row sum -- nothing else.
        // Whether the running max has to move is read off the SUM the tile needs anyway (round 4; before: a 22-instruction max tree +
        // a cross-lane exchange in front of every tile's exponentials): a lane whose 32 exponentials add up to <= 2^15 holds no p above
        // 2^15 (inside binary16, so the split is exact).  Otherwise -- tile 0, or a row whose maximum grew by more than ~2^10 over
        // m_run -- the wave takes the slow path: tile maximum, new running max, rescale of l and O, and the exponentials again.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 pfw[2][2], plw[2][2];                       // packed (hi, hi) / (lo, lo) pairs: [key block][k-step of 16 keys]
        [[maybe_unused]] i32x8 p8h, p8l;                  // MX: e4m3 bytes of P and of (P - Ph) 2^11: dword 4 kb + (r >> 2) = this lane's keys 8 (r >> 2) + 4 hi + 0..3 of block kb
        float tsum = 0.f;
        // Row sum: FOUR plain v_add_f32 chains.  NOT v_pk_add_f32: on gfx950 a packed-fp32 instruction does not issue while the matrix
        // pipe of its SIMD is busy -- beside a saturated MFMA stream of the other wave every other VALU class still gets a slot every
        // ~14 cycles, packed fp32 gets none (scripts/probes/mfma_valu_classes.hip, profiles/r04_probe_mfma_valu_classes.log) -- so 16
        // packed adds per tile tied this wave's softmax to the gaps of the other wave's MFMA phases.
        auto exp_split = [&]() {
#if OG_ATTN_PKSUM
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 psum2 = {0.f, 0.f};
#else
            float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
            // a C add the compiler schedules itself (it knows the wait state between a transcendental and a VALU reading its result; an
            // inline-asm v_add_f32 right behind its v_exp_f32 read stale lanes), made opaque so that the SLP vectoriser cannot pair it
            auto add1 = [](float& acc, float x) { acc += x; asm("" : "+v"(acc)); };
#endif
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float p0 = __builtin_amdgcn_exp2f(s[kb][r]);
                    const float p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                    const float p2 = __builtin_amdgcn_exp2f(s[kb][r + 2]);
                    const float p3 = __builtin_amdgcn_exp2f(s[kb][r + 3]);
#if OG_ATTN_PKSUM
                    psum2 += f32x2{p0, p1};
                    psum2 += f32x2{p2, p3};
#else
                    add1(ps0, p0); add1(ps1, p1); add1(ps2, p2); add1(ps3, p3);
#endif
                    unsigned ha, la, hb, lb;
                    if constexpr (MX) {
                        og_split4_mx(p0, p1, p2, p3, mx_rscale, ha, hb, la, lb);      // la / lb: the e4m3 quadruples of P and of its residual
                        p8h[4 * kb + (r >> 2)] = (int)la; p8l[4 * kb + (r >> 2)] = (int)lb;
                    } else {
                        og_split4(p0, p1, p2, p3, ha, la, hb, lb);      // og_common.h: 3 instructions per pair, hazard-safe
                        plw[kb][r >> 3][(r & 7) >> 1] = la; plw[kb][r >> 3][((r & 7) >> 1) + 1] = lb;
                    }
                    pfw[kb][r >> 3][(r & 7) >> 1] = ha; pfw[kb][r >> 3][((r & 7) >> 1) + 1] = hb;
                }
#if OG_ATTN_PKSUM
            tsum = psum2[0] + psum2[1];
#else
            tsum = (ps0 + ps1) + (ps2 + ps3);
#endif
        };
        bool redo = kt == 0;                              // the first tile replaces the arbitrary start value of m_run
#if !OG_ATTN_MAXFIRST
        if (!redo) {
            exp_split();
            redo = __any(!(tsum <= (MX ? 256.f : SUM_LIMIT)));           // wave-uniform; catches inf and NaN too (MX: every p inside e4m3's 448)
        }
#endif
        float mt = 0.f;
        if (OG_ATTN_MAXFIRST || redo) {
            mt = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));      // max of s - m_run over the tile; finite: every tile holds >= 1 valid key
#if OG_ATTN_MAXFIRST
            redo = redo || __any(mt > RESCALE_THR);
#endif
        }
        if (redo) {
            const float delta = kt == 0 ? mt : fmaxf(mt, 0.f);       // new running max = m_run + delta
            m_run += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_run;
            if (kt > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);  // rows that did not grow: 2^0 = 1
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDV; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
        }
        if (OG_ATTN_MAXFIRST || redo) exp_split();
        l_run += tsum;
        f16x8 pf[2][2], pl[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) { pf[kb][t] = __builtin_bit_cast(f16x8, pfw[kb][t]); if constexpr (!MX) pl[kb][t] = __builtin_bit_cast(f16x8, plw[kb][t]); }

        OG_TP(3);
        // ---- O^T += V^T P^T of the same tile: group g = (key block kb, half t); A operand element e of lane (dv, hi) is
        //      key kb*32 + 16t + 8(e>>2) + 4hi + (e&3) -> two transposing reads per plane ----
        fence();
        if constexpr (MX) {
            static_for<4>([&](auto G) {
                constexpr int g = decltype(G)::value;
                constexpr int gb = g & 1, kb = g >> 1, t = g & 1;
                constexpr bool has8 = NDV == 2 || (g & 1);                      // an 8-bit product closes this group
                constexpr int jp = NDV == 2 ? g : g >> 1;
                if constexpr (NDV == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]), "+v"(vh0[gb][1]), "+v"(vh1[gb][1]),
                                 "+v"(v8[jp & 1][0]), "+v"(v8[jp & 1][1]), "+v"(v8[jp & 1][2]), "+v"(v8[jp & 1][3]) :: "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]),
                                 "+v"(v8[jp & 1][0]), "+v"(v8[jp & 1][1]), "+v"(v8[jp & 1][2]), "+v"(v8[jp & 1][3]) :: "memory");
                fence();
                f16x8 vh[NDV];
#pragma unroll
                for (int d = 0; d < NDV; ++d) vh[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vh0[gb][d], vh1[gb][d], 0, 1, 2, 3, 4, 5, 6, 7));
                constexpr int SLOTS = NDV + (has8 ? 1 : 0), RPS = (4 * NDV + SLOTS - 1) / SLOTS;      // the next group's reads, spread behind this group's MFMAs
                static_for<SLOTS>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    if constexpr (m < NDV) oacc[m] = og_attn_mfma(vh[m], pf[kb][t], oacc[m]);
                    else {
                        constexpr int d = jp % NDV, part = 1 - jp / NDV;            // part 1 = V8lo with P8h, part 0 = V8hi with P8l
                        const i32x8 a8 = __builtin_shufflevector(__builtin_shufflevector(v8[jp & 1][0], v8[jp & 1][1], 0, 1, 2, 3),
                                                                 __builtin_shufflevector(v8[jp & 1][2], v8[jp & 1][3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
                        oacc[d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, part ? p8h : p8l, oacc[d], 0, 0, 0, mx_sa, 0, 0x7F7F7F7F);
                    }
                    fence();
                    if constexpr (g + 1 < 4) {
                        static_for<RPS>([&](auto R) {
                            constexpr int j = m * RPS + decltype(R)::value;
                            if constexpr (j < 4 * NDV) read_vm(std::integral_constant<int, g + 1>{}, std::integral_constant<int, j>{});
                        });
                        fence();
                    }
                });
            });
        } else
        static_for<4>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int gb = g & 1, kb = g >> 1, t = g & 1;
            if constexpr (NDV == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]), "+v"(vl0[gb][0]), "+v"(vl1[gb][0]),
                             "+v"(vh0[gb][NDV - 1]), "+v"(vh1[gb][NDV - 1]), "+v"(vl0[gb][NDV - 1]), "+v"(vl1[gb][NDV - 1]) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]), "+v"(vl0[gb][0]), "+v"(vl1[gb][0]) :: "memory");
            fence();
            f16x8 vh[NDV], vl[NDV];
#pragma unroll
            for (int d = 0; d < NDV; ++d) {
                vh[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vh0[gb][d], vh1[gb][d], 0, 1, 2, 3, 4, 5, 6, 7));
                vl[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vl0[gb][d], vl1[gb][d], 0, 1, 2, 3, 4, 5, 6, 7));
            }
            static_for<3 * NDV>([&](auto M) {
                constexpr int m = decltype(M)::value, d = m % NDV, pass = m / NDV;
                if constexpr (pass == 0) oacc[d] = og_attn_mfma(vl[d], pf[kb][t], oacc[d]);
                else if constexpr (pass == 1) oacc[d] = og_attn_mfma(vh[d], pl[kb][t], oacc[d]);
                else oacc[d] = og_attn_mfma(vh[d], pf[kb][t], oacc[d]);
                fence();
                if constexpr (2 * m < 4 * NDV && g + 1 < 4) {                  // the next group's 4 NDV reads, two behind each MFMA
                    read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 2 * m>{});
                    read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 2 * m + 1>{});
                    fence();
                }
            });
        });

        OG_TP(4);
        if (kt + 1 < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of tile t+1 landed
            OG_TP(5);
            __syncthreads();                                      // ... everybody's did, and everybody is done with tile t
        }
        OG_TP(6);
#if OG_ATTN_TRACE
        tp[7] = tp[6];
        if (tsel >= 0 && lane == 0 && kt < 16)
#pragma unroll
            for (int i = 0; i < 8; ++i) og_attn_trace_buf[tsel][wave][kt][i] = tp[i];
#endif
    };

    // ---------------------------------------------------------------------------------------------------------------------------------------------
    // PIPE: the software-pipelined tile loop (round 6).  The phase structure above -- QK^T(t) | softmax(t) | PV(t) per wave -- leaves the matrix pipe of
    // a SIMD to whichever OTHER wave happens to be in a matrix phase; measured (scripts/probes/attn_pipe.hip, profiles/r06_b_probe_attn_pipe.log) the same
    // instruction mix costs 1932 cycles per SIMD and tile in phases and 1535 (= the 48 MFMAs' own issue time) when every wave runs ONE continuous MFMA
    // stream with the vector work of another tile dealt out behind the MFMAs.  Step t of a wave:
    //     MFMAs  0..23   O += V(t-1)^T P(t-1)^T      (4 groups of 16 keys; the V fragments of group g+1 are read behind the MFMAs of group g)
    //     MFMAs 24..47   S(t+1) = K(t+1) Q^T - m_run   (4 chunks of 16 channels; K fragments one chunk ahead)
    //     vector stream  softmax(t): 8 blocks of 4 scores = 4 v_exp, the (hi, lo) split, 4 row-sum adds; block 0 in front of the first MFMA (it covers
    //                    the latency of the first V reads), block k behind MFMAs 6 (k - 1) .. 6 (k - 1) + 4 in five pieces
    //     DMA            V(t) and K(t+2) into the OTHER buffer (its V was consumed by PV(t-2), its K by QK^T(t), both in step t-1), wave w's four
    //                    instruction pairs behind MFMAs 2 + 12 j + 3 w
    //     end of step    the running max (rare slow path: rescale O, l, S(t), S(t+1), redo P(t)), s_waitcnt vmcnt(0), ONE barrier
    // State in registers between steps: S(t) and P(t-1) in, S(t+1) and P(t) out -- two sets used alternately (steps AB and BA), so nothing is copied.
    // Buffer b = (t - 1) & 1 holds V(t-1) and K(t+1).  Prologue: QK^T(0), softmax(0) on the slow path (it sets m_run), QK^T(1).
    if constexpr (PIPE) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        f32x16 sA[2], sB[2];
        u32x4 pfA[2][2], plA[2][2], pfB[2][2], plB[2][2];
        auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
        // LDS reads from a ring slot: the slot's byte offset is folded into the address registers once per step (kfs, vas), everything else is an immediate
        // (moved IN PLACE -- the fragment address registers are this loop's alone -- so the ring costs no registers)
        int k_off_now = 0, v_off_now = 0;
        auto set_k_slot = [&](int off) {
            const unsigned d = (unsigned)(off - k_off_now);
            k_off_now = off;
#pragma unroll
            for (int c = 0; c < NCH; ++c) kf[c] += d;
        };
        auto set_v_slot = [&](int off) {
            const unsigned d = (unsigned)(off - v_off_now);
            v_off_now = off;
#pragma unroll
            for (int d2 = 0; d2 < NDV; ++d2) va[d2] += d;
        };
        auto read_k1 = [&](auto C, auto J) {
            constexpr int c = decltype(C)::value, j = decltype(J)::value, kb = j >> 1;
            if constexpr ((j & 1) == 0) lds_read_b128<kb * 32 * ROWB>(kh[c & 1][kb], kf[c]);
            else lds_read_b128<PLANE + kb * 32 * ROWB>(kl[0][kb], kf[c]);
        };
        auto read_v1 = [&](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value, d = j >> 2, off = g * 16 * ROWB;
            if constexpr ((j & 3) == 0) lds_read_tr16_b64<off>(vh0[g & 1][d], va[d]);
            else if constexpr ((j & 3) == 1) lds_read_tr16_b64<off + 8 * ROWB>(vh1[g & 1][d], va[d]);
            else if constexpr ((j & 3) == 2) lds_read_tr16_b64<off + PLANE>(vl0[0][d], va[d]);
            else lds_read_tr16_b64<off + PLANE + 8 * ROWB>(vl1[0][d], va[d]);
        };
        auto add1 = [](float& acc, float x) { acc += x; asm("" : "+v"(acc)); };
        // -m_run in all sixteen registers of a C operand, made right in front of the two MFMAs that open the QK^T chains (held for the whole step it is 16
        // registers the loop does not have; opaque, so that the compiler does not hoist the broadcast back out)
        auto make_negm = [&]() {
            float nm = -m_run;
            asm volatile("" : "+v"(nm));
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = nm;
            return v;
        };
        // the whole softmax of one tile in one go: prologue (tile 0) and the slow path
        auto exp_split_all = [&](f32x16 (&sc)[2], u32x4 (&pf)[2][2], u32x4 (&pl)[2][2]) -> float {
            float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float p0 = __builtin_amdgcn_exp2f(sc[kb][r]), p1 = __builtin_amdgcn_exp2f(sc[kb][r + 1]);
                    const float p2 = __builtin_amdgcn_exp2f(sc[kb][r + 2]), p3 = __builtin_amdgcn_exp2f(sc[kb][r + 3]);
                    add1(ps0, p0); add1(ps1, p1); add1(ps2, p2); add1(ps3, p3);
                    unsigned ha, la, hb, lb;
                    og_split4(p0, p1, p2, p3, ha, la, hb, lb);
                    pf[kb][r >> 3][(r & 7) >> 1] = ha; pf[kb][r >> 3][((r & 7) >> 1) + 1] = hb;
                    pl[kb][r >> 3][(r & 7) >> 1] = la; pl[kb][r >> 3][((r & 7) >> 1) + 1] = lb;
                }
            return (ps0 + ps1) + (ps2 + ps3);
        };
        auto row_max = [&](const f32x16 (&sc)[2]) -> float {
            float mt = sc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sc[kb][r]);
            // the other half of the keys sits 32 lanes away: v_permlane32_swap (no address register, no LDS round trip, unlike ds_bpermute)
            // (inline asm on two distinct registers: given the same value twice, the builtin's result lost its second half -- the fmaxf below was folded away)
            float a0 = mt, a1 = mt;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
            return fmaxf(a0, a1);
        };
        auto mask_tile = [&](f32x16 (&sc)[2], int key0) {          // padded keys of the last tile -> -inf
            int ln = lane;
            asm volatile("" : "+v"(ln));          // opaque: the 32 key indices are made HERE, in the rare branch (hoisted out of the loop they were spilled)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + mfma32_row(r, ln);
                    sc[kb][r] = key < nk ? sc[kb][r] : OG_NEG_INF;
                }
        };
        // S(kt) = K(kt) Q^T - m_run from buffer b, on its own (prologue only)
        auto qk_alone = [&](int kslot_off, f32x16 (&sn)[2], int kt) {
            set_k_slot(kslot_off);
            fence();
            static_for<4>([&](auto J) { read_k1(std::integral_constant<int, 0>{}, J); });
            const f32x16 negm = make_negm();
            static_for<NCH>([&](auto C) {
                constexpr int c = decltype(C)::value, cb = c & 1;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh[cb][0]), "+v"(kh[cb][1]), "+v"(kl[0][0]), "+v"(kl[0][1]) :: "memory");
                fence();
                static_for<6>([&](auto M) {
                    constexpr int m = decltype(M)::value, kb = m & 1, pass = m >> 1;
                    if constexpr (pass == 0) sn[kb] = og_attn_mfma(kl[0][kb], qh[c], c == 0 ? negm : sn[kb]);
                    else if constexpr (pass == 1) sn[kb] = og_attn_mfma(kh[cb][kb], ql[c], sn[kb]);
                    else sn[kb] = og_attn_mfma(kh[cb][kb], qh[c], sn[kb]);
                    fence();
                    if constexpr (m < 4 && c + 1 < NCH) { read_k1(std::integral_constant<int, c + 1>{}, std::integral_constant<int, og_korder(m)>{}); fence(); }
                });
            });
            if (kt * KV_TILE + KV_TILE > nk) mask_tile(sn, kt * KV_TILE);
        };
        // One step.  BUF: the buffer of V(t-1) and K(t+1); (sc, pf, pl) = S(t), P(t-1) in; (sn, nf, nl) = S(t+1), P(t) out.  The last two steps have no
        // QK^T (no tile t+1), the last one no softmax either: wave-uniform run-time branches inside ONE body per buffer parity -- eight compile-time variants
        // of a 48-MFMA body cost ~70 KB of code and register-allocation trouble at their joins (spills; a spilled LDS-read destination is stored BEFORE its
        // data arrive, so spills are not only slow here, they are wrong).  The last step's vector stream runs on dead score registers; its results are unused.
        // ring bookkeeping (wave-uniform): V(kt) lives in V slot kt % 3, K(kt) in K slot kt & 1
        int vslot_cur = 0;                                                      // slot of V(t-1) in step t; t starts at 1
        auto vslot_off = [](int sl) { return 4 * PLANE + sl * 2 * PLANE; };
        auto step = [&](int t, f32x16 (&sc)[2], u32x4 (&pfw)[2][2], u32x4 (&plw)[2][2], f32x16 (&sn)[2], u32x4 (&nfw)[2][2], u32x4 (&nlw)[2][2]) {
            const bool has_soft = t < ntiles, has_qk = t + 1 < ntiles;
            const bool dma_v = t + 1 < ntiles, dma_k = t + 2 < ntiles;         // V(t+1) -> V slot (t+1) % 3 (held V(t-2)), K(t+2) -> K slot t & 1 (held K(t)): both consumed in step t-1
            const int vslot_next = vslot_cur == 2 ? 0 : vslot_cur + 1;        // slot of V(t)
            const int vslot_dma = vslot_next == 2 ? 0 : vslot_next + 1;       // slot of V(t+1)
            const int kdma_off = (t & 1) * 2 * PLANE, vdma_off = vslot_off(vslot_dma);
            set_k_slot(((t + 1) & 1) * 2 * PLANE);                            // K(t+1); the V addresses of V(t-1) were set when its first fragments were requested
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
            // P(t) is written word by word: "define" the eight vectors first, or the words not written yet keep the previous contents alive (32 registers)
            asm volatile("" : "=v"(nfw[0][0]), "=v"(nfw[0][1]), "=v"(nfw[1][0]), "=v"(nfw[1][1]), "=v"(nlw[0][0]), "=v"(nlw[0][1]), "=v"(nlw[1][0]), "=v"(nlw[1][1]));
            // softmax(t), block blk = (key block blk >> 2, registers 4 (blk & 3) ..), piece ph
            auto soft = [&](auto BLK, auto PH) {
                constexpr int blk = decltype(BLK)::value, ph = decltype(PH)::value, kb = blk >> 2, r = 4 * (blk & 3);
                if constexpr (ph == 0) { p0 = __builtin_amdgcn_exp2f(sc[kb][r]); p1 = __builtin_amdgcn_exp2f(sc[kb][r + 1]); }
                else if constexpr (ph == 1) { p2 = __builtin_amdgcn_exp2f(sc[kb][r + 2]); p3 = __builtin_amdgcn_exp2f(sc[kb][r + 3]); }
                else if constexpr (ph == 2) {
                    unsigned ha, la, hb, lb;
                    og_split4(p0, p1, p2, p3, ha, la, hb, lb);
                    nfw[kb][r >> 3][(r & 7) >> 1] = ha; nfw[kb][r >> 3][((r & 7) >> 1) + 1] = hb;
                    nlw[kb][r >> 3][(r & 7) >> 1] = la; nlw[kb][r >> 3][((r & 7) >> 1) + 1] = lb;
                } else if constexpr (ph == 3) { add1(ps0, p0); add1(ps1, p1); }
                else if constexpr (ph == 4) { add1(ps2, p2); add1(ps3, p3); }
                fence();
            };
            // behind slot mm of the step: the softmax piece and, for the wave whose turn it is, one DMA pair
            auto behind = [&](auto MM) {
                constexpr int mm = decltype(MM)::value;
                if constexpr (mm % 6 < 5 && mm / 6 + 1 < 8) soft(std::integral_constant<int, mm / 6 + 1>{}, std::integral_constant<int, mm % 6>{});
                if constexpr (mm >= 2 && (mm - 2) % 3 == 0) {
                    // four waves: pair j = 0, 1 = K(t+2) pieces, 2, 3 = V(t+1) pieces of wave w; eight waves (one piece per plane pair): K of waves 0..7, then V
                    constexpr int k16 = (mm - 2) / 3;
                    constexpr int j = PIPE == 2 ? 2 * (k16 >> 3) : (mm - 2) / 12, w = PIPE == 2 ? (k16 & 7) : ((mm - 2) % 12) / 3;
                    constexpr int i = j & 1, pp = j >> 1;
                    if constexpr (i < NPI) {
#if !(OG_PIPE_ABL & 1)
                        if (wave == w && (pp == 0 ? dma_k : dma_v)) pipe_dma(pp == 0 ? t + 2 : t + 1, std::integral_constant<int, i>{}, std::integral_constant<int, pp>{}, pp == 0 ? kdma_off : vdma_off);
                        fence();
#endif
                    }
                }
            };
            // a step has 24 NDV MFMAs and 48 slots for the vector / DMA pieces: 2 / NDV slots behind every MFMA
            auto after_mfma = [&](auto K) {
                constexpr int k = decltype(K)::value, SPM = 2 / NDV;
                static_for<SPM>([&](auto I) { behind(std::integral_constant<int, k * SPM + decltype(I)::value>{}); });
            };
            fence();
            // (the first V fragments of tile t-1 were requested before the barrier that ended the previous step: V travels TWO steps ahead of its use through
            //  a ring of three slots, so it was visible to every wave a whole step ago -- nothing waits on a fresh LDS read behind the barrier)
            // ---- the running max moves BEFORE the exponentials (the phase form reads the need off the row sum afterwards and keeps S(t) alive for a second
            //      pass; here the 32 score registers must die block by block -- the step holds S(t+1), P(t-1) and P(t) next to them -- and the max tree's
            //      ~16 vector instructions ride in front of the first MFMA, where the wave waits for its V fragments anyway).  Slow path (rare): new
            //      m_run, S(t) -= delta at once; O and l, which still gather P(t-1) V(t-1) at the OLD scale in this step, are rescaled at its end. ----
            float alpha = 1.f;
            bool moved = false;
            if (has_soft) {
                const float mt = row_max(sc);
                moved = __any(mt > RESCALE_THR);
                if (moved) {
                    const float delta = fmaxf(mt, 0.f);
                    m_run += delta;
                    alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
                }
            }
            fence();
            static_for<5>([&](auto PH) { soft(std::integral_constant<int, 0>{}, PH); });
            // ---- O^T += V(t-1)^T P(t-1)^T ----
            static_for<4>([&](auto G) {
                constexpr int g = decltype(G)::value, gb = g & 1, kb = g >> 1, tt = g & 1;
                if constexpr (NDV == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]), "+v"(vl0[0][0]), "+v"(vl1[0][0]),
                                 "+v"(vh0[gb][NDV - 1]), "+v"(vh1[gb][NDV - 1]), "+v"(vl0[0][NDV - 1]), "+v"(vl1[0][NDV - 1]) :: "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vh0[gb][0]), "+v"(vh1[gb][0]), "+v"(vl0[0][0]), "+v"(vl1[0][0]) :: "memory");
                fence();
                f16x8 vh[NDV], vl[NDV];
#pragma unroll
                for (int d = 0; d < NDV; ++d) {
                    vh[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vh0[gb][d], vh1[gb][d], 0, 1, 2, 3, 4, 5, 6, 7));
                    vl[d] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vl0[0][d], vl1[0][d], 0, 1, 2, 3, 4, 5, 6, 7));
                }
                const f16x8 pf = __builtin_bit_cast(f16x8, pfw[kb][tt]), pl = __builtin_bit_cast(f16x8, plw[kb][tt]);
                static_for<3 * NDV>([&](auto M) {
                    constexpr int m = decltype(M)::value, d = m % NDV, pass = m / NDV;
                    if constexpr (pass == 0) oacc[d] = og_attn_mfma(vl[d], pf, oacc[d]);
                    else if constexpr (pass == 1) oacc[d] = og_attn_mfma(vh[d], pl, oacc[d]);
                    else oacc[d] = og_attn_mfma(vh[d], pf, oacc[d]);
                    fence();
                    // the next group's fragments.  The lo fragments are SINGLE-buffered (16 registers less): they feed the pass-0 MFMAs only (m < NDV), so
                    // their next reads go out behind the later MFMAs of the group, the double-buffered hi fragments behind the first ones.
                    if constexpr (g + 1 < 4) {
                        if constexpr (m < NDV) {                       // hi of dv block m
                            read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 4 * m>{});
                            read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 4 * m + 1>{});
                            fence();
                        } else if constexpr (m < 2 * NDV) {            // lo of dv block m - NDV: every pass-0 MFMA has been issued
                            read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 4 * (m - NDV) + 2>{});
                            read_v1(std::integral_constant<int, g + 1>{}, std::integral_constant<int, 4 * (m - NDV) + 3>{});
                            fence();
                        }
                    } else if constexpr (m >= 3 * NDV - 4 || NDV == 1) {          // the first K fragments of tile t+1 (hi first, see the QK^T loop)
                        constexpr int q = NDV == 2 ? m - 2 : m;                   // NDV = 2: behind MFMAs 2..5; NDV = 1: 0..2, the fourth with the third
                        if constexpr (q >= 0) {
                            if (has_qk) read_k1(std::integral_constant<int, 0>{}, std::integral_constant<int, og_korder(q)>{});
                            if constexpr (NDV == 1 && m == 2) { if (has_qk) read_k1(std::integral_constant<int, 0>{}, std::integral_constant<int, og_korder(3)>{}); }
                            fence();
                        }
                    }
                    after_mfma(std::integral_constant<int, 3 * NDV * g + m>{});
                });
            });
            // ---- S(t+1) = K(t+1) Q^T - m_run ----
            if (has_qk) {
                const f32x16 negm = make_negm();
                static_for<NCH>([&](auto C) {
                    constexpr int c = decltype(C)::value, cb = c & 1;
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh[cb][0]), "+v"(kh[cb][1]), "+v"(kl[0][0]), "+v"(kl[0][1]) :: "memory");
                    fence();
                    static_for<6>([&](auto M) {
                        constexpr int m = decltype(M)::value, kb = m & 1, pass = m >> 1;
                        if constexpr (pass == 0) sn[kb] = og_attn_mfma(kl[0][kb], qh[c], c == 0 ? negm : sn[kb]);
                        else if constexpr (pass == 1) sn[kb] = og_attn_mfma(kh[cb][kb], ql[c], sn[kb]);
                        else sn[kb] = og_attn_mfma(kh[cb][kb], qh[c], sn[kb]);
                        fence();
                        // the next chunk's fragments: hi (double-buffered) behind MFMAs 0, 1; lo (single-buffered, read by MFMAs 0, 1 only) behind 2, 3
                        if constexpr (m < 4 && c + 1 < NCH) { read_k1(std::integral_constant<int, c + 1>{}, std::integral_constant<int, og_korder(m)>{}); fence(); }
                        after_mfma(std::integral_constant<int, 12 * NDV + 6 * c + m>{});
                    });
                });
                if ((t + 1) * KV_TILE + KV_TILE > nk) mask_tile(sn, (t + 1) * KV_TILE);
            } else {
                // no QK^T in this step: the rest of the softmax and of the DMA slots on their own
                static_for<24>([&](auto M) { behind(std::integral_constant<int, 24 + decltype(M)::value>{}); });
                asm volatile("" : "=v"(sn[0]), "=v"(sn[1]));      // S(t+1) does not exist: "defined" here, so that the old contents are not kept alive through the step
            }
            // ---- l and, after a move of the running max, O: every P(t-1) V(t-1) product of this step is in ----
            if (has_soft) {
                if (moved) {
#pragma unroll
                    for (int d = 0; d < NDV; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
                }
                l_run = l_run * alpha + ((ps0 + ps1) + (ps2 + ps3));
            }
            if (t < ntiles) {               // another step follows: the first fragments of ITS V tile (V(t): landed a step ago), then: this step's DMA pieces landed,
                set_v_slot(vslot_off(vslot_next));          // everybody is done with the slots the next step's DMA will overwrite
                static_for<4 * NDV>([&](auto J) { read_v1(std::integral_constant<int, 0>{}, J); });
                fence();
                vslot_cur = vslot_next;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(OG_PIPE_ABL & 2)
                __syncthreads();
#endif
            }
        };
        // ---- prologue: S(0) -> m_run, P(0); S(1) ----
        qk_alone(0, sB, 0);
        {
            const float mt = row_max(sB);
            m_run = mt;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sB[kb][r] -= mt;
            l_run = exp_split_all(sB, pfA, plA);
        }
        if (ntiles > 1) {
            __syncthreads();                                     // everybody has read K(0): its slot takes K(2)
            if (ntiles > 2) static_for<NPI>([&](auto I) { pipe_dma(2, I, std::integral_constant<int, 0>{}, 0); });
            qk_alone(2 * PLANE, sA, 1);
        }
        set_v_slot(vslot_off(0));                                // the first fragments of V(0) for step 1
        static_for<4 * NDV>([&](auto J) { read_v1(std::integral_constant<int, 0>{}, J); });
        fence();
        if (ntiles > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- steps t = 1 .. ntiles: AB (t odd, buffer 0) and BA (t even, buffer 1) ----
        // (both steps of the loop body always run: a step skipped by a branch would keep the state it does not touch alive around it -- 64 registers)
        int t = 1;
        for (; t + 1 <= ntiles; t += 2) {
            step(t, sA, pfA, plA, sB, pfB, plB);
            step(t + 1, sB, pfB, plB, sA, pfA, plA);
        }
        if (t <= ntiles) step(t, sA, pfA, plA, sB, pfB, plB);
    } else
    if (ntiles > 0) {
        using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
        int kt = 0;
        for (; kt + 2 < ntiles; kt += 2) {                 // pairs of tiles that are not the last one: no mask code at all
            tile_step(kt, B0{}, std::false_type{});
            tile_step(kt + 1, B1{}, std::false_type{});
        }
        if (kt + 1 < ntiles) {
            tile_step(kt, B0{}, std::false_type{});
            tile_step(kt + 1, B1{}, std::true_type{});
        } else {
            tile_step(kt, B0{}, std::true_type{});
        }
    }

    if constexpr (KS == 2) {
        // the halves walked ntiles - 1 (or 0) barriers inside their loops: the shorter one catches up, then (O, m, l) of half 1 cross LDS
        const int mine = ntiles > 1 ? ntiles - 1 : 0, other = ntiles_other > 1 ? ntiles_other - 1 : 0;
        for (int i = mine; i < other; ++i) __syncthreads();
        __syncthreads();                                   // every wave is done with the K / V rings
        float* const xch = reinterpret_cast<float*>(smem_all) + (wave * (16 * NDV + 2)) * 64 + lane;      // [wave][register][lane]
        if (ntiles == 0) m_run = -1e30f;                   // an empty half weighs nothing in the merge
        if (half == 1) {
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64] = oacc[d][r];
            xch[(16 * NDV) * 64] = m_run;
            xch[(16 * NDV + 1) * 64] = l_run;
        }
        __syncthreads();
        if (half == 1) return;
        const float m1 = xch[(16 * NDV) * 64], l1 = xch[(16 * NDV + 1) * 64];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = oacc[d][r] * a0 + xch[(d * 16 + r) * 64] * a1;
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
    }
    if constexpr (GS > 1) {
        constexpr int PSTRIDE = (16 * NDV + 2) * 64;      // floats per wave: [register][lane]
        float* const tile_base = a.partial + ((int64_t)(grp * a.qtiles + qt) * GS * 4) * PSTRIDE;
        float* const mine = tile_base + (part * 4 + wave) * PSTRIDE + lane;
        // Hand-over, independent of where the parts run (MI355X_MICROARCH.md: the block -> XCD map is undefined): every access to the scratch is an
        // agent-scope relaxed atomic = `sc1` (write-through stores, L1-bypassing loads -- the form the guide lists as valid across XCDs: sc1 payload,
        // vmcnt(0), sc1 flag, sc1 loads), the arrival counter is an agent-scope RMW, and the counter word also carries every part's XCC id: if the last
        // arriver finds a part that ran on another XCD it takes a full agent-scope acquire (buffer_inv sc1) before it reads -- in the usual placement
        // (all parts of a tile on one XCD, which the block index mapping arranges for speed) no fence is executed at all.  A device-scope RELEASE
        // fence is deliberately not used: it writes the whole L2 back on this part (first version, with __threadfence(): attention 0.72 -> 1.16 ms
        // per single-pair step); the write-through stores are complete when vmcnt reaches 0.
        auto st = [](float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) st(mine + (d * 16 + r) * 64, oacc[d][r]);
        st(mine + (16 * NDV) * 64, ntiles == 0 ? -1e30f : m_run);      // an empty part weighs nothing in the merge
        st(mine + (16 * NDV + 1) * 64, l_run);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my partial result has reached the coherence point ...
        __syncthreads();                                   // ... and so has everybody's of this workgroup (and nobody reads the rings any more)
        int* const s_last = reinterpret_cast<int*>(smem_all);
        if (tid == 0) {
            int* const cnt = a.counters + grp * a.qtiles + qt;
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u;                 // HW_REG_XCC_ID, XCC_ID [3:0]
            const unsigned mine_word = 1u + (xcc << (8 + 4 * part));                              // arrivals in bits 0..7, part p's XCC id in bits 8 + 4p ..
            const unsigned old = (unsigned)__hip_atomic_fetch_add(cnt, (int)mine_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = (old & 255u) == GS - 1;
            int code = 0;
            if (last) {
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // the last arriver re-arms the counter for the next launch
                const unsigned all = (old + mine_word) >> 8;
                bool same = true;
#pragma unroll
                for (int p = 0; p < GS; ++p) same &= ((all >> (4 * p)) & 15u) == xcc;
                code = same ? 1 : 2;
            }
            *s_last = code;
        }
        __syncthreads();
        const int arrival = *s_last;
        if (!arrival) return;
        if (arrival == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // some part ran on another XCD
        // merge the GS partial results in index order (the same order whoever arrives last): m = max, O and l rescaled to it
        auto ld = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        const float* const src = tile_base + wave * PSTRIDE + lane;
        float mp[GS], m = -1e30f;
#pragma unroll
        for (int p = 0; p < GS; ++p) { mp[p] = ld(src + (int64_t)p * 4 * PSTRIDE + (16 * NDV) * 64); m = fmaxf(m, mp[p]); }
        l_run = 0.f;
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
        for (int p = 0; p < GS; ++p) {
            const float w = __builtin_amdgcn_exp2f(mp[p] - m);
            const float* const sp = src + (int64_t)p * 4 * PSTRIDE;
            l_run += ld(sp + (16 * NDV + 1) * 64) * w;
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] += ld(sp + (d * 16 + r) * 64) * w;
        }
        m_run = m;
    }
    // ---- normalise and store O[q][h*DH + dv] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qi = q0 + wave * 32 + l31;
    // row log-sum-exp of the (scaled) scores in natural units, for a backward pass that recomputes P (uniform calls only):
    // the kernel works in base 2 (q carries log2 e): L = ln 2 (m_run + log2 l)
    if (a.lse && hi == 0 && qi < nq) a.lse[((int64_t)z * a.num_heads + h) * nq + qi] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    if (qi < nq) {
        const int64_t orow = (q_row0 + qi) * a.ldo;
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = d * 32 + 8 * g4 + 4 * hi;
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



// -----------------------------------------------------------------------------------------------------------------------------------------------------
// attention_p16_kernel (round 6): the software-pipelined loop of attention_dma_kernel<64, ., 1, 1, 0, 1> on v_mfma_f32_16x16x32_f16 -- the f16 MFMA form that
// costs the least energy per flop on this part (scripts/probes/mfma_energy.hip: +19 % sustained under the power cap in isolation; the timing build of the
// 32x32x16 kernels issued as pairs of 16x16x32, profiles/r06_n_attention_abl16.log: -7 % for the phase form, -13 % for the pipelined form).  dh = 64, batch form.
//
// Tiling of a wave's 64-key x 32-query tile in 16 x 16 blocks (lane l: column l & 15, k-group / row-group g = l >> 4):
//   S^T block (i, j) = keys 16 i.. x queries 16 j..:  A = K rows (lane: key 16 i + (l & 15), channels 32 s + 8 g ..), B = Q^T (lane: query 16 j + (l & 15),
//       the same channels), two k-steps s; accumulator register r = key 16 i + 4 g + r.  A lane owns TWO queries (j = 0, 1) and 16 keys of each per tile.
//   O^T block (t, j) = channels 16 t.. x queries 16 j..:  B = P^T straight from the exponentiated accumulators -- for k-step s the lane's eight values are
//       blocks i = 2 s, 2 s + 1 = keys 32 s + 16 (e >> 2) + 4 g + (e & 3) -- and A = V^T by two transposing reads per plane (rows = those keys).
//   Row statistics of a query live in the four lanes {c, c + 16, c + 32, c + 48}: v_permlane16_swap + v_permlane32_swap.
// LDS: K rows swizzled as everywhere (chunk ^ ((row >> 1) & 7): the 16-lane groups of a ds_read_b128 -- rows 0-3, 12-15 at chunk c and rows 4-11 at chunk c ^ 1 --
// still cover sixteen distinct 16-byte slots); V rows chunk ^ (((row >> 1) & 3) << 1): a transposing read's 32 lanes fetch 8 key rows x 32 bytes.
// Loop structure, ring slots, DMA and the running-max handling are the PIPE form's (see there).
template <class RD>
__global__ __launch_bounds__(256, 2) void attention_p16_kernel(AttnArgs a, RD rd) {
    constexpr int DH = 64, ROWB = 128, RPI = 8, LPR = 8, NPI = 2, RW = 16;
    constexpr int PLANE = KV_TILE * ROWB;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(1024))) char smem_all[10 * PLANE];      // K ring of two slots (Kh | Kl), V ring of three (Vh | Vl)

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int grp = (local / a.qtiles) * 8 + xcd;          // (problem, head) group: all its query tiles on one XCD (speed only)
    if (grp >= a.nz * a.num_heads) return;
    const int qt = local % a.qtiles;
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
    const int q0 = qt * Q_TILE;
    if (q0 >= nq) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    char* const smem = smem_all;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- DMA: wave w fills rows [16 w, 16 w + 16) of each plane, two pieces of eight rows; K source chunk ^ ((r >> 1) & 7), V source chunk ^ (((r >> 1) & 3) << 1) ----
    const int rl = lane / LPR, pc = lane % LPR;
    const int ldkb = (int)a.ldk * 2, ldvb = (int)a.ldv * 2;
    unsigned ksw[2], vsw[2], koffs[2], voffs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * RPI + rl;                                   // + 16 w: invisible to either swizzle
        ksw[i] = (unsigned)(pc ^ ((r >> 1) & 7)) * 16u;
        vsw[i] = (unsigned)(pc ^ (((r >> 1) & 3) << 1)) * 16u;
        const int rr = wave * RW + r;
        koffs[i] = (unsigned)(rr * ldkb) + ksw[i];
        voffs[i] = (unsigned)(rr * ldvb) + vsw[i];
        asm volatile("" : "+v"(koffs[i]), "+v"(voffs[i]));
    }
    const int64_t k_tile0 = (kv_row0 * a.ldk + h * DH) * 2, v_tile0 = (kv_row0 * a.ldv + h * DH) * 2;   // bytes, uniform
    auto sptr = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    auto dma = [&](int kt, auto I, auto PP, int slot_off) {          // one (hi, lo) instruction pair of tile kt, piece i, into the ring slot at slot_off
        constexpr int i = decltype(I)::value, pp = decltype(PP)::value;
        const int key0 = kt * KV_TILE;
        const int last = nk - 1 - key0;
        unsigned off = pp == 0 ? koffs[i] : voffs[i];
        if (last < KV_TILE - 1) {                        // wave-uniform: only the last tile of a problem clamps its rows
            int r = wave * RW + i * RPI + rl;
            r = r < last ? r : last;
            off = pp == 0 ? (unsigned)(r * ldkb) + ksw[i] : (unsigned)(r * ldvb) + vsw[i];
        }
        const int64_t o = pp == 0 ? k_tile0 + (int64_t)key0 * ldkb : v_tile0 + (int64_t)key0 * ldvb;       // uniform
        const char* bh = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kh : a.vh) + o);
        const char* bl = sptr(reinterpret_cast<const char*>(pp == 0 ? a.kl : a.vl) + o);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot_off + (unsigned)((wave * RW + i * RPI) * ROWB));
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "s_add_u32 m0, m0, %4\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2"
                     :: "v"(off), "s"(bh), "s"(bl), "s"(m0v), "n"(PLANE) : "memory");
    };
    const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
    static_for<NPI>([&](auto I) {
        dma(0, I, std::integral_constant<int, 0>{}, 0);
        dma(0, I, std::integral_constant<int, 1>{}, 4 * PLANE);
        if (ntiles > 1) {
            dma(1, I, std::integral_constant<int, 0>{}, 2 * PLANE);
            dma(1, I, std::integral_constant<int, 1>{}, 6 * PLANE);
        }
    });

    // ---- Q fragments (B operand of K Q^T): lane (query 16 j + c16, channels 32 s + 8 g ..) ----
    f16x8 qh[2][2], ql[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int qi = q0 + wave * 32 + 16 * j + c16;
        if (qi >= nq) qi = nq - 1;     // clamp: computed but never stored
        const int64_t qo = (q_row0 + qi) * a.ldq + h * DH + 8 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qh[j][s] = *reinterpret_cast<const f16x8*>(a.qh + qo + 32 * s);
            ql[j][s] = *reinterpret_cast<const f16x8*>(a.ql + qo + 32 * s);
        }
    }
    f32x4_ oacc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[t][j] = f32x4_{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};

    // fragment addresses (bytes), moved in place from ring slot to ring slot
    unsigned kf[2], va[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) kf[s] = lds0 + c16 * ROWB + (((4 * s + g) ^ ((c16 >> 1) & 7)) * 16);
    {
        const int fsw = ((2 * g + (c16 >> 3)) & 3) << 1;
        const unsigned rowlane = (unsigned)((4 * g + (c16 >> 2)) * ROWB + 8 * (c16 & 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) va[t] = lds0 + rowlane + (unsigned)((((2 * t) ^ fsw) + ((c16 & 3) >> 1)) * 16);
    }
    int k_off_now = 0, v_off_now = 0;
    auto set_k_slot = [&](int off) { const unsigned d = (unsigned)(off - k_off_now); k_off_now = off; kf[0] += d; kf[1] += d; };
    auto set_v_slot = [&](int off) {
        const unsigned d = (unsigned)(off - v_off_now);
        v_off_now = off;
#pragma unroll
        for (int t = 0; t < 4; ++t) va[t] += d;
    };
    // K chunk (key-tile pair ip, k-step s): hi fragments double-buffered by chunk parity, lo fragments in ONE buffer (only the pass-0 MFMAs read them)
    f16x8 khf[2][2], klf[2];
    auto read_k = [&](auto CC, auto Q) {                 // piece q of chunk cc: 0, 1 = hi of key tiles 2 ip, 2 ip + 1; 2, 3 = lo
        constexpr int cc = decltype(CC)::value, q = decltype(Q)::value, ip = cc >> 1, s = cc & 1, aa = q & 1;
        if constexpr (q < 2) lds_read_b128<(2 * ip + aa) * 16 * ROWB>(khf[cc & 1][aa], kf[s]);
        else lds_read_b128<PLANE + (2 * ip + aa) * 16 * ROWB>(klf[aa], kf[s]);
    };
    // V chunk (k-step s, channel-tile pair tp): per channel tile two transposing reads per plane (keys 32 s + 4 g .., + 16); hi double-buffered, lo single
    s16x4 vhf[2][2][2], vlf[2][2];                       // [buffer][tile b2][half], [tile b2][half]
    auto read_v = [&](auto VC, auto Q) {                 // piece q of chunk vc: 0..3 = hi (tile q >> 1, half q & 1), 4..7 = lo
        constexpr int vc = decltype(VC)::value, q = decltype(Q)::value, s = vc >> 1, tp = vc & 1, b2 = (q >> 1) & 1, hh = q & 1, t = 2 * tp + b2;
        if constexpr (q < 4) lds_read_tr16_b64<(32 * s + 16 * hh) * ROWB>(vhf[vc & 1][b2][hh], va[t]);
        else lds_read_tr16_b64<PLANE + (32 * s + 16 * hh) * ROWB>(vlf[b2][hh], va[t]);
    };
    auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
    auto mfma16 = [](f16x8 x, f16x8 y, f32x4_ c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };
    auto add1 = [](float& acc, float x) { acc += x; asm("" : "+v"(acc)); };
    // maximum of a per-lane value over the four lanes that share a query column
    auto max_over_groups = [](float v) {
        float a0 = v, a1 = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
        float b0 = fmaxf(a0, a1), b1 = b0;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));
        return fmaxf(b0, b1);
    };
    auto sum_over_groups = [](float v) {
        float a0 = v, a1 = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
        float b0 = a0 + a1, b1 = b0;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));
        return b0 + b1;
    };
    auto row_max = [&](const f32x4_ (&sc)[4][2], int j) {
        float mt = sc[0][j][0];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sc[i][j][r]);
        return max_over_groups(mt);
    };
    auto make_negm = [&](int j) {
        float nm = -m_run[j];
        asm volatile("" : "+v"(nm));
        return f32x4_{nm, nm, nm, nm};
    };
    auto mask_tile = [&](f32x4_ (&sc)[4][2], int key0) {          // padded keys of the last tile -> -inf
        int gg = g;
        asm volatile("" : "+v"(gg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = key0 + 16 * i + 4 * gg + r < nk;
#pragma unroll
                for (int j = 0; j < 2; ++j) sc[i][j][r] = ok ? sc[i][j][r] : OG_NEG_INF;
            }
    };
    // softmax of a whole tile in one go (prologue): S -> P as packed (hi, hi) / (lo, lo) words; returns the row sums of the lane's two queries
    auto exp_split_all = [&](f32x4_ (&sc)[4][2], u32x4 (&pf)[2][2], u32x4 (&pl)[2][2], float (&sum)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p0 = __builtin_amdgcn_exp2f(sc[i][j][0]), p1 = __builtin_amdgcn_exp2f(sc[i][j][1]);
                const float p2 = __builtin_amdgcn_exp2f(sc[i][j][2]), p3 = __builtin_amdgcn_exp2f(sc[i][j][3]);
                add1(ps0, p0); add1(ps1, p1); add1(ps0, p2); add1(ps1, p3);
                unsigned ha, la, hb, lb;
                og_split4(p0, p1, p2, p3, ha, la, hb, lb);
                pf[j][i >> 1][(i & 1) * 2] = ha; pf[j][i >> 1][(i & 1) * 2 + 1] = hb;
                pl[j][i >> 1][(i & 1) * 2] = la; pl[j][i >> 1][(i & 1) * 2 + 1] = lb;
            }
            sum[j] = ps0 + ps1;
        }
    };
    // the twelve MFMAs of K chunk cc into the S accumulators `sn`, with the next chunk's fragment reads and `after(k)` behind MFMA k
    auto qk_chunk = [&](auto CC, f32x4_ (&sn)[4][2], const f32x4_ (&negm)[2], auto NEXT, auto&& after) {
        constexpr int cc = decltype(CC)::value, ip = cc >> 1, s = cc & 1, cb = cc & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(khf[cb][0]), "+v"(khf[cb][1]), "+v"(klf[0]), "+v"(klf[1]) :: "memory");
        fence();
        static_for<12>([&](auto M) {
            constexpr int m = decltype(M)::value, j = m & 1, aa = (m >> 1) & 1, pass = m >> 2, i = 2 * ip + aa;
            if constexpr (pass == 0) sn[i][j] = mfma16(klf[aa], qh[j][s], s == 0 ? negm[j] : sn[i][j]);
            else if constexpr (pass == 1) sn[i][j] = mfma16(khf[cb][aa], ql[j][s], sn[i][j]);
            else sn[i][j] = mfma16(khf[cb][aa], qh[j][s], sn[i][j]);
            fence();
            constexpr int nx = decltype(NEXT)::value;            // next K chunk to request (-1: none): hi behind MFMAs 0, 1; lo behind 4, 5 (every pass-0 MFMA is out)
            if constexpr (nx >= 0) {
                if constexpr (m < 2) { read_k(std::integral_constant<int, nx>{}, std::integral_constant<int, m>{}); fence(); }
                else if constexpr (m == 4 || m == 5) { read_k(std::integral_constant<int, nx>{}, std::integral_constant<int, m - 2>{}); fence(); }
            }
            after(std::integral_constant<int, 12 * cc + m>{});
        });
    };

    f32x4_ sA[4][2], sB[4][2];
    u32x4 pfA[2][2], plA[2][2], pfB[2][2], plB[2][2];

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto qk_alone = [&](int kslot_off, f32x4_ (&sn)[4][2], int kt) {
        set_k_slot(kslot_off);
        fence();
        static_for<4>([&](auto Q) { read_k(std::integral_constant<int, 0>{}, Q); });
        f32x4_ negm[2] = {make_negm(0), make_negm(1)};
        auto nothing = [](auto) {};
        qk_chunk(std::integral_constant<int, 0>{}, sn, negm, std::integral_constant<int, 1>{}, nothing);
        qk_chunk(std::integral_constant<int, 1>{}, sn, negm, std::integral_constant<int, 2>{}, nothing);
        qk_chunk(std::integral_constant<int, 2>{}, sn, negm, std::integral_constant<int, 3>{}, nothing);
        qk_chunk(std::integral_constant<int, 3>{}, sn, negm, std::integral_constant<int, -1>{}, nothing);
        if (kt * KV_TILE + KV_TILE > nk) mask_tile(sn, kt * KV_TILE);
    };

    // ring bookkeeping (wave-uniform): V(kt) lives in V slot kt % 3, K(kt) in K slot kt & 1
    int vslot_cur = 0;
    auto vslot_off = [](int sl) { return 4 * PLANE + sl * 2 * PLANE; };
    // One step t: O += V(t-1)^T P(t-1)^T (24 MFMAs... 48 of 16x16x32), S(t+1) = K(t+1) Q^T - m_run (48), softmax(t) dealt out behind them.
    auto step = [&](int t, f32x4_ (&sc)[4][2], u32x4 (&pfw)[2][2], u32x4 (&plw)[2][2], f32x4_ (&sn)[4][2], u32x4 (&nfw)[2][2], u32x4 (&nlw)[2][2]) {
        const bool has_soft = t < ntiles, has_qk = t + 1 < ntiles;
        const bool dma_v = t + 1 < ntiles, dma_k = t + 2 < ntiles;
        const int vslot_next = vslot_cur == 2 ? 0 : vslot_cur + 1;
        const int vslot_dma = vslot_next == 2 ? 0 : vslot_next + 1;
        const int kdma_off = (t & 1) * 2 * PLANE, vdma_off = vslot_off(vslot_dma);
        set_k_slot(((t + 1) & 1) * 2 * PLANE);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, psum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        asm volatile("" : "=v"(nfw[0][0]), "=v"(nfw[0][1]), "=v"(nfw[1][0]), "=v"(nfw[1][1]), "=v"(nlw[0][0]), "=v"(nlw[0][1]), "=v"(nlw[1][0]), "=v"(nlw[1][1]));
        // softmax(t), block blk = (key tile i = blk >> 1, query tile j = blk & 1), piece ph
        auto soft = [&](auto BLK, auto PH) {
            constexpr int blk = decltype(BLK)::value, ph = decltype(PH)::value, i = blk >> 1, j = blk & 1;
            if constexpr (ph == 0) { p0 = __builtin_amdgcn_exp2f(sc[i][j][0]); p1 = __builtin_amdgcn_exp2f(sc[i][j][1]); }
            else if constexpr (ph == 1) { p2 = __builtin_amdgcn_exp2f(sc[i][j][2]); p3 = __builtin_amdgcn_exp2f(sc[i][j][3]); }
            else if constexpr (ph == 2) {
                unsigned ha, la, hb, lb;
                og_split4(p0, p1, p2, p3, ha, la, hb, lb);
                nfw[j][i >> 1][(i & 1) * 2] = ha; nfw[j][i >> 1][(i & 1) * 2 + 1] = hb;
                nlw[j][i >> 1][(i & 1) * 2] = la; nlw[j][i >> 1][(i & 1) * 2 + 1] = lb;
            } else if constexpr (ph == 3) { add1(psum[j][0], p0); add1(psum[j][1], p1); }
            else if constexpr (ph == 4) { add1(psum[j][0], p2); add1(psum[j][1], p3); }
            fence();
        };
        // behind MFMA number k of the step (0..95): a step has 96 MFMAs and 48 slots for vector / DMA pieces: one slot behind every second MFMA
        auto behind = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr ((k & 1) == 1) {
                constexpr int mm = k >> 1;
                if constexpr (mm % 6 < 5 && mm / 6 + 1 < 8) soft(std::integral_constant<int, mm / 6 + 1>{}, std::integral_constant<int, mm % 6>{});
                if constexpr (mm >= 2 && (mm - 2) % 3 == 0) {
                    constexpr int jj = (mm - 2) / 12, w = ((mm - 2) % 12) / 3;       // pair jj: 0, 1 = K(t+2) pieces, 2, 3 = V(t+1) pieces of wave w
                    constexpr int i = jj & 1, pp = jj >> 1;
                    if (wave == w && (pp == 0 ? dma_k : dma_v)) dma(pp == 0 ? t + 2 : t + 1, std::integral_constant<int, i>{}, std::integral_constant<int, pp>{}, pp == 0 ? kdma_off : vdma_off);
                    fence();
                }
            }
        };
        fence();
        // ---- the running max moves before the exponentials (as in the PIPE form), per query tile ----
        float alpha[2] = {1.f, 1.f};
        bool moved = false;
        if (has_soft) {
            const float mt0 = row_max(sc, 0), mt1 = row_max(sc, 1);
            moved = __any(fmaxf(mt0, mt1) > RESCALE_THR);
            if (moved) {
                const float d0 = fmaxf(mt0, 0.f), d1 = fmaxf(mt1, 0.f);
                m_run[0] += d0; m_run[1] += d1;
                alpha[0] = __builtin_amdgcn_exp2f(-d0); alpha[1] = __builtin_amdgcn_exp2f(-d1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sc[i][0][r] -= d0; sc[i][1][r] -= d1; }
            }
        }
        fence();
        static_for<5>([&](auto PH) { soft(std::integral_constant<int, 0>{}, PH); });
        // ---- O^T += V(t-1)^T P(t-1)^T: chunks vc = (k-step s, channel-tile pair tp) ----
        static_for<4>([&](auto VC) {
            constexpr int vc = decltype(VC)::value, s = vc >> 1, tp = vc & 1, vb = vc & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vhf[vb][0][0]), "+v"(vhf[vb][0][1]), "+v"(vhf[vb][1][0]), "+v"(vhf[vb][1][1]),
                         "+v"(vlf[0][0]), "+v"(vlf[0][1]), "+v"(vlf[1][0]), "+v"(vlf[1][1]) :: "memory");
            fence();
            f16x8 vh[2], vl[2];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                vh[b2] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vhf[vb][b2][0], vhf[vb][b2][1], 0, 1, 2, 3, 4, 5, 6, 7));
                vl[b2] = __builtin_bit_cast(f16x8, __builtin_shufflevector(vlf[b2][0], vlf[b2][1], 0, 1, 2, 3, 4, 5, 6, 7));
            }
            static_for<12>([&](auto M) {
                constexpr int m = decltype(M)::value, j = m & 1, b2 = (m >> 1) & 1, pass = m >> 2, tt = 2 * tp + b2;
                const f16x8 pf = __builtin_bit_cast(f16x8, pfw[j][s]), pl = __builtin_bit_cast(f16x8, plw[j][s]);
                if constexpr (pass == 0) oacc[tt][j] = mfma16(vl[b2], pf, oacc[tt][j]);
                else if constexpr (pass == 1) oacc[tt][j] = mfma16(vh[b2], pl, oacc[tt][j]);
                else oacc[tt][j] = mfma16(vh[b2], pf, oacc[tt][j]);
                fence();
                // the next chunk's fragments: hi pieces behind MFMAs 0..3, lo pieces (single buffer: every pass-0 MFMA is out after MFMA 3) behind 4..7;
                // after the last V chunk: the first K fragments of tile t+1
                if constexpr (vc + 1 < 4) {
                    if constexpr (m < 8) { read_v(std::integral_constant<int, vc + 1>{}, std::integral_constant<int, m>{}); fence(); }
                } else {
                    if constexpr (m == 4 || m == 5) { if (has_qk) read_k(std::integral_constant<int, 0>{}, std::integral_constant<int, m - 4>{}); fence(); }
                    else if constexpr (m == 6 || m == 7) { if (has_qk) read_k(std::integral_constant<int, 0>{}, std::integral_constant<int, m - 4>{}); fence(); }
                }
                behind(std::integral_constant<int, 12 * vc + m>{});
            });
        });
        // ---- S(t+1) = K(t+1) Q^T - m_run ----
        if (has_qk) {
            f32x4_ negm[2] = {make_negm(0), make_negm(1)};
            auto aft = [&](auto K) { behind(std::integral_constant<int, 48 + decltype(K)::value>{}); };
            qk_chunk(std::integral_constant<int, 0>{}, sn, negm, std::integral_constant<int, 1>{}, aft);
            qk_chunk(std::integral_constant<int, 1>{}, sn, negm, std::integral_constant<int, 2>{}, aft);
            qk_chunk(std::integral_constant<int, 2>{}, sn, negm, std::integral_constant<int, 3>{}, aft);
            qk_chunk(std::integral_constant<int, 3>{}, sn, negm, std::integral_constant<int, -1>{}, aft);
            if ((t + 1) * KV_TILE + KV_TILE > nk) mask_tile(sn, (t + 1) * KV_TILE);
        } else {
            static_for<48>([&](auto M) { behind(std::integral_constant<int, 48 + decltype(M)::value>{}); });
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(sn[i][0]), "=v"(sn[i][1]));
        }
        if (has_soft) {
            if (moved) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { oacc[tt][0][r] *= alpha[0]; oacc[tt][1][r] *= alpha[1]; }
            }
            l_run[0] = l_run[0] * alpha[0] + (psum[0][0] + psum[0][1]);
            l_run[1] = l_run[1] * alpha[1] + (psum[1][0] + psum[1][1]);
        }
        if (t < ntiles) {
            set_v_slot(vslot_off(vslot_next));
            static_for<8>([&](auto Q) { read_v(std::integral_constant<int, 0>{}, Q); });
            fence();
            vslot_cur = vslot_next;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };

    // ---- prologue: S(0) -> m_run, P(0); S(1) ----
    qk_alone(0, sB, 0);
    {
        const float mt0 = row_max(sB, 0), mt1 = row_max(sB, 1);
        m_run[0] = mt0; m_run[1] = mt1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sB[i][0][r] -= mt0; sB[i][1][r] -= mt1; }
        exp_split_all(sB, pfA, plA, l_run);
    }
    if (ntiles > 1) {
        __syncthreads();                                     // everybody has read K(0): its slot takes K(2)
        if (ntiles > 2) static_for<NPI>([&](auto I) { dma(2, I, std::integral_constant<int, 0>{}, 0); });
        qk_alone(2 * PLANE, sA, 1);
    }
    set_v_slot(vslot_off(0));
    static_for<8>([&](auto Q) { read_v(std::integral_constant<int, 0>{}, Q); });
    fence();
    if (ntiles > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int t = 1;
    for (; t + 1 <= ntiles; t += 2) {
        step(t, sA, pfA, plA, sB, pfB, plB);
        step(t + 1, sB, pfB, plB, sA, pfA, plA);
    }
    if (t <= ntiles) step(t, sA, pfA, plA, sB, pfB, plB);

    // ---- normalise and store O[q][h * 64 + channel]: lane (query tile j, column c16) holds channels 16 tt + 4 g .. + 3 of each channel tile ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float l_tot = sum_over_groups(l_run[j]);
        const float inv = 1.f / l_tot;
        const int qi = q0 + wave * 32 + 16 * j + c16;
        if (a.lse && g == 0 && qi < nq) a.lse[((int64_t)z * a.num_heads + h) * nq + qi] = (m_run[j] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
        if (qi < nq) {
            const int64_t orow = (q_row0 + qi) * a.ldo;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int dv = 16 * tt + 4 * g;
                f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = oacc[tt][j][e] * inv;
                    asm("" : "+v"(o));
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
    // dh = 64 with full-line head rows: the LDS-DMA kernel (OG_ATTN_DMA=0 keeps the register-staged one, for A/B runs)
    static const bool dma_on = [] { const char* e = getenv("OG_ATTN_DMA"); return !(e && e[0] == '0'); }();
    const bool dma = dma_on && (a.dh == 64 || a.dh == 32);       // head rows of one or half a 128-byte line: the LDS-DMA kernel
    const int groups8 = (a.nz * a.num_heads + 7) / 8 * 8;
    dim3 grid(groups8 * a2.qtiles), block(256);
    // Few workgroups (at most one per CU) and enough keys: the key range of a query tile is split over the two halves of an 8-wave
    // workgroup (attention_dma_kernel<64, ., 2>).  OG_ATTN_KSPLIT=0 / 1 forces.
    static const int ks_mode = [] { const char* e = getenv("OG_ATTN_KSPLIT"); return e ? atoi(e) : -1; }();
    int nkmin = 1 << 30;
    for (int g = 0; g < 2; ++g) {
        const bool used = g == 0 ? a.split > 0 : a.split < a.nz;
        if (used && a.nk[g] < nkmin) nkmin = a.nk[g];
    }
    // One or two pairs (the grid covers at most half of the CUs) and scratch at hand: the key tiles of a query tile go to 2 or 4 workgroups
    // (attention_dma_kernel<64, ., 1, GS>).  OG_ATTN_GSPLIT=0 / 2 / 4 forces.
    static const int gs_mode = [] { const char* e = getenv("OG_ATTN_GSPLIT"); return e ? atoi(e) : -1; }();
    int gs = 1;
    if (!a.mx && dma && (a.dh == 64 || a.dh == 32) && a.partial && a.counters && (int)grid.x <= OG_ATTN_COUNTERS) {
        // The split launch may hold up to two workgroups per CU at dh = 32 (they do not wait for each other: the last arriver of a query tile merges),
        // one at dh = 64 -- measured in one call (profiles/r05_j_bench_attn_gsplit_maxwg_ab.jsonl): dh = 32, 512 against 256 workgroups: a single
        // 2048-keypoint pair 1.428 -> 1.394 ms per step, two pairs 1.703 -> 1.640, one 4096-keypoint pair 3.47 -> 3.24; dh = 64: one pair +-0, two pairs
        // 1.962 -> 1.997, four 2.385 -> 2.427 (slower).  OG_ATTN_GS_MAXWG overrides (experiments).
        static const int maxwg_env = [] { const char* e = getenv("OG_ATTN_GS_MAXWG"); return e ? atoi(e) : 0; }();
        const int maxwg = maxwg_env > 0 ? maxwg_env : (a.dh == 32 ? 512 : 256);
        if (gs_mode >= 0) gs = gs_mode == 2 || gs_mode == 4 ? gs_mode : 1;
        else if ((int)grid.x * 4 <= maxwg && (a.rag || nkmin >= 16 * KV_TILE)) gs = 4;
        else if ((int)grid.x * 2 <= maxwg && (a.rag || nkmin >= 8 * KV_TILE)) gs = 2;
        if ((int)grid.x * gs > maxwg && gs_mode < 0) gs = 1;
        if ((int64_t)grid.x * gs * 4 * 34 * 64 > OG_ATTN_PARTIAL_FLOATS) gs = 1;
    }
    if (gs > 1) {
        dim3 g2(grid.x * gs);
        const char* const sc = getenv("OG_ATTN_GS_SCATTER");      // tests (read per launch, so that one process can compare both placements): deal the parts of a tile to different XCDs
        a2.gs_scatter = sc && sc[0] == '1' ? 1 : 0;
        if (a.dh == 64) {
            if (a.rag) {
                if (gs == 4) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 1, 4>), g2, block, 0, stream, a2, rd);
                else hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 1, 2>), g2, block, 0, stream, a2, rd);
            } else {
                if (gs == 4) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 1, 4>), g2, block, 0, stream, a2, RaggedNone{});
                else hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 1, 2>), g2, block, 0, stream, a2, RaggedNone{});
            }
        } else {          // dh = 32 (round 5): a single 2048-keypoint pair of the 128-d family = 128 workgroups of 32 key tiles each without the split
            if (a.rag) {
                if (gs == 4) hipLaunchKernelGGL((attention_dma_kernel<32, RaggedDesc, 1, 4>), g2, block, 0, stream, a2, rd);
                else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedDesc, 1, 2>), g2, block, 0, stream, a2, rd);
            } else {
                if (gs == 4) hipLaunchKernelGGL((attention_dma_kernel<32, RaggedNone, 1, 4>), g2, block, 0, stream, a2, RaggedNone{});
                else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedNone, 1, 2>), g2, block, 0, stream, a2, RaggedNone{});
            }
        }
        return og_launch_status();
    }
    const bool ksplit = !a.mx && dma && a.dh == 64 && (ks_mode >= 0 ? ks_mode != 0 : ((int)grid.x <= 256 && (a.rag || nkmin >= 4 * KV_TILE)));
    if (ksplit) {
        if (a.rag) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 2>), grid, dim3(512), 0, stream, a2, rd);
        else hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 2>), grid, dim3(512), 0, stream, a2, RaggedNone{});
        return og_launch_status();
    }
    if (a.mx) {          // 8-bit P V cross products: batch form of the LDS-DMA kernel only (the callers set mx only where the producers wrote the 8-bit rows)
        if (!dma) return OG_E_SHAPE;
        const int e = 127 + a.mx_sv - 11;
        a2.mx_scale_a = e | (e << 8) | (e << 16) | (e << 24);
        if (a.rag) {
            if (a.dh == 64) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 1, 1, 1>), grid, block, 0, stream, a2, rd);
            else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedDesc, 1, 1, 1>), grid, block, 0, stream, a2, rd);
        } else {
            if (a.dh == 64) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 1, 1, 1>), grid, block, 0, stream, a2, RaggedNone{});
            else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedNone, 1, 1, 1>), grid, block, 0, stream, a2, RaggedNone{});
        }
        return og_launch_status();
    }
    // the 16x16x32 pipelined kernel (dh = 64, batch form): OG_ATTN_P16=0 / 1
    static const int p16_mode = [] { const char* e = getenv("OG_ATTN_P16"); return e ? atoi(e) : -1; }();
    if (dma && a.dh == 64 && (p16_mode >= 0 ? p16_mode != 0 : false)) {
        if (a.rag) hipLaunchKernelGGL((attention_p16_kernel<RaggedDesc>), grid, block, 0, stream, a2, rd);
        else hipLaunchKernelGGL((attention_p16_kernel<RaggedNone>), grid, block, 0, stream, a2, RaggedNone{});
        return og_launch_status();
    }
    // the software-pipelined tile loop (PIPE) for the batch form; OG_ATTN_PIPE=0 / 1 forces
    static const int pipe_mode = [] { const char* e = getenv("OG_ATTN_PIPE"); return e ? atoi(e) : -1; }();
    const bool pipe = dma && (pipe_mode >= 0 ? pipe_mode != 0 : false);
#ifndef OG_ATTN_PIPE8
#define OG_ATTN_PIPE8 0      // experiment build (scripts/build_attn_ablation.sh pipe8 -DOG_ATTN_PIPE8=1): instantiates the eight-wave form, OG_ATTN_PIPE=2 selects it.
#endif                       // Measured (profiles/r06_j_attention_pipe8_ab.log): parity as the other forms, 230 vs 221 us at the C2 shape, 840 vs 773 us at 2048 keys.
#if OG_ATTN_PIPE8
    if (pipe && pipe_mode == 2 && a.dh == 64) {          // eight waves, 256 queries per workgroup
        a2.qtiles = (nqmax + 2 * Q_TILE - 1) / (2 * Q_TILE);
        dim3 grid8(groups8 * a2.qtiles);
        if (a.rag) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 1, 1, 0, 2>), grid8, dim3(512), 0, stream, a2, rd);
        else hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 1, 1, 0, 2>), grid8, dim3(512), 0, stream, a2, RaggedNone{});
        return og_launch_status();
    }
#endif
    if (pipe) {
        if (a.rag) {
            if (a.dh == 64) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc, 1, 1, 0, 1>), grid, block, 0, stream, a2, rd);
            else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedDesc, 1, 1, 0, 1>), grid, block, 0, stream, a2, rd);
        } else {
            if (a.dh == 64) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone, 1, 1, 0, 1>), grid, block, 0, stream, a2, RaggedNone{});
            else hipLaunchKernelGGL((attention_dma_kernel<32, RaggedNone, 1, 1, 0, 1>), grid, block, 0, stream, a2, RaggedNone{});
        }
        return og_launch_status();
    }
    if (a.rag) {
        switch (a.dh) {
            case 16: hipLaunchKernelGGL((attention_kernel<16, RaggedDesc>), grid, block, 0, stream, a2, rd); break;
            case 128: hipLaunchKernelGGL((attention_kernel<128, RaggedDesc>), grid, block, 0, stream, a2, rd); break;      // register-staged kernel, generic in the head size
            case 32:
                if (dma) hipLaunchKernelGGL((attention_dma_kernel<32, RaggedDesc>), grid, block, 0, stream, a2, rd);
                else hipLaunchKernelGGL((attention_kernel<32, RaggedDesc>), grid, block, 0, stream, a2, rd);
                break;
            case 64:
                if (dma) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedDesc>), grid, block, 0, stream, a2, rd);
                else hipLaunchKernelGGL((attention_kernel<64, RaggedDesc>), grid, block, 0, stream, a2, rd);
                break;
            default: return OG_E_SHAPE;
        }
    } else {          // uniform batch: no descriptor in the kernarg segment (og_common.h: RaggedNone)
        switch (a.dh) {
            case 16: hipLaunchKernelGGL((attention_kernel<16, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{}); break;
            case 128: hipLaunchKernelGGL((attention_kernel<128, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{}); break;      // register-staged kernel, generic in the head size
            case 32:
                if (dma) hipLaunchKernelGGL((attention_dma_kernel<32, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{});
                else hipLaunchKernelGGL((attention_kernel<32, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{});
                break;
            case 64:
                if (dma) hipLaunchKernelGGL((attention_dma_kernel<64, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{});
                else hipLaunchKernelGGL((attention_kernel<64, RaggedNone>), grid, block, 0, stream, a2, RaggedNone{});
                break;
            default: return OG_E_SHAPE;
        }
    }
    return og_launch_status();
}

#if OG_ATTN_TRACE
extern "C" int og_debug_attn_trace(void* host_dst, size_t bytes) {
    if (bytes > sizeof(og_attn_trace_buf)) bytes = sizeof(og_attn_trace_buf);
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(og_attn_trace_buf), bytes);
}
#endif

extern "C" int og_attention(const void* qh, const void* ql, int64_t ldq, const void* kh, const void* kl, int64_t ldk,
                            const void* vh, const void* vl, int64_t ldv, void* oh, void* ol, int64_t ldo, int32_t batch,
                            int32_t nq, int32_t nk, int32_t num_heads, int32_t dh, float* lse, void* stream) {
    og_clear_status();
    AttnArgs a{};
    a.lse = lse;
    a.qh = (const _Float16*)qh; a.ql = (const _Float16*)ql; a.ldq = ldq;
    a.kh = (const _Float16*)kh; a.kl = (const _Float16*)kl; a.ldk = ldk;
    a.vh = (const _Float16*)vh; a.vl = (const _Float16*)vl; a.ldv = ldv;
    a.oh = (_Float16*)oh; a.ol = (_Float16*)ol; a.ldo = ldo; a.o_hl = 0;
    a.nz = batch; a.num_heads = num_heads; a.dh = dh; a.split = batch;
    a.q_base[0] = 0; a.q_step[0] = nq; a.kv_base[0] = 0; a.kv_step[0] = nk;
    a.nq[0] = nq; a.nk[0] = nk;
    a.q_base[1] = a.q_step[1] = a.kv_base[1] = a.kv_step[1] = 0; a.nq[1] = a.nk[1] = 0;
    // experiments (scripts/bench_attention.py): OG_ATTN_MX_SV=<sv> declares that `vl` holds the 8-bit rows of the MX form (og_common.h: AttnArgs::mx)
    if (const char* e = getenv("OG_ATTN_MX_SV")) { a.mx = 1; a.mx_sv = atoi(e); }
    return og_launch_attention(a, (hipStream_t)stream);
}

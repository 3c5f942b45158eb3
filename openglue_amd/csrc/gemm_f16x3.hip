// Split-f16 "3-pass" NT GEMM on the gfx950 matrix cores: fp32-class accuracy at 16/3 the rate of the
// exact-fp32 MFMA.
//
// Used for the 1x1 convolutions inside the attentional GNN (reference attention_gnn.py:16-20, 41:
// in_proj_q/k/v, fc.0, fc.3 -- 98 % of the GEMM FLOPs of the path).  fp32 MFMA runs at 1/16 of the f16
// rate and the parity bar (1e-3 on log-scores) rules out plain f16/bf16 operands, so
// every operand is carried as TWO f16 numbers  x = hi + lo  (og_common.h: lo at its true scale) and
//        X Wᵀ  =  Xh Whᵀ + Xh Wlᵀ + Xl Whᵀ                        (the lo*lo term is 2^-22 relative: dropped)
// with fp32 accumulation in ONE accumulator: measured error equals the fp32 GEMM's.
// Operand range: |x| < 65504 (f16); activations of this network are O(10).
//
// The (hi, lo) pairs are produced by the PRODUCER's epilogue (this kernel, the attention kernel, the fp32 GEMM
// of the encoder) and weights are split once at pack time, so no conversion happens on the load path.
// Both operands arrive in the hl32 row format (og_common.h): hi and lo of a 32-channel group share one
// 128-byte line.
//
// MFMA orientation: D[outch][token] = W · Xᵀ  (A operand = weight tile, B operand = token tile), so a lane
// owns ONE token (column l&31) and, per 4-register group, FOUR CONSECUTIVE output channels: bias, residual
// and all stores are vectorised along the channel axis.
// v_mfma_f32_32x32x16_f16; block tile 128 tokens x OC channels x 32 k, 4 waves as 2x2.
#include <stdlib.h>
#include <cmath>
#include <type_traits>

#include "og_common.h"

namespace {

constexpr int TOK = 128;
constexpr int BKH = 32;          // k per tile (halves)
constexpr int EPI_SLAB = 32 * 144;   // per-wave epilogue scratch for one 32-token slice: sized for the padded rows (128 B + 16 B) of the slow epilogues;
                                     // the fast 256-tile epilogue uses 128-byte swizzled rows inside it (gemm_nt_f16x3_big epilogue, rd_off)

// Compile-time ablations of the LDS-DMA kernel (scripts/build_ablation.sh; results are wrong by construction):
// 1 = no global stores, 2 = every block reads token tile 0 (operands L2-resident), 4 = no MFMA,
// 8 = (256-tile kernels) no LDS-DMA after the first stages, 16 = (256-tile kernels) no fragment reads after the first.
#ifndef OG_GEMM_ABL
#define OG_GEMM_ABL 0
#endif

// Epilogue.  After the MFMAs a lane owns ONE token and 4 consecutive channels per register group; storing
// that directly means 8-byte pieces scattered over 32 rows per instruction (measured: 113 of 210 us of the
// qkv GEMM).  Instead every wave transposes 32-token slices of its tile through its own LDS slab and
// writes whole rows: 16 B per lane, 128-byte (or 64-byte) contiguous row segments.
// `slab` = this wave's private LDS scratch (EPI_SLAB bytes), free once all waves passed the last
// k-tile barrier.  No block barrier is needed: a wave only re-reads what it wrote itself.
//
// Global operands of the epilogue never sit on its critical path one by one.  [Round 1 loaded every float4 of
// bias / residual inside its own `if (oc < N && tok < M)`; hipcc emitted `global_load ; s_waitcnt vmcnt(0)` per
// 4-value group: 64 exposed L2 round trips per wave per 256x256 tile, ~20 us of a 55 us tile -- the whole
// "MFMAs + barriers + epilogue arithmetic" gap of the round-1 ablation study.]
//   * the bias is folded into the accumulator initialisation (gemm_f16x3_acc_init, in the shadow of the first
//     LDS-DMA wait): acc0 = bias / scale, so that acc * scale = W x + bias;
//   * the residual of a 32-token slice (8 * TI/2 sixteen-byte pieces per lane) is fetched with clamped, always
//     valid addresses under wave-uniform conditions only, one slice AHEAD of the slice being finished.
typedef unsigned og_u32x4 __attribute__((ext_vector_type(4)));

// acc[i][j][4q + e] = bias[oc0 + 32 i + 8 q + 4 hi + e] / scale for every token block j; clears g.bias
template <int NI, int NJ>
__device__ __forceinline__ void gemm_f16x3_acc_init(GemmHArgs& g, f32x16 (&acc)[NI][NJ], int oc0, int lane) {
    const int hi = lane >> 5;
    f32x4 b[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) b[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = oc0 + i * 32 + 8 * q + 4 * hi;
                b[i][q] = *reinterpret_cast<const f32x4*>(g.bias + (oc < g.N ? oc : 0)) * g.inv_scale;   // bias => N % 4 == 0
            }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = b[i][q][e];
    g.bias = nullptr;
}

// residual of the 32-token slice starting at tok0 (one 4-dword slot per group for either residual form: they are
// mutually exclusive, og_launch_gemm_f16x3)
// ONLY_HL: the caller knows that the residual is the (hi, lo) form (no run-time branch whose two sides would have to be merged
// through 32 register copies)
template <int TI, bool ONLY_HL = false>
__device__ __forceinline__ void gemm_f16x3_load_residual(const GemmHArgs& g, og_u32x4 (&raw)[TI][4], int tok0, int oc0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int tok = tok0 + l31;
    const int tokc = tok < g.M ? tok : g.M - 1;              // clamped: rows past M are computed, never stored
    // N % 32 == 0 with either residual form (og_launch_gemm_f16x3): a 32-channel block is valid or not as a whole, so the
    // column clamp is wave-uniform and every load is one per-lane row pointer + a scalar block offset + an immediate
    if (!ONLY_HL && g.res) {
        const float* rp = g.res + (int64_t)tokc * g.ldr + 4 * hi;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ocb = oc0 + i * 32;
            const float* p2 = rp + (ocb < g.N ? ocb : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[i][q] = *reinterpret_cast<const og_u32x4*>(p2 + 8 * q);
        }
    } else if (ONLY_HL || g.res_hl) {                        // residual carried as (hi, lo): 2^-22 relative
        const _Float16* rp = g.res_hl + (int64_t)tokc * g.ldrh + 4 * hi;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ocb = oc0 + i * 32;
            const _Float16* p2 = rp + og_hl_col(ocb < g.N ? ocb : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 h2 = *reinterpret_cast<const uint2*>(p2 + 8 * q), l2 = *reinterpret_cast<const uint2*>(p2 + 8 * q + 32);
                raw[i][q] = og_u32x4{h2.x, h2.y, l2.x, l2.y};
            }
        }
    }
}

// One 32-token x TI*32-channel slice: a[i] = the accumulators of channel block i, raw = its residual (if any).
// FULL: also the residual MIX (alpha) and the channel-first copy (Ct) of the final projection; only instantiated for the
// 128-tile kernel.
template <int TI, bool FULL>
__device__ __forceinline__ void gemm_f16x3_epilogue_finish(const GemmHArgs& g, f32x16 (&a)[TI], const og_u32x4 (&raw)[TI][4], int oc0,
                                                           int lane) {
#pragma clang fp contract(off)                  // og_split: hi and lo must see the same rounded value (og_common.h)
    const int hi = lane >> 5;
    const float relu_floor = g.relu ? 0.f : OG_NEG_INF;      // max(v, -inf) = v: no branch per group

    // finish the arithmetic in registers: v = acc * scale (bias already inside) (+relu) (+res)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[i][r] = fmaxf(a[i][r] * g.scale, relu_floor);
    if (g.res) {
        if (FULL && g.alpha) {      // residual mix (superglue.py:60-62): alpha*v + (1-alpha)*res
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = oc0 + i * 32 + 8 * q + 4 * hi;
                    const f32x4 al = *reinterpret_cast<const f32x4*>(g.alpha + (oc < g.N ? oc : 0));
                    const f32x4 rr = __builtin_bit_cast(f32x4, raw[i][q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][4 * q + e] = al[e] * a[i][4 * q + e] + (1.f - al[e]) * rr[e];
                }
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 rr = __builtin_bit_cast(f32x4, raw[i][q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][4 * q + e] += rr[e];
                }
        }
    } else if (g.res_hl) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f16x4 rh = __builtin_bit_cast(f16x4, uint2{raw[i][q][0], raw[i][q][1]});
                const f16x4 rl = __builtin_bit_cast(f16x4, uint2{raw[i][q][2], raw[i][q][3]});
#pragma unroll
                for (int e = 0; e < 4; ++e) a[i][4 * q + e] += (float)rh[e] + (float)rl[e];
            }
    }

}

// ... and its stores (a = the finished values of the slice)
template <int TI, bool FULL, class RD>
__device__ __forceinline__ void gemm_f16x3_epilogue_store(const GemmHArgs& g, f32x16 (&a)[TI], int tok0, int oc0, int lane, char* slab,
                                                          const RD& rd) {
#pragma clang fp contract(off)                  // og_split: hi and lo must see the same rounded value (og_common.h)
    constexpr int OCW = TI * 32;                // channels of a wave tile
    const int l31 = lane & 31, hi = lane >> 5;
    const int tok = tok0 + l31;
    // ---- channel-first fp32 copy: a lane owns one token, so 32 consecutive lanes write 32 consecutive tokens of a channel ----
    if (FULL && g.Ct && tok < g.M) {
        float* cb;
        int64_t ldct = g.ldct;
        if (g.ct_rag) {      // ragged batch: pair b's [N][rows_b] block starts at N * off[b] (superglue.py:68-72 per pair)
            const int* off = g.ct_rag == 1 ? rd.off0 : rd.off1;
            int lo = 0, hb = rd.B - 1;           // last b with off[b] <= tok
            while (lo < hb) {
                const int mid = (lo + hb + 1) >> 1;
                if (off[mid] <= tok) lo = mid; else hb = mid - 1;
            }
            ldct = off[lo + 1] - off[lo];
            cb = g.Ct + (int64_t)off[lo] * g.N + (tok - off[lo]);
        } else {
            const int bz = tok / g.ct_rows, ri = tok - bz * g.ct_rows;
            cb = g.Ct + (int64_t)bz * g.ldct * g.N + ri;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int oc = oc0 + i * 32 + 8 * q + 4 * hi + e;
                    if (oc < g.N) cb[(int64_t)oc * ldct] = a[i][4 * q + e];
                }
    }
    // ---- hl32 rows: per 32-channel group one pass through a [32 tok][hi 64 B | lo 64 B] slab, stored as whole
    //      128-byte lines (8 lanes x 16 B per token) ----
    if (g.Ch && g.c_hl) {
        constexpr int ROWB = 128 + 16;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned ha, la, hb, lb;
                og_split4(a[i][4 * q], a[i][4 * q + 1], a[i][4 * q + 2], a[i][4 * q + 3], ha, la, hb, lb);
                char* d = slab + l31 * ROWB + (8 * q + 4 * hi) * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(ha, hb);
                *reinterpret_cast<uint2*>(d + 64) = make_uint2(la, lb);
            }
            const int oc = oc0 + i * 32;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), c = lane & 7;
                const f16x8 t = *reinterpret_cast<const f16x8*>(slab + r * ROWB + c * 16);
                const int tk = tok0 + r;
                if (tk < g.M && oc < g.N && !((OG_GEMM_ABL & 1) && tk >= 0))
                    *reinterpret_cast<f16x8*>(g.Ch + (int64_t)tk * g.ldch + og_hl_col(oc) + c * 8) = t;
            }
        }
    }
    // ---- split-f16 planes: two passes (hi, lo) through a [32 tok][OCW halves] slab ----
    if (g.Ch && !g.c_hl) {
        constexpr int ROWB = OCW * 2 + 16;              // padded LDS row (bytes), 16-byte aligned
        constexpr int CPR = OCW * 2 / 16;               // 16-byte chunks per row
        constexpr int RPI = 64 / CPR;                   // rows per store instruction
        unsigned sh[TI][4][2], sl[TI][4][2];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                og_split4(a[i][4 * q], a[i][4 * q + 1], a[i][4 * q + 2], a[i][4 * q + 3], sh[i][q][0], sl[i][q][0], sh[i][q][1], sl[i][q][1]);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2*>(slab + l31 * ROWB + (i * 32 + 8 * q + 4 * hi) * 2) =
                        pass == 0 ? make_uint2(sh[i][q][0], sh[i][q][1]) : make_uint2(sl[i][q][0], sl[i][q][1]);
            _Float16* dst = pass == 0 ? g.Ch : g.Cl;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + lane / CPR, c = lane % CPR;
                const f16x8 t = *reinterpret_cast<const f16x8*>(slab + r * ROWB + c * 16);
                const int tk = tok0 + r, oc = oc0 + c * 8;
                if (tk < g.M && oc < g.N && !((OG_GEMM_ABL & 1) && tk >= 0)) *reinterpret_cast<f16x8*>(dst + (int64_t)tk * g.ldch + oc) = t;
            }
        }
    }
    // ---- fp32 output: one pass per 32-channel block through a [32 tok][32 floats] slab ----
    if (g.C32) {
        constexpr int ROWB = 32 * 4 + 16;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = a[i][4 * q + e];
                *reinterpret_cast<f32x4*>(slab + l31 * ROWB + (8 * q + 4 * hi) * 4) = t;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), c = lane & 7;
                const f32x4 t = *reinterpret_cast<const f32x4*>(slab + r * ROWB + c * 16);
                const int tk = tok0 + r, oc = oc0 + i * 32 + c * 4;
                if (tk < g.M && oc < g.N && !((OG_GEMM_ABL & 1) && tk >= 0)) *reinterpret_cast<f32x4*>(g.C32 + (int64_t)tk * g.ldc + oc) = t;
            }
        }
    }
}

// Fast path of the 256-tile kernels: the whole 64-channel x 128-token wave tile is inside the matrix, the output is hl32
// rows (HL) or (hi, lo) planes, no fp32 / channel-first copy.  Per 32-token slice both channel blocks go through TWO slabs
// (HL: one per channel block; planes: one per plane) -- 16 LDS writes, 8 LDS reads, 8 stores with no predicate, no per-store
// address arithmetic: the row/column part of a store address is a scalar base, the lane part one 32-bit offset computed once.
// [The generic path above costs ~13k cycles per tile (scripts/trace_gemm.py): every store sits under its own exec-mask branch
// with a 64-bit multiply for its row, and each of the 8 (slice, channel block) passes waits for its own LDS round trip.]
// EM: the arithmetic of the epilogue fixed at compile time (chosen by the launcher), 0 = decided at run time by the generic code
// above.  [The run-time form costs, per 32-token slice and wave, 32 v_mul + 32 v_max whatever the activation, 3 VALU per element
// for a (hi, lo) residual and ~32 v_mov: the wave-uniform branches on g.res / g.res_hl / g.alpha merge their results through
// register copies.  VALU and MFMA issue do not overlap on this part, so these count.]
//   1: v = acc * scale                  (q/k/v projections: packed multiplies, 0.5 VALU per element)
//   2: v = max(acc * scale, 0)          (fc.0)
//   3: v = acc * scale + rh + rl        (fc.3: the (hi, lo) residual enters through two mixed-precision FMAs, 2 VALU per element)
enum { OG_EM_RUNTIME = 0, OG_EM_NONE = 1, OG_EM_RELU = 2, OG_EM_RES_HL = 3 };

template <int EM>
__device__ __forceinline__ void gemm_f16x3_epilogue_finish_spec(const GemmHArgs& g, f32x16 (&a)[2], const og_u32x4 (&raw)[2][4]) {
    const float sc = g.scale;
    if constexpr (EM == OG_EM_NONE) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = a[i] * sc;                    // vector form: v_pk_mul_f32
    } else if constexpr (EM == OG_EM_RELU) {
        // multiply first: the product is known to be canonical, so fmaxf is ONE v_max (on the raw accumulator it costs a second
        // v_max that quiets a possible NaN)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = a[i] * sc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[i][r] = fmaxf(a[i][r], 0.f);
    } else {
        static_assert(EM == OG_EM_RES_HL, "unknown epilogue mode");
        // raw[i][q] = {h01, h23, l01, l23} (f16 pairs).  v_fma_mix_f32: f32 acc * f32 scale + f16 hi, then f16 lo * 1.0 + that.
        // Full-register VALU writes (no op_sel partial-write hazard).  The hazard recognizer does not look inside inline asm: the only
        // software-managed hazard in reach is "MFMA writes a VGPR -> VALU reads it" (up to 19 wait states); between the last MFMA and this
        // block lie the block barrier, the residual loads and their s_waitcnt vmcnt -- hundreds of cycles.
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x0 = a[i][4 * q], x1 = a[i][4 * q + 1], x2 = a[i][4 * q + 2], x3 = a[i][4 * q + 3];
                asm("v_fma_mix_f32 %0, %0, %4, %5 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mix_f32 %1, %1, %4, %5 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mix_f32 %2, %2, %4, %6 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mix_f32 %3, %3, %4, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mix_f32 %0, %7, 1.0, %0 op_sel_hi:[1,0,0]\n\t"
                    "v_fma_mix_f32 %1, %7, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                    "v_fma_mix_f32 %2, %8, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
                    "v_fma_mix_f32 %3, %8, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
                    : "s"(sc), "v"(raw[i][q][0]), "v"(raw[i][q][1]), "v"(raw[i][q][2]), "v"(raw[i][q][3]));
                a[i][4 * q] = x0; a[i][4 * q + 1] = x1; a[i][4 * q + 2] = x2; a[i][4 * q + 3] = x3;
            }
    }
}

template <bool HL, int EM, int NJ>
__device__ __forceinline__ void gemm_f16x3_epilogue_fast(const GemmHArgs& g, f32x16 (&acc)[2][NJ], int tok0, int oc0, int lane, char* slab2) {
#pragma clang fp contract(off)                  // og_split: hi and lo must see the same rounded value (og_common.h)
    // Slab rows are exactly 128 B; the 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7).  Accumulator-layout accesses (a
    // 16-lane group = 16 consecutive rows at ONE logical chunk, 8 B per lane) then touch 8 different chunks in the even rows (banks
    // 0-31) and 8 in the odd rows (banks 32-63); whole-line accesses (16 lanes = two rows x 8 chunks, 16 B per lane) cover each
    // row's 32 banks once: both conflict-free.  (Round 2 padded the rows to 144 B: the line accesses of rows r, r+1 overlapped in
    // four banks and rows r, r+16 of a column access shared theirs -- 24 % of the LDS cycles of this kernel class were conflicts.)
    constexpr int ROWB = 128;
    constexpr bool HAS_RES = EM == OG_EM_RUNTIME || EM == OG_EM_RES_HL;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned voff = (unsigned)((lane >> 3) * (int)g.ldch * 2 + (lane & 7) * 16);       // 8 rows x 128 B per store instruction
    // line access of instruction `it`: row it*8 + (lane >> 3) -> key ((it*8 + (lane >> 3)) >> 1) & 7 = (4 it + (lane >> 4)) & 7
    auto rd_off = [&](int it) { return (unsigned)((lane >> 3) * ROWB + (((lane & 7) ^ ((4 * it + (lane >> 4)) & 7)) * 16)); };
    const unsigned ckey = (unsigned)((l31 >> 1) & 7);                                          // column access: row l31
    auto col_off = [&](int chunk) { return (unsigned)(l31 * ROWB + ((chunk ^ ckey) * 16) + hi * 8); };
    char* const out_h = reinterpret_cast<char*>(g.Ch);
    char* const out_l = reinterpret_cast<char*>(g.Cl);
    og_u32x4 raw[2][4];
    // (hi, lo) residual, COALESCED: in the accumulator layout a lane owns one token and 4 channels per group, so fetching the residual
    // directly means 8-byte pieces of 64 different 128-byte lines per load instruction -- 16 instructions per slice, ~1000 line
    // accesses at the L1, ~30k cycles per 256 x 256 tile (fc.3 with the residual: 70 us against 54 without at C2).  Instead the
    // slice's residual rows come in like the output rows go out: whole lines, 16 B per lane (8 rows x 128 B per instruction, one
    // slice ahead, in registers), and are turned into the accumulator layout through the two slabs the output transposes use
    // afterwards (LDS serves one wave's instructions in order: no extra synchronisation).
    constexpr bool RES_SLAB = EM == OG_EM_RES_HL && HL;
    og_u32x4 rrow[2][4];
    const char* const res_b = reinterpret_cast<const char*>(g.res_hl);
    const unsigned roff = (unsigned)((lane >> 3) * (int)g.ldrh * 2 + (lane & 7) * 16);
    auto load_res_rows = [&](int t0s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int64_t row = (int64_t)(t0s + it * 8) * g.ldrh;                     // scalar
                rrow[i][it] = *reinterpret_cast<const og_u32x4*>(res_b + (row + og_hl_col(oc0 + i * 32)) * 2 + roff);
            }
    };
    if constexpr (RES_SLAB) load_res_rows(tok0);
    else if constexpr (HAS_RES) gemm_f16x3_load_residual<2, EM == OG_EM_RES_HL>(g, raw, tok0, oc0, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x16 a[2];
        a[0] = acc[0][j]; a[1] = acc[1][j];
        if constexpr (RES_SLAB) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int it = 0; it < 4; ++it) *reinterpret_cast<og_u32x4*>(slab2 + i * EPI_SLAB + it * 8 * ROWB + rd_off(it)) = rrow[i][it];
            if (j + 1 < NJ) load_res_rows(tok0 + (j + 1) * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const char* sb = slab2 + i * EPI_SLAB;
                    const uint2 h2 = *reinterpret_cast<const uint2*>(sb + col_off(q)), l2 = *reinterpret_cast<const uint2*>(sb + col_off(q + 4));
                    raw[i][q] = og_u32x4{h2.x, h2.y, l2.x, l2.y};
                }
        }
        if constexpr (EM == OG_EM_RUNTIME) gemm_f16x3_epilogue_finish<2, false>(g, a, raw, oc0, lane);
        else gemm_f16x3_epilogue_finish_spec<EM>(g, a, raw);
        if constexpr (HAS_RES && !RES_SLAB) { if (j + 1 < NJ) gemm_f16x3_load_residual<2, EM == OG_EM_RES_HL>(g, raw, tok0 + (j + 1) * 32, oc0, lane); }
        // registers -> slabs.  HL: slab i = [32 tok][hi 64 B | lo 64 B] of channel block i; planes: slab 0 = hi, slab 1 = lo of [32 tok][64 ch]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned ha, la, hb, lb;
                og_split4(a[i][4 * q], a[i][4 * q + 1], a[i][4 * q + 2], a[i][4 * q + 3], ha, la, hb, lb);
                if (HL) {
                    char* sb = slab2 + i * EPI_SLAB;
                    *reinterpret_cast<uint2*>(sb + col_off(q)) = make_uint2(ha, hb);
                    *reinterpret_cast<uint2*>(sb + col_off(q + 4)) = make_uint2(la, lb);
                } else {
                    char* d = slab2 + col_off(4 * i + q);
                    *reinterpret_cast<uint2*>(d) = make_uint2(ha, hb);
                    *reinterpret_cast<uint2*>(d + EPI_SLAB) = make_uint2(la, lb);
                }
            }
        // slabs -> whole 128-byte lines
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f16x8 t[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) t[it] = *reinterpret_cast<const f16x8*>(slab2 + i * EPI_SLAB + it * 8 * ROWB + rd_off(it));
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int64_t row = (int64_t)(tok0 + j * 32 + it * 8) * g.ldch;          // scalar
                if (!(OG_GEMM_ABL & 1)) {
                    if (HL) *reinterpret_cast<f16x8*>(out_h + (row + og_hl_col(oc0 + i * 32)) * 2 + voff) = t[it];
                    else *reinterpret_cast<f16x8*>((i == 0 ? out_h : out_l) + (row + oc0) * 2 + voff) = t[it];
                }
            }
        }
    }
}

// Experiment builds only (scripts/build_ablation.sh gemm_trace -DOG_GEMM_TRACE=1): shader-cycle stamps of every wave of every
// block of the 256-tile kernel at the stage hand-overs (read back by og_debug_gemm_trace, scripts/trace_gemm.py).  The stamps
// sit where the LDS queue is already drained; they cost an SMEM round trip each (compare the traced build's time first).
#ifndef OG_GEMM_TRACE
#define OG_GEMM_TRACE 0
#endif
#if OG_GEMM_TRACE
constexpr int OG_GT_BLOCKS = 1024, OG_GT_WORDS = 64;
__device__ unsigned og_gemm_trace_buf[OG_GT_BLOCKS][8][OG_GT_WORDS];
// stamp i of this wave lives in lane i of ONE VGPR (v_writelane): no LDS, no VMEM while the kernel runs
#define OG_GT_PUT(idx_, val_) do { gt_v = lane == (int)(idx_) ? (int)(val_) : gt_v; } while (0)
#define OG_GT(i)                                                                                             \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                          \
        if ((i) < OG_GT_WORDS) OG_GT_PUT((i), t_);                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define OG_GT_FLUSH(nk_)                                                                                     \
    do {                                                                                                     \
        OG_GT_PUT(0, blockIdx.x);                                                                            \
        OG_GT_PUT(1, __builtin_amdgcn_s_getreg(0xF804));                                                     \
        OG_GT_PUT(2, __builtin_amdgcn_s_getreg(0xF814));                                                     \
        OG_GT_PUT(3, gt_rt0);                                                                                \
        OG_GT_PUT(4, (unsigned)__builtin_amdgcn_s_memrealtime());                                            \
        OG_GT_PUT(5, (nk_));                                                                                 \
        if (blockIdx.y == 0 && blockIdx.x < OG_GT_BLOCKS) og_gemm_trace_buf[blockIdx.x][wave][lane] = (unsigned)gt_v;   \
    } while (0)
#else
#define OG_GT(i) do {} while (0)
#endif

// Operands reach LDS by LDS-DMA (global_load_lds, 16 B per lane, no staging registers) into a 2-deep ring; two
// blocks per CU.  A stage is one 32-channel k-slab: in the hl32 row format (og_common.h) that is ONE full
// 128-byte line per row (64 B hi + 64 B lo).  [Measured on MI355X, scripts/probes/l2_bandwidth.hip: fetching
// 64-byte row pieces caps the L2->LDS path at 18 TB/s, full lines reach 35 TB/s; with separate hi/lo planes
// this kernel was bound by exactly that.]
// LDS rows are unpadded (an LDS-DMA writes wave-uniform base + lane*16); bank conflicts are avoided by an XOR
// swizzle applied on the SOURCE address and again on the fragment read: 16-byte chunk c (0-3 hi, 4-7 lo) of
// tile row r lives at chunk position c ^ ((r >> 1) & 7), which puts the 16 lanes of a ds_read_b128 group on 16
// distinct 4-bank slots.  Waits are counted by hand: s_waitcnt vmcnt(N) + raw s_barrier.
typedef __attribute__((address_space(3))) void og_lds_void;
typedef __attribute__((address_space(1))) const void og_glb_void;

// EPI / EM as in the 256-tile kernel below (EPI 0: the generic epilogue; 1 / 2: every tile inside the matrix, split-f16 output only,
// OC == 128): the launcher picks them for the cross-layer launches over one image (q projection, fc.3)
template <int OC, int NS, class RD, int EPI = 0, int EM = 0>
__global__ __launch_bounds__(256, (NS <= 2 ? 2 : 1)) void gemm_nt_f16x3_kernel(GemmHArgs g, int tiles_m, int tiles_n, RD rd) {
    constexpr int TI = OC / 64;                // MFMA tiles per wave along channels
    constexpr int XB = TOK * 128;              // bytes of the token tile per stage (hi|lo rows)
    constexpr int WB = OC * 128;
    constexpr int STAGE = XB + WB;
    constexpr int WPC = OC / 32;               // W pieces per wave per stage (1 KiB = 8 rows x 128 B each)
    constexpr int PPW = 4 + WPC;               // DMA instructions per wave per stage
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE > 4 * EPI_SLAB ? NS * STAGE : 4 * EPI_SLAB];

    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  Token tiles are numbered over ALL problems of a batched launch
    // (virtual tile vt = z * tiles_m + tm) and tile vt runs on XCD vt % 8 with all its channel tiles.  [Round 2 numbered them per
    // problem (grid.y = problem): with 4 token tiles per problem -- the score matrices of 1024-keypoint pairs -- XCDs 4-7 only ever
    // received the padding blocks that exit at once, and the batched score GEMM ran on half the chip: 130 us at C2.]
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int nzk = (rd.B > 0 && !g.ct_rag) ? rd.B : (g.batch > 1 ? g.batch : 1);
    const int vt = (local / tiles_n) * 8 + xcd;
    const int tn = local % tiles_n;
    if (vt >= tiles_m * nzk) return;
    const int z = vt / tiles_m, tm = vt - z * tiles_m;
    if (g.batch > 1 || (rd.B > 0 && !g.ct_rag)) {   // batched problems (the per-pair score matrices)
        if (rd.B > 0 && !g.ct_rag) {           // ragged: problem z = pair z, operands are row ranges of the packed token matrix
            g.M = rd.off0[z + 1] - rd.off0[z];
            g.N = rd.off1[z + 1] - rd.off1[z];
            g.A += (int64_t)rd.off0[z] * g.lda;
            g.B += (int64_t)(rd.off0[rd.B] + rd.off1[z]) * g.ldb;
        } else {
            g.A += z * g.strideA; g.B += z * g.strideB;
        }
        if (g.C32) g.C32 += z * g.strideC32;
    }
    const int t0 = tm * TOK, n0 = tn * OC;
    if (t0 >= g.M || n0 >= g.N) return;          // ragged batches: this problem is smaller than the grid
    if (g.scale_dev) { const float s_ = *g.scale_dev; g.scale = s_; g.inv_scale = 1.f / s_; }      // per-matrix pre-scale chosen at pack time

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave >> 1, wo = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- this wave's DMA pieces: rows [32w, 32w+32) of the token tile, rows [w*OC/4, (w+1)*OC/4) of the W tile ----
    const char* src[PPW];
    {
        const int rl = lane >> 3;                          // row inside the 8-row piece
        const int pc = lane & 7;                           // physical chunk this lane fills
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int rt = wave * 32 + h * 8 + rl;         // tile row
            int row = t0 + rt; if (row >= g.M) row = g.M - 1;
            if (OG_GEMM_ABL & 2) row = rt;
            src[h] = reinterpret_cast<const char*>(g.A + (int64_t)row * g.lda) + (pc ^ ((rt >> 1) & 7)) * 16;
        }
#pragma unroll
        for (int h = 0; h < WPC; ++h) {
            const int rt = wave * (OC / 4) + h * 8 + rl;
            int row = n0 + rt; if (row >= g.N) row = g.N - 1;
            src[4 + h] = reinterpret_cast<const char*>(g.B + (int64_t)row * g.ldb) + (pc ^ ((rt >> 1) & 7)) * 16;
        }
    }
    auto issue_stage = [&](int kt) {
        char* sbase = smem + (kt % NS) * STAGE;
        const int64_t koff = (int64_t)kt * 128;
#pragma unroll
        for (int h = 0; h < 4; ++h)
            __builtin_amdgcn_global_load_lds((og_glb_void*)(src[h] + koff), (og_lds_void*)(sbase + (wave * 32 + h * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int h = 0; h < WPC; ++h)
            __builtin_amdgcn_global_load_lds((og_glb_void*)(src[4 + h] + koff),
                                             (og_lds_void*)(sbase + XB + (wave * (OC / 4) + h * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[TI][2];

    const int swz = (l31 >> 1) & 7;                    // tile rows differ from l31 by multiples of 16 only
    const int x_row = (wt * 64 + l31) * 128;           // byte offset of this lane's token row (j = 0)
    const int w_row = XB + (wo * (OC / 2) + l31) * 128;

    const int nk = g.K / BKH;
    const int pre = nk < NS - 1 ? nk : NS - 1;
    for (int kt = 0; kt < pre; ++kt) issue_stage(kt);
    gemm_f16x3_acc_init<TI, 2>(g, acc, n0 + wo * (OC / 2), lane);     // bias / scale, in the shadow of the first DMA (the vmcnt
                                                                       // waits below are counted from the newest DMA: they also
                                                                       // cover these older loads)
    for (int kt = 0; kt < nk; ++kt) {
        const int issued = (kt + NS - 1 < nk) ? kt + NS - 1 : nk;
        const int ahead = issued - kt - 1;              // later stages that may stay in flight
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue_stage(kt + NS - 1);   // slot (kt-1)%NS: every wave is past its reads of k-tile kt-1
        const char* sb = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < BKH / 16; ++ks) {
            const int ch = ((2 * ks + hi) ^ swz) * 16;  // hi chunk; its lo partner is chunk + 4 -> byte offset ^ 64
            f16x8 wh[TI], wl[TI], xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                wh[i] = *reinterpret_cast<const f16x8*>(sb + w_row + i * 32 * 128 + ch);
                wl[i] = *reinterpret_cast<const f16x8*>(sb + w_row + i * 32 * 128 + (ch ^ 64));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xh[j] = *reinterpret_cast<const f16x8*>(sb + x_row + j * 32 * 128 + ch);
                xl[j] = *reinterpret_cast<const f16x8*>(sb + x_row + j * 32 * 128 + (ch ^ 64));
            }
#if OG_GEMM_ABL & 4
#pragma unroll
            for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(wh[i]), "v"(wl[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(xh[j]), "v"(xl[j]));
#else
            // pass-major order: consecutive MFMAs write different accumulators
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
#endif
        }
    }

    __builtin_amdgcn_s_barrier();      // every wave is past its last fragment reads: the ring is free
    if constexpr (EPI != 0) {
        static_assert(TI == 2 && 4 * 2 * EPI_SLAB <= NS * STAGE, "fast epilogue: 64-channel wave tiles, two slabs per wave");
        gemm_f16x3_epilogue_fast<EPI == 1, EM, 2>(g, acc, t0 + wt * 64, n0 + wo * (OC / 2), lane, smem + wave * 2 * EPI_SLAB);
    } else {
        // two 32-token slices; the residual of slice 1 is fetched (into the same registers) as soon as slice 0 has consumed
        // its own, and is in flight while slice 0 is split, transposed and stored
        og_u32x4 raw[TI][4];
        const int tok0 = t0 + wt * 64, oc0 = n0 + wo * (OC / 2);
        char* slab = smem + wave * EPI_SLAB;
        gemm_f16x3_load_residual<TI>(g, raw, tok0, oc0, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 a[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = acc[i][j];
            gemm_f16x3_epilogue_finish<TI, true>(g, a, raw, oc0, lane);
            if (j + 1 < 2) gemm_f16x3_load_residual<TI>(g, raw, tok0 + (j + 1) * 32, oc0, lane);
            gemm_f16x3_epilogue_store<TI, true>(g, a, tok0 + j * 32, oc0, lane, slab, rd);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Large-tile variant for the big GEMMs of the GNN: 256 tokens x 256 channels per block, 8 waves as 2 (tokens) x 4
// (channels), wave tile 128 x 64 = 2 x 4 MFMA tiles (128 accumulator registers), one block per CU.
// Why: the 128 x 128 kernel above is bound by its operand traffic, not by the matrix pipe (ablation: 62 us of
// global->LDS->register traffic + 41 us of MFMA for fc.0 at C2, barely overlapped).  Here a k-step
// moves the same 64 KB through LDS for 4x the MFMA work of a 128 x 128 step (half the L2->LDS bytes and 3/4 of
// the LDS fragment reads per MFMA), and a k-step holds 2 x 48 MFMAs per SIMD (~1.3 us), which covers the latency
// of the one-stage-ahead LDS-DMA prefetch.
constexpr int BIG = 256;

template <class RD>
__global__ __launch_bounds__(512) void gemm_nt_f16x3_big_kernel(GemmHArgs g, int tiles_m, int tiles_n, RD rd) {
    constexpr int NS = 2;
    constexpr int XB = BIG * 128;              // bytes of the token tile per stage
    constexpr int STAGE = 2 * XB;              // token tile + weight tile
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
    static_assert(NS * STAGE >= 8 * EPI_SLAB, "epilogue slabs must fit in the ring");

    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  Token tiles are numbered over ALL problems of a batched launch
    // (virtual tile vt = z * tiles_m + tm) and tile vt runs on XCD vt % 8 with all its channel tiles.  [Round 2 numbered them per
    // problem (grid.y = problem): with 4 token tiles per problem -- the score matrices of 1024-keypoint pairs -- XCDs 4-7 only ever
    // received the padding blocks that exit at once, and the batched score GEMM ran on half the chip: 130 us at C2.]
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int nzk = (rd.B > 0 && !g.ct_rag) ? rd.B : (g.batch > 1 ? g.batch : 1);
    const int vt = (local / tiles_n) * 8 + xcd;
    const int tn = local % tiles_n;
    if (vt >= tiles_m * nzk) return;
    const int z = vt / tiles_m, tm = vt - z * tiles_m;
    if (g.batch > 1 || (rd.B > 0 && !g.ct_rag)) {   // batched problems (the per-pair score matrices)
        if (rd.B > 0 && !g.ct_rag) {           // ragged: problem z = pair z, operands are row ranges of the packed token matrix
            g.M = rd.off0[z + 1] - rd.off0[z];
            g.N = rd.off1[z + 1] - rd.off1[z];
            g.A += (int64_t)rd.off0[z] * g.lda;
            g.B += (int64_t)(rd.off0[rd.B] + rd.off1[z]) * g.ldb;
        } else {
            g.A += z * g.strideA; g.B += z * g.strideB;
        }
        if (g.C32) g.C32 += z * g.strideC32;
    }
    const int t0 = tm * BIG, n0 = tn * BIG;
    if (t0 >= g.M || n0 >= g.N) return;          // ragged batches: this problem is smaller than the grid
    if (g.scale_dev) { const float s_ = *g.scale_dev; g.scale = s_; g.inv_scale = 1.f / s_; }      // per-matrix pre-scale chosen at pack time

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave >> 2, wo = wave & 3;
#if OG_GEMM_TRACE
    int gt_v = 0;
    const unsigned gt_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    OG_GT(6);
#endif
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- DMA pieces (1 KiB = 8 rows x 128 B): wave w fills rows [32w, 32w+32) of the token tile and of the W tile ----
    const char* src[8];
    {
        const int rl = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int rt = wave * 32 + h * 8 + rl;
            const int sw = (pc ^ ((rt >> 1) & 7)) * 16;
            int row = t0 + rt; if (row >= g.M) row = g.M - 1;
            src[h] = reinterpret_cast<const char*>(g.A + (int64_t)row * g.lda) + sw;
            row = n0 + rt; if (row >= g.N) row = g.N - 1;
            src[4 + h] = reinterpret_cast<const char*>(g.B + (int64_t)row * g.ldb) + sw;
        }
    }
    auto issue_stage = [&](int kt) {
        char* sbase = smem + (kt % NS) * STAGE + wave * 32 * 128;
        const int64_t koff = (int64_t)kt * 128;
#pragma unroll
        for (int h = 0; h < 4; ++h)
            __builtin_amdgcn_global_load_lds((og_glb_void*)(src[h] + koff), (og_lds_void*)(sbase + h * 8 * 128), 16, 0, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h)
            __builtin_amdgcn_global_load_lds((og_glb_void*)(src[4 + h] + koff), (og_lds_void*)(sbase + XB + h * 8 * 128), 16, 0, 0);
    };

    f32x16 acc[2][4];

    const int swz = (l31 >> 1) & 7;
    const int x_row = (wt * 128 + l31) * 128;
    const int w_row = XB + (wo * 64 + l31) * 128;

    // Fragment pipeline.  A k-step is 8 groups (2 k16 halves x 4 token tiles) of 6 MFMAs; the LDS reads of group
    // g+1 are issued BEFORE the MFMAs of group g (double-buffered fragment registers), so the matrix pipe never waits
    // for a full LDS round trip.  The stage hand-over (wait for the DMA of stage kt+1, barrier, DMA of stage kt+2,
    // first fragments of stage kt+1) sits in front of the LAST group of stage kt, whose MFMAs cover it.
    // The fragment reads and their waits are inline asm: hipcc's own waitcnt insertion drains the LDS queue
    // (lgkmcnt(0)) in front of every second group here, which serialises the read of group g+1 with the MFMAs of
    // group g.  LDS returns data in order, so "all but the N newest reads" is what s_waitcnt lgkmcnt(N) waits for;
    // the "+v" operands tie each wait to the registers it releases.
    f16x8 wh[2][2], wl[2][2], xh[2], xl[2];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto lds_read = [&](f16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
    auto read_w = [&](unsigned sb, int ks, int buf) {
        const unsigned a = sb + w_row + (((2 * ks + hi) ^ swz) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_read(wh[buf][i], a + i * 32 * 128);
            lds_read(wl[buf][i], (a + i * 32 * 128) ^ 64);
        }
    };
    auto read_x = [&](unsigned sb, int ks, int j, int buf) {
        const unsigned a = sb + x_row + j * 32 * 128 + (((2 * ks + hi) ^ swz) * 16);
        lds_read(xh[buf], a);
        lds_read(xl[buf], a ^ 64);
    };
    // frees the fragments of (w buffer wb, x buffer xb) while `newer` younger reads may stay in flight
    auto wait_frags = [&](int wb, int xb, int newer) {
        if (newer == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
        else if (newer == 2)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
        else
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
    };

    const int nk = g.K / BKH;
    issue_stage(0);
    gemm_f16x3_acc_init<2, 4>(g, acc, n0 + wo * 64, lane);     // bias / scale, in the shadow of the first DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    OG_GT(7);
    if (nk > 1) issue_stage(1);
    read_x(lds0, 0, 0, 0);
    read_w(lds0, 0, 0);
    if (OG_GEMM_ABL & 16) { read_x(lds0, 0, 1, 1); read_w(lds0, 1, 1); }
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned sb = lds0 + (kt % NS) * STAGE;
        const unsigned sbn = lds0 + ((kt + 1) % NS) * STAGE;
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int ks = grp >> 2, j = grp & 3;
            if (grp < 7) {
                const int ks1 = (grp + 1) >> 2, j1 = (grp + 1) & 3;
                if (!(OG_GEMM_ABL & 16)) {
                    read_x(sb, ks1, j1, (grp + 1) & 1);
                    if (j1 == 0) read_w(sb, ks1, ks1 & 1);
                    wait_frags(ks & 1, grp & 1, j1 == 0 ? 6 : 2);
                }
            } else {
                wait_frags(ks & 1, grp & 1, 0);                                // all my reads of stage kt are done
                if (kt + 1 < nk) {
                    OG_GT(8 + 3 * kt);
                    // (the "memory" clobber also pins the LDS-DMA issue below behind the volatile fragment reads above)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my DMA pieces of stage kt+1 landed
                    OG_GT(9 + 3 * kt);
                    __builtin_amdgcn_s_barrier();
                    OG_GT(10 + 3 * kt);
                    if (kt + 2 < nk && !(OG_GEMM_ABL & 8)) issue_stage(kt + 2);   // overwrites the slot of stage kt
                    if (!(OG_GEMM_ABL & 16)) { read_x(sbn, 0, 0, 0); read_w(sbn, 0, 0); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#if OG_GEMM_ABL & 4
            asm volatile("" ::"v"(wh[ks & 1][0]), "v"(wl[ks & 1][0]), "v"(wh[ks & 1][1]), "v"(wl[ks & 1][1]), "v"(xh[grp & 1]), "v"(xl[grp & 1]));
#else
            // pass-major: consecutive MFMAs write different accumulators
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 1][0], xh[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 1][1], xh[grp & 1], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][0], xl[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][1], xl[grp & 1], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][0], xh[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][1], xh[grp & 1], acc[1][j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if OG_GEMM_TRACE
    const int gt_e = 8 + 3 * (nk - 1);
    OG_GT(gt_e);
#endif

    __builtin_amdgcn_s_barrier();      // the ring is free: per-wave epilogue slabs
    {
        // four 32-token slices; the residual of slice s + 1 is fetched (into the same registers) as soon as slice s has
        // consumed its own, and is in flight while slice s is split, transposed and stored
        og_u32x4 raw[2][4];
        const int tok0 = t0 + wt * 128, oc0 = n0 + wo * 64;
        char* slab = smem + wave * EPI_SLAB;
        gemm_f16x3_load_residual<2>(g, raw, tok0, oc0, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x16 a[2];
            a[0] = acc[0][j]; a[1] = acc[1][j];
            gemm_f16x3_epilogue_finish<2, false>(g, a, raw, oc0, lane);
            if (j + 1 < 4) gemm_f16x3_load_residual<2>(g, raw, tok0 + (j + 1) * 32, oc0, lane);
            gemm_f16x3_epilogue_store<2, false>(g, a, tok0 + j * 32, oc0, lane, slab, rd);
        }
    }
#if OG_GEMM_TRACE
    OG_GT(gt_e + 1);                                   // all stores issued (not necessarily complete)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OG_GT(gt_e + 2);                                   // all stores acknowledged
    OG_GT_FLUSH(nk);
#endif
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 tile, second generation.  What the per-wave cycle trace of the kernel above showed at C2
// (scripts/trace_gemm.py, profiles/r02_gemm_trace_*.log): a k-stage takes ~4650 cycles against 3072 of matrix-pipe
// time, and the gap is the stage hand-over -- every wave meets at the barrier and then issues its 8 LDS-DMA pieces
// (60-185 cycles of issue each, MI355X_MICROARCH.md) back to back while the matrix pipe of all four SIMDs runs dry; the
// DMA itself is never waited for (156 cycles = the stamp).  Here
//   * the DMA pieces of the NEXT stages are issued two at a time behind the first MFMAs of groups 0-3 of a stage, in
//     the shadow of the matrix pipe; the hand-over is wait + barrier + first fragment reads only;
//   * the token operand (HBM / Infinity Cache, long latency) has a 3-slot ring and is fetched TWO stages ahead, the
//     weight operand (L2-resident) a 2-slot ring, one stage ahead: 3 x 32 KB + 2 x 32 KB = the whole 160 KB LDS.
//     Counted waits: LDS-DMA completes in issue order, every stage issues W(kt+1) first and X(kt+2) second, so at the
//     end of stage kt `s_waitcnt vmcnt(4)` = "everything but the four X(kt+2) pieces has landed" = X(kt+1), W(kt+1).
// One problem per launch (no batch, no ragged descriptor): the per-pair score GEMM stays on the kernel above.
// EPI: 0 = generic epilogue, 1 = fast hl32 rows, 2 = fast planes (every tile inside the matrix; chosen by the launcher -- one
// epilogue per instantiation keeps the kernel inside its 256 registers)
template <int EPI, int EM>
__global__ __launch_bounds__(512) void gemm_nt_f16x3_big2_kernel(GemmHArgs g, int tiles_m, int tiles_n) {
    constexpr int XS = BIG * 128;              // one stage of one operand: 256 rows x 128 B (hi 64 B | lo 64 B)
    constexpr int WOFF = 3 * XS;
    __shared__ __attribute__((aligned(16))) char smem[5 * XS];
    static_assert(5 * XS == 163840 && 5 * XS >= 8 * EPI_SLAB, "the rings take the whole LDS; the epilogue slabs alias them");

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int tm = (local / tiles_n) * 8 + xcd;       // all channel tiles of a token tile on one XCD
    const int tn = local % tiles_n;
    if (tm >= tiles_m) return;
    const int t0 = tm * BIG, n0 = tn * BIG;
    if (t0 >= g.M || n0 >= g.N) return;
    if (t0 < g.split_row && n0 >= g.split_n) return;     // row-split launch: the first row range has fewer columns (exits at once: the
                                                         // dispatcher back-fills the CU with the next block)
    if (g.scale_dev) { const float s_ = *g.scale_dev; g.scale = s_; g.inv_scale = 1.f / s_; }      // per-matrix pre-scale chosen at pack time

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave >> 2, wo = wave & 3;
#if OG_GEMM_TRACE
    int gt_v = 0;
    const unsigned gt_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    OG_GT(6);
#endif
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- DMA pieces (1 KiB = 8 rows x 128 B): wave w fills rows [32w, 32w+32) of the token tile and of the W tile ----
    // per-lane 32-bit byte offsets from the block's (scalar) tile bases: the DMA address of a stage is base + kt * 128 (SALU) + offset,
    // no vector arithmetic per stage
    unsigned soff[8];
    const char* const baseA = reinterpret_cast<const char*>(g.A + (int64_t)((OG_GEMM_ABL & 2) ? 0 : t0) * g.lda);   // ablation 2: token tile 0 for all
    const char* const baseB = reinterpret_cast<const char*>(g.B + (int64_t)n0 * g.ldb);
    {
        const int rl = lane >> 3, pc = lane & 7;
        const int lastA = g.M - 1 - t0, lastB = g.N - 1 - n0;           // rows past the matrix are clamped (computed, never stored)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int rt = wave * 32 + h * 8 + rl;
            const unsigned sw = (unsigned)(pc ^ ((rt >> 1) & 7)) * 16u;
            soff[h] = (unsigned)((rt < lastA ? rt : lastA) * (int)g.lda * 2) + sw;
            soff[4 + h] = (unsigned)((rt < lastB ? rt : lastB) * (int)g.ldb * 2) + sw;
            asm volatile("" : "+v"(soff[h]), "+v"(soff[4 + h]));        // opaque: computed once, never rematerialised inside the stage loop
        }
    }
    auto scalar_ptr = [](const char* p) {        // wave-uniform by construction: pinned to an SGPR pair so that the DMA takes the
        const uint64_t v = (uint64_t)(uintptr_t)p;   // (scalar base + 32-bit lane offset) addressing form
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    // piece h of the token / weight operand of stage kt
    auto lane_off = [&](int i) {                  // re-launder per use: keeps the zero-extension next to the address add (else the
        unsigned o = soff[i];                      // hoisted 64-bit copy hides that the lane part is 32 bits wide)
        asm volatile("" : "+v"(o));
        return o;
    };
    auto issue_x = [&](int kt, int h) {
        __builtin_amdgcn_global_load_lds((og_glb_void*)(scalar_ptr(baseA + (int64_t)kt * 128) + lane_off(h)),
                                         (og_lds_void*)(smem + (kt % 3) * XS + (wave * 32 + h * 8) * 128), 16, 0, 0);
    };
    auto issue_w = [&](int kt, int h) {
        __builtin_amdgcn_global_load_lds((og_glb_void*)(scalar_ptr(baseB + (int64_t)kt * 128) + lane_off(4 + h)),
                                         (og_lds_void*)(smem + WOFF + (kt & 1) * XS + (wave * 32 + h * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[2][4];
    const int swz = (l31 >> 1) & 7;
    const int x_row = (wt * 128 + l31) * 128;
    const int w_row = (wo * 64 + l31) * 128;

    // fragment pipeline as in the kernel above (hand-counted lgkmcnt: LDS-DMA does not touch that counter)
    f16x8 wh[2][2], wl[2][2], xh[2], xl[2];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // Fragment addresses: the lane part (row, swizzled chunk of k-step ks, hi or lo half = chunk ^ 4) is loop invariant -- four
    // registers per operand; a stage adds its ring-slot base once (8 VALU per stage), token / channel block offsets are immediates.
    unsigned xk[2][2], wk[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const unsigned c = (unsigned)(((2 * ks + hi) ^ swz) * 16) ^ (hl ? 64u : 0u);
            xk[ks][hl] = (unsigned)x_row + c;
            wk[ks][hl] = (unsigned)w_row + c;
        }
    auto lds_read = [&](f16x8& dst, unsigned addr, int imm) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); };
    unsigned xa[2][2], wa[2][2];                   // the addresses of the stage being read (and, at the hand-over, of the next one)
    auto set_x = [&](unsigned xb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { xa[ks][0] = xb + xk[ks][0]; xa[ks][1] = xb + xk[ks][1]; }
    };
    auto set_w = [&](unsigned wb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { wa[ks][0] = wb + wk[ks][0]; wa[ks][1] = wb + wk[ks][1]; }
    };
    auto read_w = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_read(wh[buf][i], wa[ks][0], i * 32 * 128);
            lds_read(wl[buf][i], wa[ks][1], i * 32 * 128);
        }
    };
    auto read_x = [&](int ks, int j, int buf) {
        lds_read(xh[buf], xa[ks][0], j * 32 * 128);
        lds_read(xl[buf], xa[ks][1], j * 32 * 128);
    };
    auto wait_frags = [&](int wb, int xb, int newer) {
        if (newer == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
        else if (newer == 2)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
        else
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wh[wb][0]), "+v"(wl[wb][0]), "+v"(wh[wb][1]), "+v"(wl[wb][1]), "+v"(xh[xb]), "+v"(xl[xb]));
    };

    const int nk = g.K / BKH;
    // prologue: X(0), W(0), X(1); the bias (acc init) in their shadow
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_x(0, h);
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_w(0, h);
    if (nk > 1) {
#pragma unroll
        for (int h = 0; h < 4; ++h) issue_x(1, h);
    }
    gemm_f16x3_acc_init<2, 4>(g, acc, n0 + wo * 64, lane);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // X(0), W(0) landed; X(1) may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    OG_GT(7);
    set_x(lds0); set_w(lds0 + WOFF);
    read_x(0, 0, 0);
    read_w(0, 0);
    int xs = 0;                                     // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int xs1 = xs == 2 ? 0 : xs + 1;
        const unsigned xbn = lds0 + xs1 * XS, wbn = lds0 + WOFF + ((kt + 1) & 1) * XS;
        const bool iw = kt + 1 < nk, ix = kt + 2 < nk;      // W(kt+1) / X(kt+2) exist
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int ks = grp >> 2, j = grp & 3;
            if (grp < 7) {
                const int ks1 = (grp + 1) >> 2, j1 = (grp + 1) & 3;
                if (!(OG_GEMM_ABL & 16)) {                  // ablation 16: no fragment reads after the first
                    read_x(ks1, j1, (grp + 1) & 1);
                    if (j1 == 0) read_w(ks1, ks1 & 1);
                }
                wait_frags(ks & 1, grp & 1, j1 == 0 ? 6 : 2);
            } else {
                wait_frags(ks & 1, grp & 1, 0);                                // all my reads of stage kt are done
                if (iw) {
                    OG_GT(8 + 3 * kt);
                    // my pieces of X(kt+1), W(kt+1) landed (everything but the four newest = X(kt+2))
                    if (ix) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    OG_GT(9 + 3 * kt);
                    __builtin_amdgcn_s_barrier();
                    OG_GT(10 + 3 * kt);
                    set_x(xbn); set_w(wbn);              // my reads of stage kt are complete: the address registers move on
                    if (!(OG_GEMM_ABL & 16)) {
                        read_x(0, 0, 0);
                        read_w(0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // pass-major: consecutive MFMAs write different accumulators
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 1][0], xh[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 1][1], xh[grp & 1], acc[1][j], 0, 0, 0);
            if (grp < 4) {
                // two DMA pieces in the shadow of the matrix pipe: W(kt+1) in groups 0-1 (its slot held W(kt-1): free since
                // the barrier that ended stage kt-1), then X(kt+2) in groups 2-3 (its slot held X(kt-1))
                __builtin_amdgcn_sched_barrier(0);
                if (grp < 2) {
                    if (iw && !(OG_GEMM_ABL & 8)) { issue_w(kt + 1, 2 * grp); issue_w(kt + 1, 2 * grp + 1); }
                } else {
                    if (ix && !(OG_GEMM_ABL & 8)) { issue_x(kt + 2, 2 * (grp - 2)); issue_x(kt + 2, 2 * (grp - 2) + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][0], xl[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][1], xl[grp & 1], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][0], xh[grp & 1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 1][1], xh[grp & 1], acc[1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        xs = xs1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if OG_GEMM_TRACE
    const int gt_e = 8 + 3 * (nk - 1);
    OG_GT(gt_e);
#endif

    __builtin_amdgcn_s_barrier();      // every wave is past its last fragment reads, no DMA is in flight: the rings are free
    {
        const int tok0 = t0 + wt * 128, oc0 = n0 + wo * 64;
        if constexpr (EPI == 1) {
            gemm_f16x3_epilogue_fast<true, EM, 4>(g, acc, tok0, oc0, lane, smem + wave * 2 * EPI_SLAB);
        } else if constexpr (EPI == 2) {
            gemm_f16x3_epilogue_fast<false, EM, 4>(g, acc, tok0, oc0, lane, smem + wave * 2 * EPI_SLAB);
        } else {
            og_u32x4 raw[2][4];
            char* slab = smem + wave * EPI_SLAB;
            const RaggedNone no_rd{};               // FULL = false: the channel-first copy (the only user of the descriptor) is compiled out
            gemm_f16x3_load_residual<2>(g, raw, tok0, oc0, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 a[2];
                a[0] = acc[0][j]; a[1] = acc[1][j];
                gemm_f16x3_epilogue_finish<2, false>(g, a, raw, oc0, lane);
                if (j + 1 < 4) gemm_f16x3_load_residual<2>(g, raw, tok0 + (j + 1) * 32, oc0, lane);
                gemm_f16x3_epilogue_store<2, false>(g, a, tok0 + j * 32, oc0, lane, slab, no_rd);
            }
        }
    }
#if OG_GEMM_TRACE
    OG_GT(gt_e + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OG_GT(gt_e + 2);
    OG_GT_FLUSH(nk);
#endif
}

// x -> (hi, lo) planes, elementwise (test helper and weight/activation conversion outside the GEMMs)
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int64_t n4, _Float16* __restrict__ h,
                                                        _Float16* __restrict__ l) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    f16x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        og_split(v[e], h, l);
        vh[e] = h; vl[e] = l;
    }
    *reinterpret_cast<f16x4*>(h + 4 * i) = vh;
    *reinterpret_cast<f16x4*>(l + 4 * i) = vl;
}

// x [rows][cols] (row stride ldx) fp32 -> (hi, lo) planes with row stride ldo; columns [0, scale_cols) are multiplied by s1, then by s2, first
// (two roundings, like the two tensor multiplications they replace: the attention scale dh^-1/2 and the log2(e) of the kernel's base-2 softmax).
// The training step's q | k | v matrix goes to the attention kernel's operand format in ONE launch, no slices copied out first.
__global__ __launch_bounds__(256) void split_f16_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, int scale_cols,
                                                             float s1, float s2, _Float16* __restrict__ h, _Float16* __restrict__ l, int64_t ldo) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = cols / 4;
    if (i >= rows * c4) return;
    const int64_t r = i / c4; const int c = (int)(i % c4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    if (c < scale_cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] * s1) * s2;
    }
    f16x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 hh, ll;
        og_split(v[e], hh, ll);
        vh[e] = hh; vl[e] = ll;
    }
    *reinterpret_cast<f16x4*>(h + r * ldo + c) = vh;
    *reinterpret_cast<f16x4*>(l + r * ldo + c) = vl;
}

// out = float(hi) + float(lo): the attention kernel's output planes back to fp32 in one launch
__global__ __launch_bounds__(256) void merge_f16_kernel(const _Float16* __restrict__ h, const _Float16* __restrict__ l, int64_t n4,
                                                        float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f16x4 vh = *reinterpret_cast<const f16x4*>(h + 4 * i), vl = *reinterpret_cast<const f16x4*>(l + 4 * i);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)vh[e] + (float)vl[e];
    *reinterpret_cast<f32x4*>(out + 4 * i) = v;
}

// x [rows][cols] fp32 -> hl32 rows (test helper / conversions outside the GEMMs)
__global__ __launch_bounds__(256) void split_f16_hl_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ldx,
                                                           _Float16* __restrict__ out, int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = cols / 4;
    if (i >= rows * c4) return;
    const int64_t r = i / c4; const int c = (int)(i % c4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    f16x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        og_split(v[e], h, l);
        vh[e] = h; vl[e] = l;
    }
    _Float16* d = out + r * ldo + og_hl_col(c);
    *reinterpret_cast<f16x4*>(d) = vh;
    *reinterpret_cast<f16x4*>(d + 32) = vl;
}

// hl32 rows -> fp32 [rows][cols] (inverse of split_f16_hl_kernel: hi + lo; the stage taps of og_forward_tap)
__global__ __launch_bounds__(256) void merge_f16_hl_kernel(const _Float16* __restrict__ in, int64_t rows, int cols, int64_t ldi,
                                                           float* __restrict__ out, int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = cols / 4;
    if (i >= rows * c4) return;
    const int64_t r = i / c4; const int c = (int)(i % c4) * 4;
    const _Float16* d = in + r * ldi + og_hl_col(c);
    const f16x4 vh = *reinterpret_cast<const f16x4*>(d), vl = *reinterpret_cast<const f16x4*>(d + 32);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)vh[e] + (float)vl[e];
    *reinterpret_cast<f32x4*>(out + r * ldo + c) = v;
}

}  // namespace

int og_launch_merge_f16_hl(const void* in, int64_t rows, int cols, int64_t ldi, float* out, int64_t ldo, hipStream_t stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return OG_E_INVALID;
    if ((cols & 31) || (ldi & 3) || (ldo & 3) || ldi < 2 * (int64_t)cols || ((uintptr_t)in & 7) || ((uintptr_t)out & 15)) return OG_E_ALIGN;
    const int64_t n4 = rows * (cols / 4);
    hipLaunchKernelGGL(merge_f16_hl_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const _Float16*)in, rows, cols, ldi, out, ldo);
    return og_launch_status();
}

// A row-split launch (GemmHArgs::split_row) must end up on the second-generation 256-tile kernel: same conditions as the launcher's.
bool og_gemm_f16x3_row_split_ok(const GemmHArgs& a) {
    if (a.split_row <= 0 || a.split_row >= a.M || a.split_row % BIG || a.split_n <= 0 || a.split_n >= a.N || a.split_n % BIG) return false;
    if (a.M % BIG || a.N % BIG || a.batch > 1 || a.rag || a.alpha || a.Ct || a.C32 || !a.Ch) return false;
    static const int force = [] { const char* e = getenv("OG_GEMM_TILE"); return e ? atoi(e) : 0; }();
    static const bool big2 = [] { const char* e = getenv("OG_GEMM_BIG2"); return !e || atoi(e) != 0; }();
    static const bool on = [] { const char* e = getenv("OG_GEMM_ROW_SPLIT"); return !e || atoi(e) != 0; }();      // experiments: 0 = two launches
    const int64_t blocks = (int64_t)(a.split_row / BIG) * (a.split_n / BIG) + (int64_t)((a.M - a.split_row) / BIG) * (a.N / BIG);
    return on && big2 && force != 128 && blocks >= 192;
}

int og_launch_gemm_f16x3(const GemmHArgs& a, hipStream_t stream) {
    if (!a.A || !a.B || a.M <= 0 || a.N <= 0 || a.K <= 0) return OG_E_INVALID;
    if (a.split_row && !og_gemm_f16x3_row_split_ok(a)) return OG_E_INVALID;
    if (!a.C32 && !a.Ch && !a.Ct) return OG_E_INVALID;
    if (a.Ch && !a.c_hl && !a.Cl) return OG_E_INVALID;
    if ((a.K % BKH) || (a.lda & 7) || (a.ldb & 7) || a.lda < 2 * (int64_t)a.K || a.ldb < 2 * (int64_t)a.K) return OG_E_ALIGN;
    // N % 4 != 0 is allowed for a bare fp32 output whose rows are padded to a multiple of 4 (the score matrix): the last
    // float4 of a row then spills into the padding
    if ((a.N & 3) && (a.bias || a.res || a.res_hl || a.Ch || a.Ct || !a.C32 || a.ldc < (a.N + 3) / 4 * 4)) return OG_E_ALIGN;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return OG_E_ALIGN;
    if (a.C32 && (((uintptr_t)a.C32 & 15) || (a.ldc & 3))) return OG_E_ALIGN;
    if (a.Ch && a.c_hl && (((uintptr_t)a.Ch & 15) || (a.ldch & 7) || (a.N & 31))) return OG_E_ALIGN;
    if (a.Ch && !a.c_hl && (((uintptr_t)a.Ch & 7) || ((uintptr_t)a.Cl & 7) || (a.ldch & 3))) return OG_E_ALIGN;
    if (a.res && (((uintptr_t)a.res & 15) || (a.ldr & 3) || (a.N & 31))) return OG_E_ALIGN;
    if (a.res_hl && (a.res || ((uintptr_t)a.res_hl & 15) || (a.ldrh & 7) || (a.N & 31))) return OG_E_ALIGN;
    if (a.bias && ((uintptr_t)a.bias & 15)) return OG_E_ALIGN;
    if (a.alpha && (!a.res || ((uintptr_t)a.alpha & 15))) return OG_E_INVALID;
    if (!(a.scale != 0.f) || !std::isfinite(a.scale)) return OG_E_INVALID;     // the bias enters the accumulators as bias / scale
    if (a.Ct && ((!a.ct_rag && a.ct_rows <= 0) || a.batch > 1)) return OG_E_INVALID;
    if (a.ct_rag && (!a.Ct || !a.rag || a.batch > 1 || a.ct_rag < 0 || a.ct_rag > 2)) return OG_E_INVALID;
    const int nz = a.batch > 1 ? a.batch : 1;
    if (nz > 1 && (a.Ch || a.res || a.res_hl || (a.strideA & 7) || (a.strideB & 7) || (a.strideC32 & 3))) return OG_E_INVALID;
    GemmHArgs g = a;
    g.inv_scale = (float)(1.0 / (double)a.scale);
    if (a.rag && !a.ct_rag && nz != a.rag->B) return OG_E_INVALID;
    g.rag = nullptr;
    static const int force = [] { const char* e = getenv("OG_GEMM_TILE"); return e ? atoi(e) : 0; }();   // experiments: 128 / 256
    // The per-pair descriptor travels by value in the kernarg segment ONLY for ragged launches (og_common.h: RaggedNone).
    auto launch = [&](auto rd) -> int {
        using RD = decltype(rd);
        {   // large tiles when they still give (nearly) every CU a block
            const int tiles_m = (a.M + BIG - 1) / BIG, tiles_n = (a.N + BIG - 1) / BIG;
            const bool fits = (a.N % BIG == 0 || nz > 1) && (int64_t)tiles_m * tiles_n * nz >= 192 && !a.alpha && !a.Ct;
            if (force == 256 ? (a.N >= BIG && !a.alpha && !a.Ct) : (fits && force != 128)) {
                const int tiles_m8 = (tiles_m + 7) / 8 * 8;
                static const bool big2 = [] { const char* e = getenv("OG_GEMM_BIG2"); return !e || atoi(e) != 0; }();   // experiments: 0 = first generation
                if (big2 && nz == 1 && !a.rag) {
                    const bool whole = g.M % 256 == 0 && g.N % 256 == 0 && g.Ch && !g.C32;      // every tile inside, split-f16 output only
                    static const bool fast_epi = [] { const char* e = getenv("OG_GEMM_FAST_EPI"); return !e || atoi(e) != 0; }();   // experiments
                    const dim3 grid2(tiles_m8 * tiles_n), block2(512);
                    // the epilogue arithmetic as a compile-time mode where the launch is one of the GNN's four forms (OG_GEMM_SPEC_EPI=0: experiments)
                    static const bool spec_epi = [] { const char* e = getenv("OG_GEMM_SPEC_EPI"); return !e || atoi(e) != 0; }();
                    int em = OG_EM_RUNTIME;
                    if (spec_epi && !g.alpha) {
                        if (!g.res && !g.res_hl) em = g.relu ? OG_EM_RELU : OG_EM_NONE;
                        else if (!g.relu && g.res_hl) em = OG_EM_RES_HL;      // (an fp32 residual -- one launch per step -- stays on the run-time form:
                                                                              //  its 32 prefetched residual registers spill in a specialised kernel)
                    }
#define OG_BIG2(EPI_, EM_) hipLaunchKernelGGL((gemm_nt_f16x3_big2_kernel<EPI_, EM_>), grid2, block2, 0, stream, g, tiles_m, tiles_n)
                    if (whole && fast_epi && g.c_hl) {
                        if (em == OG_EM_RELU) OG_BIG2(1, OG_EM_RELU);
                        else if (em == OG_EM_RES_HL) OG_BIG2(1, OG_EM_RES_HL);
                        else if (em == OG_EM_NONE) OG_BIG2(1, OG_EM_NONE);
                        else OG_BIG2(1, OG_EM_RUNTIME);
                    } else if (whole && fast_epi && g.Cl) {
                        if (em == OG_EM_NONE) OG_BIG2(2, OG_EM_NONE);
                        else OG_BIG2(2, OG_EM_RUNTIME);
                    } else OG_BIG2(0, OG_EM_RUNTIME);
#undef OG_BIG2
                    return og_launch_status();
                }
                hipLaunchKernelGGL(gemm_nt_f16x3_big_kernel<RD>, dim3((tiles_m * nz + 7) / 8 * 8 * tiles_n), dim3(512), 0, stream, g, tiles_m, tiles_n, rd);
                return og_launch_status();
            }
        }
        const int tiles_m = (a.M + TOK - 1) / TOK;
        const int tiles_m8 = (tiles_m + 7) / 8 * 8;
        if (a.N > 64) {
            const int tiles_n = (a.N + 127) / 128;
            if constexpr (std::is_same<RD, RaggedNone>::value) {
                static const bool spec128 = [] { const char* e = getenv("OG_GEMM_SPEC_EPI"); return !e || atoi(e) != 0; }();
                const bool whole = spec128 && nz == 1 && g.M % TOK == 0 && g.N % 128 == 0 && g.Ch && !g.C32 && !g.Ct && !g.alpha && !g.res;
                if (whole) {
                    const dim3 grid(tiles_m8 * tiles_n), block(256);
#define OG_T128(EPI_, EM_) hipLaunchKernelGGL((gemm_nt_f16x3_kernel<128, 2, RD, EPI_, EM_>), grid, block, 0, stream, g, tiles_m, tiles_n, rd)
                    if (g.c_hl && !g.res_hl && g.relu) { OG_T128(1, OG_EM_RELU); return og_launch_status(); }
                    if (g.c_hl && !g.res_hl && !g.relu) { OG_T128(1, OG_EM_NONE); return og_launch_status(); }
                    if (g.c_hl && g.res_hl && !g.relu) { OG_T128(1, OG_EM_RES_HL); return og_launch_status(); }
                    if (!g.c_hl && g.Cl && !g.res_hl && !g.relu) { OG_T128(2, OG_EM_NONE); return og_launch_status(); }
#undef OG_T128
                }
            }
            hipLaunchKernelGGL((gemm_nt_f16x3_kernel<128, 2, RD>), dim3((tiles_m * nz + 7) / 8 * 8 * tiles_n), dim3(256), 0, stream, g, tiles_m, tiles_n, rd);
        } else {
            hipLaunchKernelGGL((gemm_nt_f16x3_kernel<64, 2, RD>), dim3((tiles_m * nz + 7) / 8 * 8), dim3(256), 0, stream, g, tiles_m, 1, rd);
        }
        return og_launch_status();
    };
    return a.rag ? launch(*a.rag) : launch(RaggedNone{});
}

int og_launch_split_f16(const float* x, int64_t n, void* hi, void* lo, hipStream_t stream) {
    if (!x || !hi || !lo || n <= 0) return OG_E_INVALID;
    if ((n & 3) || ((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return OG_E_ALIGN;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, n4, (_Float16*)hi,
                       (_Float16*)lo);
    return og_launch_status();
}

int og_launch_split_f16_hl(const float* x, int64_t rows, int cols, int64_t ldx, void* out, int64_t ldo, hipStream_t stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return OG_E_INVALID;
    if ((cols & 31) || (ldx & 3) || (ldo & 3) || ldo < 2 * (int64_t)cols || ((uintptr_t)x & 15) || ((uintptr_t)out & 7)) return OG_E_ALIGN;
    const int64_t n4 = rows * (cols / 4);
    hipLaunchKernelGGL(split_f16_hl_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, rows, cols, ldx,
                       (_Float16*)out, ldo);
    return og_launch_status();
}

#if OG_GEMM_TRACE
extern "C" int og_debug_gemm_trace(void* host_dst, size_t bytes) {
    if (bytes > sizeof(og_gemm_trace_buf)) bytes = sizeof(og_gemm_trace_buf);
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(og_gemm_trace_buf), bytes);
}
#endif

extern "C" int og_split_f16(const float* x, int64_t n, void* hi, void* lo, void* stream) {
    og_clear_status();
    return og_launch_split_f16(x, n, hi, lo, (hipStream_t)stream);
}

extern "C" int og_split_f16_rows(const float* x, int64_t ldx, int64_t rows, int32_t cols, int32_t scale_cols, float s1, float s2, void* hi,
                                 void* lo, int64_t ldo, void* stream) {
    og_clear_status();
    if (!x || !hi || !lo || rows <= 0 || cols <= 0 || scale_cols < 0 || scale_cols > cols) return OG_E_INVALID;
    if ((cols & 3) || (scale_cols & 3) || (ldx & 3) || (ldo & 3) || ldx < cols || ldo < cols || ((uintptr_t)x & 15) || ((uintptr_t)hi & 7) ||
        ((uintptr_t)lo & 7))
        return OG_E_ALIGN;
    const int64_t n4 = rows * (cols / 4);
    if ((n4 + 255) / 256 > 0x7fffffffLL) return OG_E_SHAPE;
    hipLaunchKernelGGL(split_f16_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, scale_cols,
                       s1, s2, (_Float16*)hi, (_Float16*)lo, ldo);
    return og_launch_status();
}

extern "C" int og_merge_f16(const void* hi, const void* lo, int64_t n, float* out, void* stream) {
    og_clear_status();
    if (!hi || !lo || !out || n <= 0) return OG_E_INVALID;
    if ((n & 3) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7) || ((uintptr_t)out & 15)) return OG_E_ALIGN;
    const int64_t n4 = n / 4;
    if ((n4 + 255) / 256 > 0x7fffffffLL) return OG_E_SHAPE;
    hipLaunchKernelGGL(merge_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)hi,
                       (const _Float16*)lo, n4, out);
    return og_launch_status();
}

extern "C" int og_split_f16_hl(const float* x, int64_t rows, int32_t cols, int64_t ldx, void* out, int64_t ldo, void* stream) {
    og_clear_status();
    return og_launch_split_f16_hl(x, rows, cols, ldx, out, ldo, (hipStream_t)stream);
}

extern "C" int og_gemm_nt_f16x3(const void* A, int64_t lda, const void* B, int64_t ldb, int32_t M, int32_t N, int32_t K,
                                float scale, const float* bias, int32_t relu, const float* res, int64_t ldr, float* C32, int64_t ldc,
                                void* Ch, void* Cl, int64_t ldch, int32_t c_hl, void* stream) {
    og_clear_status();
    GemmHArgs g{};
    g.A = (const _Float16*)A; g.lda = lda; g.B = (const _Float16*)B; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.scale = scale; g.bias = bias; g.relu = relu; g.res = res; g.ldr = ldr;
    g.C32 = C32; g.ldc = ldc; g.Ch = (_Float16*)Ch; g.Cl = c_hl ? (Ch ? (_Float16*)Ch + 32 : nullptr) : (_Float16*)Cl;
    g.ldch = ldch; g.c_hl = c_hl ? 1 : 0;
    return og_launch_gemm_f16x3(g, (hipStream_t)stream);
}

// the fc.3 form of the GNN: the residual arrives as hl32 rows (og_forward keeps the residual stream that way)
extern "C" int og_gemm_nt_f16x3_reshl(const void* A, int64_t lda, const void* B, int64_t ldb, int32_t M, int32_t N, int32_t K,
                                      float scale, const float* bias, int32_t relu, const void* res_hl, int64_t ldrh, float* C32, int64_t ldc,
                                      void* Ch, void* Cl, int64_t ldch, int32_t c_hl, void* stream) {
    og_clear_status();
    GemmHArgs g{};
    g.A = (const _Float16*)A; g.lda = lda; g.B = (const _Float16*)B; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.scale = scale; g.bias = bias; g.relu = relu; g.res_hl = (const _Float16*)res_hl; g.ldrh = ldrh;
    g.C32 = C32; g.ldc = ldc; g.Ch = (_Float16*)Ch; g.Cl = c_hl ? (Ch ? (_Float16*)Ch + 32 : nullptr) : (_Float16*)Cl;
    g.ldch = ldch; g.c_hl = c_hl ? 1 : 0;
    return og_launch_gemm_f16x3(g, (hipStream_t)stream);
}

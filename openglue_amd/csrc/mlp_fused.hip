// The message MLP of one attentional-GNN layer as ONE kernel (reference attention_gnn.py:43-55 + models/utils.py:48-58):
//
//        x  <-  x + W3' · relu(W0' · [x ; O] + b0') + b3'
//
// (out_proj folded into W0', BatchNorm folded into W3': og_pack_weights).  Round 2 ran fc.0 and fc.3 as two split-f16 GEMM launches
// with the 2D-wide hidden activation H written to and re-read from memory as hl32 rows (134 MB each way per launch at C2, 36 launches
// per step) and two prologue / epilogue rounds per layer.  Here the hidden activation never leaves the register file:
//
//   * MFMA orientation D[channel][token] = W · Xᵀ as in gemm_f16x3.hip.  In the 32x32 accumulator layout a lane owns one token and,
//     per register, one channel -- which is exactly the B-operand layout of the NEXT contraction over those channels (the order of
//     the k index inside an MFMA is free as long as both operands agree): after bias / ReLU / (hi, lo) split the fc.0 accumulators
//     ARE the B fragments of fc.3.  The channel permutation this implies is baked into the packed W3' fragments.
//   * a workgroup = 128 tokens, 8 waves, TWO waves per SIMD (<= 256 registers).  Wave (tb, a) owns token block tb (32 tokens) and
//     hidden half a (8 of the 16 hidden channel blocks), which it works through in two quarters of 4 blocks: 64 accumulator
//     registers for fc.0 plus 128 for its PARTIAL fc.3 result over all 8 output blocks (the sum over its own hidden half); the two
//     waves of a token block exchange half of their partial sums through LDS at the very end.  The token tile is streamed once per
//     quarter (twice; the second read hits the Infinity Cache).
//     [First built with 4 waves, one per SIMD, 128 + 128 accumulators in AGPRs: correct, but a lone wave serialises everything that
//      sits between two of its MFMAs -- fragment reads, LDS-DMA issue, the (hi, lo) conversion -- whatever the instruction order:
//      2520 cycles per 48-MFMA stage against 1536 of matrix-pipe time, 174 us per C2 self layer against 171 for the two launches
//      (profiles/r03_a_*, r03_b_*).  Only a second wave on the SIMD overlaps them.]
//   * weights are packed FRAGMENT-MAJOR (og_pack_mlp_stream): the 16 bytes lane l feeds to an MFMA sit at fragment base + 16 l, so
//     a fragment is one contiguous 1 KiB both in memory (LDS-DMA source: full lines) and in LDS (ds_read_b128: conflict-free, no
//     swizzle); the whole kernel consumes ONE linear stream of 48 stages x 32 KiB (16 fragments = 24 MFMAs per wave each):
//         quarter q in {0, 1}:  16 fc.0 stages (one 32-channel k-group of [x ; O] each)  then  8 fc.3 stages (hidden block j, k-step t)
//   * 3-slot weight ring + 3-slot token ring in LDS, LDS-DMA two stages ahead, one s_barrier per stage; waits counted by hand.
//
// Per 128-token tile: 9216 MFMAs (73.7k matrix-pipe cycles per SIMD); traffic per tile 1.5 MB of weights (L2) + 2 x 256 KB of tokens.
//
// Kernels in this file: mlp_fused_kernel (the above, 128-token tiles: batches); mlp_small_kernel (32-token workgroups whose waves split the
// hidden dimension, weights straight from the same stream: launches of <= 8192 rows, i.e. one to four pairs); proj_small_kernel (the q / k / v
// projections of such launches, 32-token workgroups whose waves split the OUTPUT blocks, its own fragment-major stream).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <type_traits>

#include "og_common.h"

namespace {

typedef unsigned og_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void og_lds_void;
typedef __attribute__((address_space(1))) const void og_glb_void;

constexpr int MT = 128;                       // tokens per workgroup
constexpr int WSTAGE = 32768;                 // one weight stage: 32 fragments of 1 KiB (16 per hidden half)
constexpr int XSTAGE = MT * 128;              // one 32-channel k-group of the token tile: 128 rows x (64 B hi | 64 B lo)
constexpr int XOFF = 3 * WSTAGE;
constexpr int BOFF = XOFF + 3 * XSTAGE;       // biases * 256 (fp32): b0' [2D] then b3' [D]
constexpr int EPI_SLAB = 32 * 144;            // epilogue scratch per 32-token slice (the epilogue writes 128-byte swizzled rows into it; the
                                              // size is the padded-row one of gemm_f16x3.hip, which shares the staging code)

// Compile-time ablations (scripts/build_mlp_ablation.sh; results wrong by construction): 1 = no global stores, 4 = no MFMA,
// 8 = no LDS-DMA after the prologue, 32 = no token LDS-DMA after the prologue.
#ifndef OG_MLP_ABL
#define OG_MLP_ABL 0
#endif
// Experiment builds only (-DOG_MLP_TRACE=1): shader-cycle stamps of every wave at every stage hand-over (before the DMA wait, after
// it, after the barrier), kept in the lanes of three VGPRs (stamp of stage s in lane s) and written out at the end
// (og_debug_mlp_trace, scripts/trace_mlp.py).
#ifndef OG_MLP_ABL16
#define OG_MLP_ABL16 0
#endif
#ifndef OG_MLP_TRACE
#define OG_MLP_TRACE 0
#endif
#if OG_MLP_TRACE
constexpr int OG_MT_BLOCKS = 512;
__device__ unsigned og_mlp_trace_buf[OG_MT_BLOCKS][8][4][64];
#define OG_MT(k_, idx_)                                                                     \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                         \
        mt_v[k_] = lane == (int)(idx_) ? (int)t_ : mt_v[k_];                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
#else
#define OG_MT(k_, idx_) do {} while (0)
#endif

template <int D>
__global__ __launch_bounds__(512) void mlp_fused_kernel(MlpFusedArgs g) {
    static_assert(D == 256 || D == 128, "256-d: 8 output blocks, 2 hidden halves x 2 quarters x 4 hidden blocks, 16 k-groups; 128-d: 4 output blocks, 2 halves x 4 hidden blocks, 8 k-groups");
    constexpr int G0 = 2 * D / 32;            // k-groups of fc.0 (K = 2D) = hidden blocks
    constexpr int NOB = D / 32;               // output blocks
    constexpr int NOBH = NOB / 2;             // ... each wave of a token block finishes
    constexpr int HB2 = G0 / 2;               // hidden blocks per hidden half
    constexpr int NPASS = HB2 / 4;            // passes of 4 hidden blocks per hidden half (256-d: the two quarters; 128-d: one)
    // fc.3 stages per pass, 16 fragments per wave each.  256-d: (hidden block j, k-step t) x 8 output blocks; 128-d: hidden block j, BOTH k-steps x 4 output blocks
    constexpr int NJT = D == 256 ? 8 : 4;
    constexpr int STAGES = NPASS * (G0 + NJT);
    constexpr int XSTAGES = NPASS * G0;
    constexpr int SMEM = BOFF + (2 * D + D) * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    static_assert(SMEM <= 163840 && 8 * 16384 <= BOFF && 2 * EPI_SLAB <= 16384, "LDS budget");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave >> 1, ha = wave & 1;             // token block, hidden half
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * MT;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#if OG_MLP_TRACE
    int mt_v[4] = {0, 0, 0, 0};
    OG_MT(3, 0);
#endif

    auto scalar_ptr = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };

    // ---- LDS-DMA pieces (1 KiB each).  Tokens: wave w fills rows [16w, 16w+16) of a stage (2 pieces of 8 rows x 128 B); weights:
    //      wave w fills fragments [4w, 4w+4).  Every wave reads its token block's 32 rows and its hidden half's 16 fragments. ----
    unsigned xoff[2];
    const char* const baseX = reinterpret_cast<const char*>(g.XO + (int64_t)t0 * g.ld);
    {
        const int rl = lane >> 3, pc = lane & 7;
        const int last = g.M - 1 - t0;                   // rows past the matrix are clamped (computed, never stored)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rt = wave * 16 + h * 8 + rl;
            xoff[h] = (unsigned)((rt < last ? rt : last) * (int)g.ld * 2) + (unsigned)(pc ^ ((rt >> 1) & 7)) * 16u;
            asm volatile("" : "+v"(xoff[h]));
        }
    }
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    // One asm block per operand and stage: a single M0 write, the pieces of a wave are contiguous both in memory and in LDS (the
    // instruction offset applies to the global AND the LDS address), no per-piece address arithmetic.  [As builtin calls every piece
    // carried ~10 SALU + a VGPR copy + its own M0 write; LDS-DMA issue is not hidden by the other wave of the SIMD: the kernel's
    // time was matrix-pipe time + ~90 cycles x pieces per SIMD (profiles/r03_c_*).]
    const char* const baseW = g.wstream + wave * 4 * 1024;
#ifndef OG_MLP_BUF
#define OG_MLP_BUF 0      // experiments: 1 = the LDS-DMA pieces as buffer_load ... lds (MUBUF) instead of global_load_lds
#endif
#if OG_MLP_BUF
    typedef unsigned og_sgpr4 __attribute__((ext_vector_type(4)));
    auto make_res = [&](const char* base, unsigned bytes) {
        const uint64_t v = (uint64_t)(uintptr_t)base;
        og_sgpr4 r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32) & 0xFFFFu);       // stride 0
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000u;
        return r;
    };
    const og_sgpr4 resW = make_res(g.wstream, (unsigned)(STAGES * WSTAGE));
    const og_sgpr4 resX = make_res(baseX, 0x7FFFFFFFu);
    auto issue_w4 = [&](int s, int slot) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(s * WSTAGE + wave * 4096));
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + slot * WSTAGE + wave * 4096);
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %1, %2 offen lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds"
                     :: "v"(lane16), "s"(resW), "s"(so), "s"(m0v) : "memory");
    };
    auto issue_x2 = [&](int xs, int slot) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((xs % G0) * 128));
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + XOFF + slot * XSTAGE + wave * 2048);
        asm volatile("s_mov_b32 m0, %4\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %2, %3 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "v"(xoff[0]), "v"(xoff[1]), "s"(resX), "s"(so), "s"(m0v) : "memory");
    };
#else
    auto issue_w4 = [&](int s, int slot) {               // the 4 pieces of this wave of weight stage s into ring slot `slot`
        const char* src = scalar_ptr(baseW + (int64_t)s * WSTAGE);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + slot * WSTAGE + wave * 4096);
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:3072"
                     :: "v"(lane16), "s"(src), "s"(m0v) : "memory");
    };
    auto issue_x2 = [&](int xs, int slot) {              // the 2 pieces of this wave of token stage xs (k-group xs % G0)
        const char* src = scalar_ptr(baseX + (int64_t)(xs % G0) * 128);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + XOFF + slot * XSTAGE + wave * 2048);
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2"
                     :: "v"(xoff[0]), "v"(xoff[1]), "s"(src), "s"(m0v) : "memory");
    };
#ifndef OG_MLP_DMA_SPREAD
#define OG_MLP_DMA_SPREAD 0      // experiment (round 6): the six LDS-DMA pieces of a stage one per free slot instead of 4 + 2 back to back
#endif
    [[maybe_unused]] auto issue_w1 = [&](int s, int slot, auto P) {      // piece p of this wave of weight stage s
        constexpr int pce = decltype(P)::value;
        const unsigned l16 = lane16;                           // (a plain use: asm operands alone do not capture inside a generic lambda)
        const char* src = scalar_ptr(baseW + (int64_t)s * WSTAGE);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + slot * WSTAGE + wave * 4096);
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:%3"
                     :: "v"(l16), "s"(src), "s"(m0v), "n"(1024 * pce) : "memory");
    };
    [[maybe_unused]] auto issue_x1 = [&](int xs, int slot, auto H) {     // piece h of this wave of token stage xs
        constexpr int hh_ = decltype(H)::value;
        const unsigned xo = xoff[hh_];
        const char* src = scalar_ptr(baseX + (int64_t)(xs % G0) * 128);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + XOFF + slot * XSTAGE + wave * 2048 + hh_ * 1024);
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1"
                     :: "v"(xo), "s"(src), "s"(m0v) : "memory");
    };

#endif

    // ---- prologue: the bias loads first (inline asm: the compiler must not drain the DMA pieces issued behind them), then W(0), X(0),
    //      W(1), X(1); the biases * 256 go to LDS in their shadow ----
    float bv0, bv3;                           // (512 threads cover the 2D + D values at least once; duplicates write the same LDS word)
    {
        const float* p0 = g.b0 + (tid & (2 * D - 1));
        const float* p3 = g.b3 + (tid & (D - 1));
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(bv0), "=&v"(bv3) : "v"(p0), "v"(p3) : "memory");
    }
    issue_w4(0, 0); issue_x2(0, 0);
    issue_w4(1, 1); issue_x2(1, 1);
    // accumulator multipliers of the two matrices: 1 / (power-of-two pre-scale chosen at pack time)
    const float sc0 = g.scales_dev ? g.scales_dev[0] : g.scale, sc3 = g.scales_dev ? g.scales_dev[1] : g.scale;
    {
        const float is0 = 1.f / sc0, is3 = 1.f / sc3;
        const unsigned ba0 = lds0 + BOFF + (unsigned)(tid & (2 * D - 1)) * 4u, ba3 = lds0 + BOFF + (unsigned)(2 * D + (tid & (D - 1))) * 4u;
        asm volatile("s_waitcnt vmcnt(12)\n\t"          // the two bias loads are older than the 12 pieces
                     "v_mul_f32 %0, %0, %4\n\t"
                     "v_mul_f32 %1, %1, %5\n\t"
                     "ds_write_b32 %2, %0\n\t"
                     "ds_write_b32 %3, %1"
                     : "+v"(bv0), "+v"(bv3) : "v"(ba0), "v"(ba3), "s"(is0), "s"(is3) : "memory");
    }

    f32x16 acc0[4], acc3[NOB];
    f16x8 wh[2], wl[2], xh[2], xl[2];
    auto lds_read = [&](f16x8& dst, unsigned addr, int imm) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); };
    // this wave's fragment (block b, part) of the current weight stage: (b * 2 + part) KiB behind the base of its hidden half
    // (part 0 = hi, 1 = lo; fc.0 stages: b = t * 4 + i, fc.3 stages: b = output block i)
    unsigned wa = 0;
    const int swz = (l31 >> 1) & 7;
    unsigned xk[2][2], xa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
            xk[t][hl] = (unsigned)((tb * 32 + l31) * 128) + ((unsigned)(((2 * t + hi) ^ swz) * 16) ^ (hl ? 64u : 0u));
    auto set_x = [&](int slot) {
        const unsigned b = lds0 + XOFF + slot * XSTAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) { xa[t][0] = b + xk[t][0]; xa[t][1] = b + xk[t][1]; }
    };
    auto read_x = [&](int t) { lds_read(xh[t], xa[t][0], 0); lds_read(xl[t], xa[t][1], 0); };
    auto set_w = [&](int slot) { wa = lds0 + slot * WSTAGE + ha * 16384 + lane16; };
    // LDS returns in order: "all but the N newest reads have landed".  The "+v" operand ties the wait to the register it releases.
    auto wait1 = [&](int newer, f16x8& r) {
        if (newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
        else if (newer == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r));
        else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r));
        else if (newer == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r));
        else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r));
    };
    auto tie2 = [&](f16x8& a, f16x8& b) { asm volatile("" : "+v"(a), "+v"(b)); };

    // accumulator initialisation from the bias area (float offset `off`, NB consecutive channel blocks).  Register r of a 32x32
    // accumulator holds channel (r & 3) + 8 (r >> 2) + 4 hi of the block.  Inline-asm reads with their own full wait: LDS reads the
    // compiler knows about would be ordered behind the LDS-DMA in flight (vmcnt(0)) and could slip between the counted fragment reads.
    auto init_acc2 = [&](f32x16& a0, f32x16& a1, int off) {
        const unsigned ad = lds0 + BOFF + (unsigned)(off + 4 * hi) * 4u;
        f32x4 b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[i][q]) : "v"(ad), "i"((i * 32 + 8 * q) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[4 * q + e] = b[0][q][e]; a1[4 * q + e] = b[1][q][e]; }
    };

    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");       // my pieces of W(0), X(0) landed (W(1), X(1) may still fly), my bias stores too
    __builtin_amdgcn_s_barrier();                                     // ... everybody's
    // fc.3 partial sums: the output blocks this wave finishes (NOBH ha .. NOBH ha + NOBH - 1) start from the bias, the ones it hands to its partner from 0
#pragma unroll
    for (int i = 0; i < NOB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[i][r] = 0.f;
    if constexpr (D == 256) {
        if (ha == 0) { init_acc2(acc3[0], acc3[1], 2 * D); init_acc2(acc3[2], acc3[3], 2 * D + 64); }
        else { init_acc2(acc3[4], acc3[5], 2 * D + 128); init_acc2(acc3[6], acc3[7], 2 * D + 192); }
    } else {
        if (ha == 0) init_acc2(acc3[0], acc3[1], 2 * D);
        else init_acc2(acc3[2], acc3[3], 2 * D + 64);
    }

    int s = 0, wslot = 0;           // weight stage being consumed and its ring slot
    int xs = 0, xslot = 0;          // token stage being consumed (fc.0 stages only) and its ring slot
    auto next3 = [](int v) { return v == 2 ? 0 : v + 1; };
    auto prev3 = [](int v) { return v == 0 ? 2 : v - 1; };

    // ---- one stage = 4 groups of 6 MFMAs per wave (two channel blocks x three split-f16 passes).  The four fragment registers of a
    //      group (lo 0, lo 1, hi 0, hi 1) are re-used by the next group: each is re-loaded right after the LAST MFMA that reads it
    //      and waited for individually in front of the first MFMA that needs it (LDS returns in order).  LDS-DMA pieces, the token
    //      fragments of the second k-step and the (hi, lo) conversion of the next hidden fragments sit behind MFMAs 2 and 3, where no
    //      fragment read is issued.  The hand-over to the next stage (DMA wait + barrier) sits between MFMA 3 and 4 of the last
    //      group, when all of this wave's reads of the stage have been waited for; the next stage's first fragments follow it. ----
    auto hand_over = [&](bool next_exists, bool next_is_fc0, int next_xslot, int issued) {
        OG_MT(0, s);
        // everything issued before this stage has landed (only this stage's own pieces may still fly)
        if (issued == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (issued == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OG_MT(1, s);
        if (next_exists) {
            __builtin_amdgcn_s_barrier();
            OG_MT(2, s);
            set_w(next3(wslot));
            if (next_is_fc0) set_x(next_xslot);
        }
    };
#define OG_SB() __builtin_amdgcn_sched_barrier(0)
#if OG_MLP_ABL & 4
#define OG_MM(acc_, a_, b_) asm volatile("" ::"v"(a_), "v"(b_))
#elif OG_MLP_ABL16     // timing experiment, results WRONG: two 16x16x32 MFMAs (same flops, same pipe cycles) on the first eight accumulator registers
#define OG_MM(acc_, a_, b_)                                                                                   \
    {                                                                                                        \
        typedef float f32x4_ __attribute__((ext_vector_type(4)));                                            \
        f32x4_ c0_ = __builtin_shufflevector(acc_, acc_, 0, 1, 2, 3), c1_ = __builtin_shufflevector(acc_, acc_, 4, 5, 6, 7);   \
        c0_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c0_, 0, 0, 0);                                  \
        c1_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c1_, 0, 0, 0);                                  \
        acc_[0] = c0_[0]; acc_[1] = c0_[1]; acc_[2] = c0_[2]; acc_[3] = c0_[3];                              \
        acc_[4] = c1_[0]; acc_[5] = c1_[1]; acc_[6] = c1_[2]; acc_[7] = c1_[3];                              \
    }
#else
#define OG_MM(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, acc_, 0, 0, 0)
#endif
    // fragment k of group G: k = 0: lo of block 2G, 1: lo of block 2G+1, 2: hi of block 2G, 3: hi of block 2G+1
#define OG_RD(G, K) lds_read((K) < 2 ? wl[(K) & 1] : wh[(K) & 1], wa, ((2 * (G) + ((K) & 1)) * 2 + ((K) < 2 ? 1 : 0)) * 1024)
    // A0 / A1: the two accumulators; BH / BL: the B fragments; C0..C3: lgkmcnt in front of MFMAs 0..3; SLOT(k): extra work behind MFMA k
#define OG_GROUP(A0, A1, BH, BL, G, C0, C1, C2, C3, SLOT)                                                     \
    {                                                                                                        \
        wait1(C0, wl[0]); OG_SB(); OG_MM(A0, wl[0], BH); OG_SB(); OG_RD((G) + 1, 0); OG_SB();                 \
        wait1(C1, wl[1]); OG_SB(); OG_MM(A1, wl[1], BH); OG_SB(); OG_RD((G) + 1, 1); OG_SB();                 \
        wait1(C2, wh[0]); OG_SB(); OG_MM(A0, wh[0], BL); OG_SB(); SLOT(2); OG_SB();                           \
        wait1(C3, wh[1]); OG_SB(); OG_MM(A1, wh[1], BL); OG_SB(); SLOT(3); OG_SB();                           \
        OG_MM(A0, wh[0], BH); OG_SB(); OG_RD((G) + 1, 2); OG_SB();                                            \
        OG_MM(A1, wh[1], BH); OG_SB(); OG_RD((G) + 1, 3); OG_SB();                                            \
    }
    // the last group of a stage
#define OG_GROUP_LAST(A0, A1, BH, BL, SLOT, HANDOVER, NEXT_EXISTS, NEXT_FC0)                                  \
    {                                                                                                        \
        wait1(3, wl[0]); OG_SB(); OG_MM(A0, wl[0], BH); OG_SB();                                              \
        wait1(2, wl[1]); OG_SB(); OG_MM(A1, wl[1], BH); OG_SB();                                              \
        wait1(1, wh[0]); OG_SB(); OG_MM(A0, wh[0], BL); OG_SB(); SLOT(2); OG_SB();                            \
        wait1(0, wh[1]); OG_SB(); OG_MM(A1, wh[1], BL); OG_SB(); SLOT(3); OG_SB();                            \
        HANDOVER; OG_SB();             /* all my LDS reads of this stage have been waited for */              \
        OG_MM(A0, wh[0], BH); OG_SB();                                                                       \
        if (NEXT_FC0) read_x(0);                                                                             \
        if (NEXT_EXISTS) { OG_RD(0, 0); OG_RD(0, 1); OG_RD(0, 2); }                                           \
        OG_SB();                                                                                             \
        OG_MM(A1, wh[1], BH); OG_SB(); if (NEXT_EXISTS) OG_RD(0, 3); OG_SB();                                 \
    }

    // ---- the residual enters through the matrix pipe: x_new = x + W3' h  =>  acc3[i] += (S3 I) · x[channel block i], and the (hi, lo)
    //      B fragments of x are exactly the token fragments the fc.0 stages of k-groups 0..7 (the x half of [x ; O]) hold in registers
    //      anyway.  In the LAST quarter's stages kg = 0..7 every wave adds HALF of the residual of output block kg to its own partial
    //      sum (the partial sums of the two waves of a token block are added in the exchange anyway): wave (tb, a) the channels of
    //      k-step t = a, 2 MFMAs with a constant "S3 x identity" A fragment (k-step t covers channels 16t .. 16t+15 of the block) --
    //      the same extra work for all eight waves, so nobody arrives late at the stage barrier.  [Round 3, first version: residual rows fetched in the epilogue, turned into the accumulator layout through the LDS
    //      slabs and added with mixed-precision FMAs -- 8 global loads, 48 LDS operations and 32 VALU per wave of an epilogue that is
    //      bound by its LDS operations.]
    const _Float16 s3h = (_Float16)(1.f / sc3);               // power of two >= 2^-14 (og_weight_prescale): exact in binary16
    auto ident = [&](int t) {                                 // built where it is used (4 stages per tile): no registers held across the loop
        int row = l31 - 8 * hi;
        asm volatile("" : "+v"(row));                         // opaque: re-computed at every use, not hoisted out of the loop and spilled
        f16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (row == 16 * t + e) ? s3h : (_Float16)0.f;
        return f;
    };
// RB_ >= 0 (compile time): this stage is k-group RB_ of the last quarter -- the wave that finishes output block RB_ adds its residual
#define OG_RESID(RB_, T_, XF_)                                                                               \
    if constexpr ((RB_) >= 0) {                                                                              \
        if (ha == (T_)) { const f16x8 idt_ = ident(T_); OG_MM(acc3[(RB_) < 0 ? 0 : (RB_)], idt_, XF_); OG_SB(); }   \
    }

    set_w(0); set_x(0);
    read_x(0);
    OG_RD(0, 0); OG_RD(0, 1); OG_RD(0, 2); OG_RD(0, 3);

#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        init_acc2(acc0[0], acc0[1], (HB2 * ha + 4 * pass) * 32);
        init_acc2(acc0[2], acc0[3], (HB2 * ha + 4 * pass + 2) * 32);

        // ================= fc.0: acc0[i] += W0'[hidden block 8 ha + 4 pass + i][k-group] · [x ; O][k-group], 16 stages =================
        auto fc0_stage = [&](int kg, auto RB) {
            constexpr int rb = decltype(RB)::value;                     // >= 0: the residual of output block rb rides in this stage
            const bool ix = xs + 2 < XSTAGES;                           // X(xs+2) exists (W(s+2) always does during fc.0)
            const int wslot2 = prev3(wslot), xslot2 = prev3(xslot);     // slots of W(s+2), X(xs+2): (s + 2) % 3 = (s - 1) % 3
            const bool next_x = kg + 1 < G0;                            // the next stage is an fc.0 stage (reads token fragments)
#if OG_MLP_DMA_SPREAD
#define OG_SLOT_G0(k) { if ((k) == 2) issue_w1(s + 2, wslot2, std::integral_constant<int, 0>{}); if ((k) == 3) { read_x(1); issue_w1(s + 2, wslot2, std::integral_constant<int, 1>{}); } }
#define OG_SLOT_G1(k) { if ((k) == 2) issue_w1(s + 2, wslot2, std::integral_constant<int, 2>{}); if ((k) == 3) issue_w1(s + 2, wslot2, std::integral_constant<int, 3>{}); }
#define OG_SLOT_G2(k) { if (ix) { if ((k) == 2) issue_x1(xs + 2, xslot2, std::integral_constant<int, 0>{}); if ((k) == 3) issue_x1(xs + 2, xslot2, std::integral_constant<int, 1>{}); } }
#else
#define OG_SLOT_G0(k) { if ((k) == 2 && !(OG_MLP_ABL & 8)) issue_w4(s + 2, wslot2); if ((k) == 3) read_x(1); }   /* the weight pieces; the second k-step's token fragments */
#define OG_SLOT_G1(k) { if ((k) == 2 && ix && !(OG_MLP_ABL & (8 | 32))) issue_x2(xs + 2, xslot2); }                  /* the token pieces */
#define OG_SLOT_G2(k) {}
#endif
#define OG_SLOT_NONE(k) {}
            tie2(xh[0], xl[0]);
            OG_GROUP(acc0[0], acc0[1], xh[0], xl[0], 0, 3, 3, 3, 2, OG_SLOT_G0)
            OG_RESID(rb, 0, xh[0])
            OG_GROUP(acc0[2], acc0[3], xh[0], xl[0], 1, 5, 5, 3, 2, OG_SLOT_G1)
            OG_RESID(rb, 0, xl[0])
            tie2(xh[1], xl[1]);
            OG_GROUP(acc0[0], acc0[1], xh[1], xl[1], 2, 3, 3, 3, 2, OG_SLOT_G2)
            OG_RESID(rb, 1, xh[1])
            OG_RESID(rb, 1, xl[1])
            OG_GROUP_LAST(acc0[2], acc0[3], xh[1], xl[1], OG_SLOT_NONE, hand_over(true, next_x, next3(xslot), ix ? 6 : 4), true, next_x)
            ++s; wslot = next3(wslot);
            ++xs; xslot = next3(xslot);
        };
        using NoRes = std::integral_constant<int, -1>;
        int kg0 = 0;
        if (pass == NPASS - 1) {        // the x half of [x ; O] (k-groups 0..NOB-1) in the last pass: unrolled, every stage knows its output block
            fc0_stage(0, std::integral_constant<int, 0>{}); fc0_stage(1, std::integral_constant<int, 1>{});
            fc0_stage(2, std::integral_constant<int, 2>{}); fc0_stage(3, std::integral_constant<int, 3>{});
            if constexpr (NOB == 8) {
                fc0_stage(4, std::integral_constant<int, 4>{}); fc0_stage(5, std::integral_constant<int, 5>{});
                fc0_stage(6, std::integral_constant<int, 6>{}); fc0_stage(7, std::integral_constant<int, 7>{});
            }
            kg0 = NOB;
        }
#pragma unroll 1
        for (int kg = kg0; kg < G0; ++kg) fc0_stage(kg, NoRes{});

        // ================= fc.3: acc3[i] += W3'[block i][hidden block 8 ha + 4 pass + j] · relu(acc0[j] / 256), 8 stages (j, t) =================
        // hidden block j as B fragments: element e of k-step t is accumulator register 8t + e (og_pack_mlp_stream permutes W3' to match)
        unsigned hh[2][4], hl[2][4];                   // [buffer][dword]
        float cv[4];
        // (hi, lo) conversion of accumulator registers 8t + 4h .. + 3 of a hidden block (half h of k-step t) in two small steps
        auto convert_step = [&](const f32x16& a, int t, int step, int buf) {
#pragma clang fp contract(off)
            const int h = step >> 1, r0 = 8 * t + 4 * h;
            if ((step & 1) == 0) {
                cv[0] = fmaxf(a[r0] * sc0, 0.f); cv[1] = fmaxf(a[r0 + 1] * sc0, 0.f); cv[2] = fmaxf(a[r0 + 2] * sc0, 0.f); cv[3] = fmaxf(a[r0 + 3] * sc0, 0.f);
            } else {
                og_split4(cv[0], cv[1], cv[2], cv[3], hh[buf][2 * h], hl[buf][2 * h], hh[buf][2 * h + 1], hl[buf][2 * h + 1]);
            }
        };
#pragma unroll
        for (int st = 0; st < 4; ++st) convert_step(acc0[0], 0, st, 0);
        if constexpr (D == 256) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                const bool iw = s + 2 < STAGES;
                const int wslot2 = prev3(wslot);
                const bool next_exists = s + 1 < STAGES;
                const bool next_x = jt + 1 == NJT && pass + 1 < NPASS;      // the next stage is the first fc.0 stage of the next quarter
                const int hb = jt & 1;
                const f16x8 bh = __builtin_bit_cast(f16x8, og_u32x4{hh[hb][0], hh[hb][1], hh[hb][2], hh[hb][3]});
                const f16x8 bl = __builtin_bit_cast(f16x8, og_u32x4{hl[hb][0], hl[hb][1], hl[hb][2], hl[hb][3]});
                constexpr int NXT = 0;
                (void)NXT;
                // the next stage's B fragments: hidden block (jt + 1) / 2, k-step (jt + 1) & 1
                auto conv = [&](int step) { if (jt + 1 < NJT) convert_step(acc0[(jt + 1 < NJT ? jt + 1 : jt) >> 1], (jt + 1) & 1, step, hb ^ 1); };
#if OG_MLP_DMA_SPREAD
#define OG_SLOT_P0(k) { if (iw) { if ((k) == 2) issue_w1(s + 2, wslot2, std::integral_constant<int, 0>{}); else issue_w1(s + 2, wslot2, std::integral_constant<int, 1>{}); } }
#define OG_SLOT_P1(k) { if (iw) { if ((k) == 2) issue_w1(s + 2, wslot2, std::integral_constant<int, 2>{}); else issue_w1(s + 2, wslot2, std::integral_constant<int, 3>{}); } }
#else
#define OG_SLOT_P0(k) { if ((k) == 2 && iw && !(OG_MLP_ABL & 8)) issue_w4(s + 2, wslot2); }
#define OG_SLOT_P1(k) {}
#endif
#define OG_SLOT_C2(k) conv((k) - 2)
#define OG_SLOT_C3(k) conv((k))
                OG_GROUP(acc3[0], acc3[1], bh, bl, 0, 3, 3, 3, 2, OG_SLOT_P0)
                OG_GROUP(acc3[2], acc3[3], bh, bl, 1, 3, 3, 3, 2, OG_SLOT_P1)
                OG_GROUP(acc3[4], acc3[5], bh, bl, 2, 3, 3, 3, 2, OG_SLOT_C2)
                OG_GROUP_LAST(acc3[6], acc3[7], bh, bl, OG_SLOT_C3, hand_over(next_exists, next_x, xslot, iw ? 4 : 0), next_exists, next_x)
                ++s; wslot = next3(wslot);
            }
        } else {
            // 128-d: stage j = hidden block j with BOTH k-steps over the 4 output blocks: groups 0, 1 take k-step 0 (B fragments in buffer 0), groups
            // 2, 3 k-step 1 (buffer 1).  Conversions ride behind the MFMAs as above: k-step 1 of block j (into buffer 1) during groups 0, 1,
            // k-step 0 of block j + 1 (into buffer 0, free once group 1 has issued) during groups 2, 3.
#pragma unroll
            for (int j = 0; j < NJT; ++j) {
                const bool iw = s + 2 < STAGES;
                const int wslot2 = prev3(wslot);
                const bool next_exists = s + 1 < STAGES;
                auto conv1 = [&](int step) { convert_step(acc0[j], 1, step, 1); };
                auto conv0n = [&](int step) { if (j + 1 < NJT) convert_step(acc0[j + 1 < NJT ? j + 1 : j], 0, step, 0); };
#define OG_SLOT_Q0(k) { if ((k) == 2 && iw && !(OG_MLP_ABL & 8)) issue_w4(s + 2, wslot2); conv1((k) - 2); }
#define OG_SLOT_Q1(k) conv1((k))
#define OG_SLOT_Q2(k) conv0n((k) - 2)
#define OG_SLOT_Q3(k) conv0n((k))
                {
                    const f16x8 bh = __builtin_bit_cast(f16x8, og_u32x4{hh[0][0], hh[0][1], hh[0][2], hh[0][3]});
                    const f16x8 bl = __builtin_bit_cast(f16x8, og_u32x4{hl[0][0], hl[0][1], hl[0][2], hl[0][3]});
                    OG_GROUP(acc3[0], acc3[1], bh, bl, 0, 3, 3, 3, 2, OG_SLOT_Q0)
                    OG_GROUP(acc3[2], acc3[3], bh, bl, 1, 3, 3, 3, 2, OG_SLOT_Q1)
                }
                {
                    const f16x8 bh = __builtin_bit_cast(f16x8, og_u32x4{hh[1][0], hh[1][1], hh[1][2], hh[1][3]});
                    const f16x8 bl = __builtin_bit_cast(f16x8, og_u32x4{hl[1][0], hl[1][1], hl[1][2], hl[1][3]});
                    OG_GROUP(acc3[0], acc3[1], bh, bl, 2, 3, 3, 3, 2, OG_SLOT_Q2)
                    OG_GROUP_LAST(acc3[2], acc3[3], bh, bl, OG_SLOT_Q3, hand_over(next_exists, false, xslot, iw ? 4 : 0), next_exists, false)
                }
                ++s; wslot = next3(wslot);
            }
        }
    }
#undef OG_RESID
#undef OG_RES1
      // (OG_GROUP / OG_GROUP_LAST / OG_MM / OG_RD / OG_SB stay defined: proj_stream_kernel below is built from the same stage groups)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OG_MT(3, 1);
    const int tok0 = t0 + tb * 32;
    char* const rows = reinterpret_cast<char*>(g.XO);
    int64_t rowb[4];                                          // byte offset of this lane's row per store instruction
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int r = tok0 + it * 8 + (lane >> 3);
        if (r > g.M - 1) r = g.M - 1;
        rowb[it] = (int64_t)r * g.ld * 2 + (lane & 7) * 16 + NOBH * ha * 128;
    }
    __syncthreads();          // every wave is past its last fragment reads, no DMA in flight: the rings are free

    // ================= the two waves of a token block exchange half of their partial sums =================
    // Wave (tb, a) finishes output blocks NOBH a .. NOBH a + NOBH - 1 and hands its partial sums of the other NOBH to its partner: 16 KiB per wave at 256-d,
    // [block][register group][lane] x 16 B (conflict-free), region `wave`; afterwards the region a wave has READ belongs to it alone
    // (epilogue slabs).
    f32x16 accf[NOBH];
    {
        char* const mine = smem + wave * 16384;
        char* const theirs = smem + (wave ^ 1) * 16384;
        const bool lowhalf = ha == 0;                          // wave-uniform; element-wise selects keep the accumulators in registers
#pragma unroll
        for (int i = 0; i < NOBH; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lowhalf ? acc3[NOBH + i][4 * q + e] : acc3[i][4 * q + e];
                *reinterpret_cast<f32x4*>(mine + ((i * 4 + q) * 64 + lane) * 16) = v;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NOBH; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + ((i * 4 + q) * 64 + lane) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) accf[i][4 * q + e] = (lowhalf ? acc3[i][4 * q + e] : acc3[NOBH + i][4 * q + e]) + v[e];
            }
        }
    }

    // ================= epilogue: x <- acc / S3 (bias and residual inside), written back as hl32 rows =================
    // A lane owns ONE token and 4 consecutive channels per register group.  Result rows go out as whole 128-byte lines (one channel
    // block of one token: 64 B hi | 64 B lo), 8 rows per instruction, through two per-wave LDS slabs (as gemm_f16x3_epilogue_fast),
    // channel blocks in pairs.
    {
#pragma clang fp contract(off)
        constexpr int ROWB = 128 + 16;
        char* const slab2 = smem + (wave ^ 1) * 16384;
        const unsigned rd_off = (unsigned)((lane >> 3) * ROWB + (lane & 7) * 16);
        const bool full = t0 + MT <= g.M;                    // block-uniform
#pragma unroll
        for (int ip = 0; ip < NOBH / 2; ++ip) {
            f32x16 a[2];
            a[0] = accf[2 * ip]; a[1] = accf[2 * ip + 1];
            // v = acc / S3 (bias and residual are inside the accumulator); pinned: og_split4 must see ONE rounded product
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = a[i] * sc3;
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(a[i][r]));
            }
            // registers -> slabs: slab i = [32 tok][hi 64 B | lo 64 B] of channel block cb0 + 2 ip + i
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned ha2, la, hb2, lb;
                    og_split4(a[i][4 * q], a[i][4 * q + 1], a[i][4 * q + 2], a[i][4 * q + 3], ha2, la, hb2, lb);
                    char* d = slab2 + i * EPI_SLAB + l31 * ROWB + (8 * q + 4 * hi) * 2;
                    *reinterpret_cast<uint2*>(d) = make_uint2(ha2, hb2);
                    *reinterpret_cast<uint2*>(d + 64) = make_uint2(la, lb);
                }
            // slabs -> whole 128-byte lines
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f16x8 tt[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) tt[it] = *reinterpret_cast<const f16x8*>(slab2 + i * EPI_SLAB + it * 8 * ROWB + rd_off);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (OG_MLP_ABL & 1) continue;
                    if (full || tok0 + it * 8 + (lane >> 3) < g.M) *reinterpret_cast<f16x8*>(rows + rowb[it] + (2 * ip + i) * 128) = tt[it];
                }
            }
        }
    }
#if OG_MLP_TRACE
    OG_MT(3, 2);                                   // all stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OG_MT(3, 3);                                   // all stores acknowledged
    mt_v[3] = lane == 4 ? (int)__builtin_amdgcn_s_getreg(0xF804) : mt_v[3];     // HW_ID
    mt_v[3] = lane == 5 ? (int)__builtin_amdgcn_s_getreg(0xF814) : mt_v[3];     // XCC_ID
    if (blockIdx.x < OG_MT_BLOCKS)
        for (int k = 0; k < 4; ++k) og_mlp_trace_buf[blockIdx.x][wave][k][lane] = (unsigned)mt_v[k];
#endif
}


// =================================================================================================================================
// proj_stream_kernel -- the q / k / v projections of a BATCH (launches of more than 8192 token rows) in the style of mlp_fused_kernel.
// The 256 x 256-tile GEMM it replaces spends a quarter of a tile's life in prologue + epilogue (K = 256 is 8 k-stages), moves BOTH
// operands through LDS and reaches 0.097 of the f16 peak at C2; at D = 128 (N = 384: no 256-tile form) the 128-token
// tile kernel runs 4 k-stages per tile.  Here:
//   * a workgroup = 128 tokens, 8 waves, wave (tb, a) = token block tb x half a of every 128-channel group of the output ("super-pair":
//     4 blocks of 32 channels, wave a owns blocks 2a, 2a + 1 -- one full 128-byte line of each output plane per token);
//   * the x rows of the tile are NOT staged in LDS: every wave loads the B fragments of its 32 tokens -- all K channels, (hi, lo) --
//     straight from the hl32 rows into registers ONCE (K / 16 x 2 fragments: 128 VGPRs at K = 256) and keeps them for all output blocks;
//   * only the weights stream through LDS: the fragment-major stream of og_pack_proj_stream_big, 32 KiB stages (4 blocks x 4 k-steps x
//     (hi, lo)), 3-slot ring, LDS-DMA two stages ahead, one barrier per stage -- the stage groups, counted waits and hand-over of
//     mlp_fused_kernel;
//   * after the K / 64 stages of a super-pair the wave scales, splits and writes its 32 x 64 outputs as whole 128-byte lines of the two
//     planes (through a private LDS slab), then starts the next super-pair from the bias.
// Workgroups whose first row is below split_row take the super-pairs [a0, a1), the others [b0, b1) (the cross layer's launch).
struct ProjStreamArgs {
    const _Float16* X; int64_t ld; int M;      // [M] hl32 rows, the first 2K halves = x
    const char* wstream;                       // og_pack_proj_stream_big
    const float* bias;                         // [N]
    const float* scale_dev;                    // DEVICE: 1 / pre-scale of the matrix
    _Float16* Ch; _Float16* Cl; int64_t ldc;   // output planes [M][ldc]
    int split_row, a0, a1, b0, b1;             // super-pair ranges (units of 128 output channels)
};

template <int K>
__global__ __launch_bounds__(512) void proj_stream_kernel(ProjStreamArgs g) {
    static_assert(K == 256 || K == 128, "K = the descriptor width");
    constexpr int KS = K / 16, SPS = KS / 4;   // k-steps; stages per super-pair
    constexpr int PBOFF = 3 * WSTAGE;          // bias / scale of the columns of this workgroup: up to 1024 floats
    constexpr int PSLAB = PBOFF + 4096;        // per-wave epilogue slabs: 32 rows x (128 + 16) B
    constexpr int ROWB = 128 + 16;
    __shared__ __attribute__((aligned(16))) char smem[PSLAB + 8 * EPI_SLAB];
    static_assert(PSLAB + 8 * EPI_SLAB <= 163840, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave >> 1, ha = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * MT;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int sp0 = __builtin_amdgcn_readfirstlane(t0 < g.split_row ? g.a0 : g.b0), sp1 = __builtin_amdgcn_readfirstlane(t0 < g.split_row ? g.a1 : g.b1);
    if (sp1 <= sp0) return;
#if OG_MLP_TRACE
    int mt_v[4] = {0, 0, 0, 0};      // experiment builds: [0..2][stage] hand-over stamps as in mlp_fused_kernel; [3]: 0 entry, 1 prologue done, 8 + 2 i / 9 + 2 i epilogue i, 30 end
    OG_MT(3, 0);
#endif
    const int nst = (sp1 - sp0) * SPS;         // weight stages of this workgroup

    auto scalar_ptr = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const char* const baseW = g.wstream + (int64_t)sp0 * SPS * WSTAGE + wave * 4 * 1024;
    auto issue_w4 = [&](int st, int slot) {               // the 4 pieces of this wave of weight stage st (relative to sp0) into ring slot `slot`
        const char* src = scalar_ptr(baseW + (int64_t)st * WSTAGE);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + slot * WSTAGE + wave * 4096);
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:3072"
                     :: "v"(lane16), "s"(src), "s"(m0v) : "memory");
    };
    // ---- this wave's B fragments: token l31 of its block, k-step ks = channels 16 ks + 8 hi .. + 7, (hi, lo).  A lane needs 16-byte pieces of ITS
    //      token's row: loaded straight from memory that is 32 row segments of 32 bytes per instruction -- 21k cycles of a 86k-cycle tile
    //      (profiles/r05_d_proj_stream_trace_v1.log).  So the tile's x rows are staged through LDS like the token stages of mlp_fused_kernel: LDS-DMA
    //      of whole 128-byte lines (k-group kg of 8 rows per piece, XOR swizzle on the source), fragment reads with the matching swizzle -- the
    //      staging area is the (still empty) weight ring and what lies behind it. ----
    f16x8 xh[KS], xl[KS];
    {
        constexpr int KG = K / 32;                         // k-groups: one 128-byte line per row each
        static_assert(KG * XSTAGE <= PSLAB + 8 * EPI_SLAB, "the staging area fits the kernel's LDS");
        unsigned xoff[2];
        const char* const baseX = reinterpret_cast<const char*>(g.X) + (int64_t)t0 * g.ld * 2;
        {
            const int rl = lane >> 3, pc = lane & 7;
            const int last = g.M - 1 - t0;                 // rows past the matrix are clamped (computed, never stored)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rt = wave * 16 + h * 8 + rl;
                xoff[h] = (unsigned)((rt < last ? rt : last) * (int)g.ld * 2) + (unsigned)(pc ^ ((rt >> 1) & 7)) * 16u;
            }
        }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const char* src = scalar_ptr(baseX + kg * 128);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + kg * XSTAGE + wave * 2048);
            asm volatile("s_mov_b32 m0, %3\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, %2\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2"
                         :: "v"(xoff[0]), "v"(xoff[1]), "s"(src), "s"(m0v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // every wave's rows are in LDS
        const int swz = (l31 >> 1) & 7;
        const unsigned rb = lds0 + (unsigned)((tb * 32 + l31) * 128);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned ah = rb + (unsigned)((ks >> 1) * XSTAGE) + (unsigned)(((2 * (ks & 1) + hi) ^ swz) * 16);
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[ks]) : "v"(ah) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(xl[ks]) : "v"(ah ^ 64u) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xh[ks]), "+v"(xl[ks]));
        __builtin_amdgcn_s_barrier();                      // everybody has its fragments: the staging area becomes the weight ring
    }
    issue_w4(0, 0);
    if (nst > 1) issue_w4(1, 1);
    const float sc = g.scale_dev[0];
    {   // bias / scale of this workgroup's columns -> LDS (the accumulators start from it)
        const float isc = 1.f / sc;
        const int ncol = 128 * (sp1 - sp0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = tid + 512 * j;
            if (c < ncol) *reinterpret_cast<float*>(smem + PBOFF + c * 4) = g.bias[128 * sp0 + c] * isc;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // W(0), W(1), the bias: all here
    __builtin_amdgcn_s_barrier();

    f32x16 acc[2];
    f16x8 wh[2], wl[2];
    unsigned wa = 0;
    auto lds_read = [&](f16x8& dst, unsigned addr, int imm) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); };
    auto set_w = [&](int slot) { wa = lds0 + slot * WSTAGE + ha * 16384 + lane16; };
    auto wait1 = [&](int newer, f16x8& r) {
        if (newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
        else if (newer == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r));
        else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r));
        else if (newer == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r));
        else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r));
    };
    auto read_x = [&](int) {};                             // (the stage groups' hook for token fragments: they live in registers here)
    auto init_acc2 = [&](f32x16& a0, f32x16& a1, int off) {          // as in mlp_fused_kernel: own asm reads, own full wait
        const unsigned ad = lds0 + PBOFF + (unsigned)(off + 4 * hi) * 4u;
        f32x4 b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[i][q]) : "v"(ad), "i"((i * 32 + 8 * q) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[4 * q + e] = b[0][q][e]; a1[4 * q + e] = b[1][q][e]; }
    };
    auto next3 = [](int v) { return v == 2 ? 0 : v + 1; };
    auto prev3 = [](int v) { return v == 0 ? 2 : v - 1; };
    int s = 0, wslot = 0;
    auto hand_over = [&](bool next_exists, int younger) {      // younger: vector-memory operations issued after the pieces of stage s + 1
        OG_MT(0, s);
        if (younger >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OG_MT(1, s);
        if (next_exists) {
            __builtin_amdgcn_s_barrier();
            OG_MT(2, s);
            set_w(next3(wslot));
        }
    };

    // epilogue addresses: store instruction `it` writes rows tok0 + 8 it + (lane >> 3), 16-byte chunk lane & 7 of a 128-byte line
    const int tok0 = t0 + tb * 32;
    int64_t rowoff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int r = tok0 + it * 8 + (lane >> 3);
        if (r > g.M - 1) r = g.M - 1;
        rowoff[it] = (int64_t)r * g.ldc * 2 + (lane & 7) * 16 + ha * 128;
    }
    const bool full = t0 + MT <= g.M;
    const unsigned slab = lds0 + PSLAB + wave * EPI_SLAB;
    const unsigned swr = slab + (unsigned)(l31 * ROWB + 8 * hi), srd = slab + (unsigned)((lane >> 3) * ROWB + (lane & 7) * 16);

    OG_MT(3, 1);
    set_w(0);
    OG_RD(0, 0); OG_RD(0, 1); OG_RD(0, 2); OG_RD(0, 3);
    // The epilogue of a super-pair, at the boundary: acc / S -> (hi, lo) halves (VALU), then one plane at a time -- 8 LDS writes, 4 LDS reads, 4
    // stores of whole 128-byte lines -- and the accumulators restart from the next bias: 2.3k cycles with an idle matrix pipe per super-pair (all
    // eight waves at once).  [Measured alternative (profiles/r05_e_proj_stream_trace_v2.log): only the conversion at the boundary (0.7k), the planes
    // deferred into the first stage of the next super-pair behind its MFMAs, at different groups for the two waves of a SIMD -- that stage then takes
    // 5.2k cycles instead of 1.9k: 3.3k per super-pair against 2.5k.  The LDS round trips of a plane drain the wave's fragment pipeline, and the
    // other wave of the SIMD does not fill the gap.]  Every part starts and ends with a full LDS wait, so the counted waits of the fragment
    // pipeline around it stay valid (they become stricter, never weaker).
    unsigned ph[2][4][2], pl[2][4][2];
    int esp = 0;                                           // the super-pair the pending planes belong to
    auto epi_convert = [&](int sp) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[i][4 * q] * sc, v1 = acc[i][4 * q + 1] * sc, v2 = acc[i][4 * q + 2] * sc, v3 = acc[i][4 * q + 3] * sc;
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));      // og_split4 must see ONE rounded product
                og_split4(v0, v1, v2, v3, ph[i][q][0], pl[i][q][0], ph[i][q][1], pl[i][q][1]);
            }
        esp = sp;
    };
    auto epi_plane = [&](int pln) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wl[0]), "+v"(wl[1]), "+v"(wh[0]), "+v"(wh[1]) :: "memory");     // fragment reads in flight land first
        // a lane's 4 consecutive channels of block i, register group q: bytes i * 64 + (8 q + 4 hi) * 2 of its token's line
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 v = pln ? make_uint2(pl[i][q][0], pl[i][q][1]) : make_uint2(ph[i][q][0], ph[i][q][1]);
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(swr), "v"(v), "i"(i * 64 + q * 16) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f16x8 tt[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tt[it]) : "v"(srd), "i"(it * 8 * ROWB) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tt[0]), "+v"(tt[1]), "+v"(tt[2]), "+v"(tt[3]) :: "memory");
        char* const base = reinterpret_cast<char*>(pln ? g.Cl : g.Ch) + (int64_t)esp * 256;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (OG_MLP_ABL & 1) continue;
            // (a full tile issues exactly 4 store instructions per plane: the hand-over of the stage counts them; a partial tile's predicated
            //  stores may be skipped, so its hand-over does not count them and waits for whatever was issued)
            if (full || tok0 + it * 8 + (lane >> 3) < g.M) *reinterpret_cast<f16x8*>(base + rowoff[it]) = tt[it];
        }
    };
#define OG_SLOT_PW(k) { if ((k) == 2 && iw && !(OG_MLP_ABL & 8)) issue_w4(s + 2, wslot2); }
#define OG_SLOT_E1(k) {}
#define OG_SLOT_E2(k) {}
#define OG_SLOT_E3(k) {}
#pragma unroll 1
    for (int sp = sp0; sp < sp1; ++sp) {
        init_acc2(acc[0], acc[1], 128 * (sp - sp0) + 64 * ha);
#pragma unroll
        for (int kq = 0; kq < SPS; ++kq) {
            const bool iw = s + 2 < nst;
            const int wslot2 = prev3(wslot);
            const bool next_exists = s + 1 < nst;
            const bool pend = kq == 0 && sp > sp0;         // the planes of the previous super-pair went out just before this stage
            const int younger = (iw ? 4 : 0) + ((pend && full && !(OG_MLP_ABL & 1)) ? 8 : 0);      // + their 8 stores (issued after the pieces of stage s + 1)
            OG_GROUP(acc[0], acc[1], xh[4 * kq], xl[4 * kq], 0, 3, 3, 3, 2, OG_SLOT_PW)
            OG_GROUP(acc[0], acc[1], xh[4 * kq + 1], xl[4 * kq + 1], 1, 3, 3, 3, 2, OG_SLOT_E1)
            OG_GROUP(acc[0], acc[1], xh[4 * kq + 2], xl[4 * kq + 2], 2, 3, 3, 3, 2, OG_SLOT_E2)
            OG_GROUP_LAST(acc[0], acc[1], xh[4 * kq + 3], xl[4 * kq + 3], OG_SLOT_E3, hand_over(next_exists, younger), next_exists, false)
            ++s; wslot = next3(wslot);
        }
        OG_MT(3, 8 + 2 * (sp - sp0));
        epi_convert(sp);
        epi_plane(0);
        epi_plane(1);
        OG_MT(3, 9 + 2 * (sp - sp0));
    }
#if OG_MLP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OG_MT(3, 30);
    if (blockIdx.x < OG_MT_BLOCKS)
        for (int k = 0; k < 4; ++k) og_mlp_trace_buf[blockIdx.x][wave][k][lane] = (unsigned)mt_v[k];
#endif
#undef OG_SLOT_E1
#undef OG_SLOT_E2
#undef OG_SLOT_E3
#undef OG_SLOT_PW
#undef OG_SLOT_PN
}
#undef OG_GROUP
#undef OG_GROUP_LAST
#undef OG_MM
#undef OG_RD


// =================================================================================================================================
// mlp_small_kernel -- the same layer for FEW token rows (the reference's inference.py regime: one or a few image pairs).
// mlp_fused_kernel gives a 128-token tile to one workgroup: 9216 MFMAs on one CU, 46 us per launch whether the batch holds 2048 or
// 8192 token rows (profiles/r04_small_batch_kernel_stats_B*.csv: 1.26 of the 3.3 ms of a single-pair step).  Here a workgroup owns 32
// tokens and its 8 waves split the HIDDEN dimension (64 of the 512 hidden channels each), so a launch spreads over four times as
// many CUs and a wave issues 288 MFMAs instead of 1152:
//   * the 32 token rows [x ; O] (64 KB of hl32 rows) are copied to LDS once (rows padded by 16 B: conflict-free ds_read_b128);
//   * wave w: fc.0 for hidden blocks 2w, 2w+1 over all 32 k-steps; accumulators -> ReLU -> (hi, lo) -> B fragments as in the big
//     kernel, handed to the other waves through LDS (64 KB);
//   * wave w: fc.3 for OUTPUT block w over all 512 hidden channels (two accumulator chains), + bias, + residual (read back from the
//     rows), (hi, lo) split, store.  [First version: every wave kept PARTIAL fc.3 sums of all eight output blocks over its own 64 hidden
//     channels and the partial sums met in LDS in two rounds -- 256 KB written and read, two more barriers: 20k of the 41k cycles of a
//     workgroup (scripts/trace_mlp_small.py).]
//   * weight fragments come straight from the big kernel's fragment-major stream (every fragment is 1 KiB contiguous, so a lane's
//     16 bytes sit at fragment base + 16 lane: one coalesced global_load_dwordx4 per fragment, no LDS staging -- no two waves share
//     a fragment), a few k-steps ahead of their MFMAs.
// Bound: the 1.5 MB weight stream every workgroup pulls from L2 (~64 B/clk per CU).
// 128-d (the reference's SIFT / HardNet descriptor family): 8 hidden blocks = one per wave; the 4 output blocks x 2 halves of the hidden
// dimension = 8 waves, the two partial sums of an output block meet in LDS (16 KB).  384 KB of weights per workgroup instead of 1.5 MB.
constexpr int SM_T = 32;                       // tokens per workgroup

template <int D>
__global__ __launch_bounds__(512) void mlp_small_kernel(MlpFusedArgs g) {
    static_assert(D == 256 || D == 128, "256-d: 8 waves x 2 hidden blocks, 8 output blocks; 128-d: 8 waves x 1 hidden block, 4 output blocks x 2 hidden halves");
    constexpr int SM_ROW = 4 * D * 2 + 16;     // bytes of one padded LDS row: 4D halves + 16
    constexpr int G0 = 2 * D / 32;             // k-groups of fc.0 = hidden blocks (16 / 8)
    constexpr int NJ = G0 / 8;                 // hidden blocks per wave in fc.0 (2 / 1)
    constexpr int NOB = D / 32;                // output blocks (8 / 4)
    constexpr int HB2 = G0 / 2, SPP = G0 + (D == 256 ? 8 : 4);      // hidden blocks per half, stages per pass of the big kernel's stream
    constexpr int KS0 = 2 * G0;                // k-steps of fc.0 (32 / 16)
    constexpr int NS3 = G0 * NOB / 8;          // fc.3 steps per wave (one hidden block, both k-steps, each): 16 / 4
    __shared__ __attribute__((aligned(16))) char smem[SM_T * SM_ROW + G0 * 2 * 2 * 1024];      // the token tile (66 / 33 KB) + the hidden activation as B fragments (64 / 32 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * SM_T;
    char* const rows = reinterpret_cast<char*>(g.XO);
    const int64_t ldb = g.ld * 2;              // row stride in bytes
#if OG_MLP_TRACE
    unsigned ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define OG_ST(i_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[i_] = (unsigned)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define OG_ST(i_) do {} while (0)
#endif
    OG_ST(0);

    // fragment addresses in the big kernel's stream (og_pack_mlp_stream): hidden block hb = HB2 a + 4 q + i; this wave's are NJ wave + j
    // Weight fragments: inline-asm loads (scalar base + 16 lane) counted by hand -- left to the compiler every load sank down to its
    // use and each k-step waited for a full L2 round trip.  All asm statements carry a memory clobber, so no compiler-issued
    // vector-memory operation moves across them; the compiler's own loads (token rows, bias) are all issued AFTER the first fragments
    // and waited for by the compiler before the main loop, where they could only make a counted wait stricter.
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    auto sbase = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    auto ldfrag = [&](f16x8& dst, const char* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane16), "s"(base) : "memory"); };
    auto frag0 = [&](f16x8& dst, int j, int kg, int t, int part) {       // W0' rows of hidden block NJ wave + j, k-step (kg, t)
        const int hb = NJ * wave + j, a = hb / HB2, q = (hb % HB2) >> 2, i = hb & 3;
        ldfrag(dst, sbase(g.wstream + (int64_t)(SPP * q + kg) * WSTAGE + ((((a * 2 + t) * 4 + i) * 2 + part) << 10)));
    };
    // "all but the n youngest loads have landed" (n a multiple of 4); the operands tie the wait to the registers it releases
    auto wait4 = [](int n, f16x8& r0, f16x8& r1, f16x8& r2, f16x8& r3) {
        if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
        else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
        else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
    };
    auto wait2 = [](int n, f16x8& r0, f16x8& r1) {      // n in {0, 2, .. 14}
        switch (n) {
            case 14: asm volatile("s_waitcnt vmcnt(14)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(r0), "+v"(r1) :: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1) :: "memory"); break;
        }
    };
    constexpr int FPK = 2 * NJ;                // fragments per k-step
    constexpr int PF = D == 256 ? 5 : 8;       // k-steps of W0' fragments in flight: 20 / 16 KB per wave
    f16x8 wf[PF][FPK];
#pragma unroll
    for (int ks = 0; ks < PF; ++ks)
#pragma unroll
        for (int f = 0; f < FPK; ++f) frag0(wf[ks][f], f >> 1, ks >> 1, ks & 1, f & 1);

    // ---- token tile -> LDS: thread (row tid >> 4, 16-byte column tid & 15 + 16 j) ----
    {
        int r = t0 + (tid >> 4);
        if (r > g.M - 1) r = g.M - 1;          // rows past the matrix are clamped (computed, never stored)
        const char* src = rows + (int64_t)r * ldb + (tid & 15) * 16;
        char* dst = smem + (tid >> 4) * SM_ROW + (tid & 15) * 16;
        og_u32x4 v[D / 32];
#pragma unroll
        for (int j = 0; j < D / 32; ++j) v[j] = *reinterpret_cast<const og_u32x4*>(src + j * 256);
#pragma unroll
        for (int j = 0; j < D / 32; ++j) *reinterpret_cast<og_u32x4*>(dst + j * 256) = v[j];
    }
    const float sc0 = g.scales_dev ? g.scales_dev[0] : g.scale, sc3 = g.scales_dev ? g.scales_dev[1] : g.scale;
    const float is0 = 1.f / sc0;

    // ================= fc.0: acc0[j] = b0' / s0 + W0'[hidden block NJ wave + j] . [x ; O], KS0 k-steps =================
    f32x16 acc0[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(g.b0 + 32 * (NJ * wave + j) + 8 * qq + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[j][4 * qq + e] = b[e] * is0;
        }
    OG_ST(1);
    __syncthreads();                           // the token tile is in LDS
    OG_ST(2);
    const char* const xrow = smem + l31 * SM_ROW + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
        const int kg = ks >> 1, t = ks & 1, slot = ks % PF;
        const f16x8 xh = *reinterpret_cast<const f16x8*>(xrow + kg * 128 + t * 32);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(xrow + kg * 128 + t * 32 + 64);
        const int ysteps = KS0 - 1 - ks < PF - 1 ? KS0 - 1 - ks : PF - 1;      // k-steps younger than this one still in flight
        if constexpr (NJ == 2) {
            wait4(4 * ysteps, wf[slot][0], wf[slot][1], wf[slot][2], wf[slot][3]);
            // the two accumulator chains alternate (fragment 2 j + part: part 0 = hi, 1 = lo)
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][1], xh, acc0[0], 0, 0, 0);
            acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][3], xh, acc0[1], 0, 0, 0);
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xl, acc0[0], 0, 0, 0);
            acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][2], xl, acc0[1], 0, 0, 0);
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xh, acc0[0], 0, 0, 0);
            acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][2], xh, acc0[1], 0, 0, 0);
        } else {
            wait2(2 * ysteps, wf[slot][0], wf[slot][1]);
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][1], xh, acc0[0], 0, 0, 0);
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xl, acc0[0], 0, 0, 0);
            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xh, acc0[0], 0, 0, 0);
        }
        if (ks + PF < KS0) {
#pragma unroll
            for (int f = 0; f < FPK; ++f) frag0(wf[slot][f], f >> 1, (ks + PF) >> 1, (ks + PF) & 1, f & 1);
        }
    }

    OG_ST(3);
    char* const hid = smem + SM_T * SM_ROW;
    // ================= fc.3.  256-d: wave w owns OUTPUT block w over all 512 hidden channels (16 steps of two k-steps, two accumulator chains);
    //                   128-d: wave w owns output block w & 3 over hidden blocks 4 (w >> 2) .. + 3 (4 steps), the two halves meet in LDS =================
    // W3' fragment (output block i, hidden block hb = HB2 a' + 4 q' + j', k-step t), og_pack_mlp_stream:
    //   256-d: stage 24 q' + 16 + 2 j' + t, fragment (a' 8 + i) 2 + part;  128-d: stage 8 + j', fragment ((a' 2 + t) 4 + i) 2 + part
    const int ob = wave & (NOB - 1);           // this wave's output block
    const int hb0 = D == 256 ? 0 : 4 * (wave >> 2);      // ... and its first hidden block
    auto frag3 = [&](f16x8& dst, int hb, int t, int part) {
        if constexpr (D == 256)
            ldfrag(dst, sbase(g.wstream + (int64_t)(24 * ((hb >> 2) & 1) + 16 + 2 * (hb & 3) + t) * WSTAGE + ((((hb >> 3) * 8 + ob) * 2 + part) << 10)));
        else
            ldfrag(dst, sbase(g.wstream + (int64_t)(8 + (hb & 3)) * WSTAGE + (((((hb >> 2) * 2 + t) * 4 + ob) * 2 + part) << 10)));
    };
    constexpr int PF3 = 4;                     // steps of two k-steps (4 fragments) in flight
    f16x8 w3f[PF3][4];
    auto frag3_step = [&](f16x8 (&dst)[4], int n) {          // step n = hidden block hb0 + n, k-steps t = 0, 1: (t0 hi, t0 lo, t1 hi, t1 lo)
#pragma unroll
        for (int f = 0; f < 4; ++f) frag3(dst[f], hb0 + n, f >> 1, f & 1);
    };
#pragma unroll
    for (int n = 0; n < PF3; ++n) frag3_step(w3f[n], n);
    // bias and residual of this wave's block for the epilogue (the compiler's loads: younger than the fragments above, see the note at the top)
    const int tok = t0 + l31;
    const bool live = tok < g.M;
    char* const orow = rows + (int64_t)(live ? tok : 0) * ldb + ob * 128;        // hl32 row: 64 B hi | 64 B lo per 32 channels
    f32x4 bias3[4];
    uint2 rxh[4], rxl[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int ch = 8 * qq + 4 * hi;        // first of this lane's 4 consecutive channels of register group qq
        bias3[qq] = *reinterpret_cast<const f32x4*>(g.b3 + 32 * ob + ch);
        rxh[qq] = *reinterpret_cast<const uint2*>(orow + ch * 2);
        rxl[qq] = *reinterpret_cast<const uint2*>(orow + ch * 2 + 64);
    }
    // ================= hidden activation -> (hi, lo) B fragments (element e of k-step t = accumulator register 8 t + e, as the stream is
    //                   packed), handed to the other waves through LDS: [hidden block][t][part][lane] x 16 B, 64 KB =================
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                og_u32x4 h4, l4;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r0 = 8 * t + 4 * h;
                    float c0 = fmaxf(acc0[j][r0] * sc0, 0.f), c1 = fmaxf(acc0[j][r0 + 1] * sc0, 0.f);
                    float c2 = fmaxf(acc0[j][r0 + 2] * sc0, 0.f), c3 = fmaxf(acc0[j][r0 + 3] * sc0, 0.f);
                    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));      // og_split4 must see ONE rounded product
                    unsigned ha, la, hb, lb;
                    og_split4(c0, c1, c2, c3, ha, la, hb, lb);
                    h4[2 * h] = ha; h4[2 * h + 1] = hb; l4[2 * h] = la; l4[2 * h + 1] = lb;
                }
                char* d = hid + ((((NJ * wave + j) * 2 + t) * 2) << 10) + lane * 16;
                *reinterpret_cast<og_u32x4*>(d) = h4;
                *reinterpret_cast<og_u32x4*>(d + 1024) = l4;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // epilogue operands and the first fragments are here; from now on the counts are exact again
    OG_ST(4);
    __syncthreads();                           // the hidden fragments of all waves are in LDS
    OG_ST(5);
    f32x16 acc3[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[c][r] = 0.f;
    const char* const hrd = hid + lane * 16 + ((hb0 * 4) << 10);
#pragma unroll
    for (int n = 0; n < NS3; ++n) {
        const int slot = n % PF3;
        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(hrd + (((n * 2 + 0) * 2) << 10)), bl0 = *reinterpret_cast<const f16x8*>(hrd + (((n * 2 + 0) * 2 + 1) << 10));
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(hrd + (((n * 2 + 1) * 2) << 10)), bl1 = *reinterpret_cast<const f16x8*>(hrd + (((n * 2 + 1) * 2 + 1) << 10));
        const int younger = n < PF3 ? 0 : (NS3 - 1 - n < PF3 - 1 ? 4 * (NS3 - 1 - n) : 4 * (PF3 - 1));      // the first PF3 steps were waited for above
        if (n >= PF3) wait4(younger, w3f[slot][0], w3f[slot][1], w3f[slot][2], w3f[slot][3]);
        acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][1], bh0, acc3[0], 0, 0, 0);
        acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][3], bh1, acc3[1], 0, 0, 0);
        acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][0], bl0, acc3[0], 0, 0, 0);
        acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][2], bl1, acc3[1], 0, 0, 0);
        acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][0], bh0, acc3[0], 0, 0, 0);
        acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[slot][2], bh1, acc3[1], 0, 0, 0);
        if (n + PF3 < NS3) frag3_step(w3f[slot], n + PF3);
    }
    OG_ST(6);
    if constexpr (D == 128) {
        // the two hidden halves of an output block meet: waves 4..7 leave their sums in the (dead) token-tile area, waves 0..3 add them and finish
        char* const red = smem + ((wave & 3) << 12) + lane * 16;       // [output block][register group][lane] x 16 B: 4 KB per block
        if (wave >= 4) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc3[0][4 * qq + e] + acc3[1][4 * qq + e];
                *reinterpret_cast<f32x4*>(red + (qq << 10)) = v;
            }
        }
        __syncthreads();
        if (wave >= 4) return;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + (qq << 10));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc3[0][4 * qq + e] += v[e];
        }
    }

    // ================= epilogue: x <- (acc / S3 + b3') + x, written back as (hi, lo) halves: 8 bytes per register group and plane =================
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const _Float16* xhh = reinterpret_cast<const _Float16*>(&rxh[qq]);
            const _Float16* xlh = reinterpret_cast<const _Float16*>(&rxl[qq]);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (acc3[0][4 * qq + e] + acc3[1][4 * qq + e]) * sc3;
                v[e] = v[e] + bias3[qq][e];
                v[e] = v[e] + ((float)xhh[e] + (float)xlh[e]);
                asm volatile("" : "+v"(v[e]));
            }
            unsigned ha, la, hb, lb;
            og_split4(v[0], v[1], v[2], v[3], ha, la, hb, lb);
            if (live) {
                char* const px = orow + (8 * qq + 4 * hi) * 2;
                *reinterpret_cast<uint2*>(px) = make_uint2(ha, hb);
                *reinterpret_cast<uint2*>(px + 64) = make_uint2(la, lb);
            }
        }
    }
#if OG_MLP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OG_ST(7);
    if (lane == 0 && blockIdx.x < OG_MT_BLOCKS)
        for (int i = 0; i < 8; ++i) og_mlp_trace_buf[blockIdx.x][wave][0][i] = ts[i];
#endif
#undef OG_ST
}


// =================================================================================================================================
// proj_small_kernel -- a 1x1 conv of the residual stream (the q / k / v projections) for FEW token rows, in the style of mlp_small_kernel:
// a workgroup owns 32 token rows, copies their x halves (32 KB of hl32 rows) to LDS, and its 8 waves split the OUTPUT blocks of 32
// channels (wave w: blocks c0 + w, c0 + w + 8, c0 + w + 16 below c1), each over all 16 k-steps, weight fragments straight from a
// fragment-major stream (og_pack_proj_stream).  Output: (hi, lo) planes.  The 128-token tile GEMM it replaces takes 17.5 us per launch at
// 2048 or 8192 rows (profiles/r04_small_batch_kernel_stats_B*.csv: 36 launches per step).  Workgroups whose first row is below
// `split_row` use the block range [a0, a1), the others [b0, b1) -- the cross layer's "q of image 0, q | k | v of image 1" in one launch.
template <int K> constexpr int PS_ROW = K * 2 * 2 + 16;       // bytes of one padded LDS row: the x half of an hl32 row (2K halves) + 16; K = D = 256 or 128

struct ProjSmallArgs {
    const _Float16* X; int64_t ld;             // [M] hl32 rows, the first 2D halves = x
    int M;
    const char* wstream;                       // og_pack_proj_stream
    const float* bias;                         // [N]
    const float* scale_dev;                    // DEVICE: 1 / pre-scale of the matrix
    _Float16* Ch; _Float16* Cl; int64_t ldc;   // output planes [M][ldc]
    int split_row, a0, a1, b0, b1;             // block ranges (units of 32 output channels)
    int parts, bpp;                            // very few rows: the block range of a token tile dealt to `parts` workgroups, bpp blocks each
};

template <int NB, int K>
__device__ __forceinline__ void proj_small_run(const ProjSmallArgs& g, const char* smem, int wave, int lane, int cb0, int t0) {
    constexpr int KS = K / 16;                 // k-steps
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    auto sbase = [](const char* p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi32 << 32) | lo));
    };
    // fragment (block i, k-step ks, part): ((i * KS + ks) * 2 + part) KiB
    auto frag = [&](f16x8& dst, int b, int ks, int part) {
        const char* base = sbase(g.wstream + ((((int64_t)(cb0 + wave + 8 * b) * KS + ks) * 2 + part) << 10));
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane16), "s"(base) : "memory");
    };
    constexpr int PF = 4, FPS = 2 * NB;        // k-steps in flight, fragments per k-step
    f16x8 wf[PF][FPS];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tile copy's loads are done; from here on the counts are mine
#pragma unroll
    for (int ks = 0; ks < PF; ++ks)
#pragma unroll
        for (int f = 0; f < FPS; ++f) frag(wf[ks][f], f >> 1, ks, f & 1);
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const char* const xrow = smem + l31 * PS_ROW<K> + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int slot = ks % PF;
        const f16x8 xh = *reinterpret_cast<const f16x8*>(xrow + (ks >> 1) * 128 + (ks & 1) * 32);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(xrow + (ks >> 1) * 128 + (ks & 1) * 32 + 64);
        const int ysteps = KS - 1 - ks < PF - 1 ? KS - 1 - ks : PF - 1;       // k-steps younger than this one still in flight
        // "all but the ysteps * FPS youngest loads have landed"
        if constexpr (NB == 1) {
            if (ysteps == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]) :: "memory");
            else if (ysteps == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]) :: "memory");
            else if (ysteps == 1) asm volatile("s_waitcnt vmcnt(2)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]) :: "memory");
        } else if constexpr (NB == 2) {
            if (ysteps == 3) asm volatile("s_waitcnt vmcnt(12)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]) :: "memory");
            else if (ysteps == 2) asm volatile("s_waitcnt vmcnt(8)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]) :: "memory");
            else if (ysteps == 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]) :: "memory");
        } else {
            if (ysteps == 3) asm volatile("s_waitcnt vmcnt(18)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]), "+v"(wf[slot][4]), "+v"(wf[slot][5]) :: "memory");
            else if (ysteps == 2) asm volatile("s_waitcnt vmcnt(12)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]), "+v"(wf[slot][4]), "+v"(wf[slot][5]) :: "memory");
            else if (ysteps == 1) asm volatile("s_waitcnt vmcnt(6)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]), "+v"(wf[slot][4]), "+v"(wf[slot][5]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf[slot][0]), "+v"(wf[slot][1]), "+v"(wf[slot][2]), "+v"(wf[slot][3]), "+v"(wf[slot][4]), "+v"(wf[slot][5]) :: "memory");
        }
        // pass-major: the accumulator chains alternate
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][2 * b + 1], xh, acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][2 * b], xl, acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][2 * b], xh, acc[b], 0, 0, 0);
        if (ks + PF < KS) {
#pragma unroll
            for (int f = 0; f < FPS; ++f) frag(wf[slot][f], f >> 1, ks + PF, f & 1);
        }
    }
    // ---- epilogue: acc / S + bias as (hi, lo) halves, 8 bytes per register group and plane ----
    const float sc = g.scale_dev[0];
    const int tok = t0 + l31;
    if (tok < g.M) {
#pragma clang fp contract(off)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col0 = 32 * (cb0 + wave + 8 * b);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int ch = col0 + 8 * qq + 4 * hi;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + ch);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[b][4 * qq + e] * sc;
                    v[e] = v[e] + bv[e];
                    asm volatile("" : "+v"(v[e]));
                }
                unsigned ha, la, hb, lb;
                og_split4(v[0], v[1], v[2], v[3], ha, la, hb, lb);
                *reinterpret_cast<uint2*>(g.Ch + (int64_t)tok * g.ldc + ch) = make_uint2(ha, hb);
                *reinterpret_cast<uint2*>(g.Cl + (int64_t)tok * g.ldc + ch) = make_uint2(la, lb);
            }
        }
    }
}

template <int K>
__global__ __launch_bounds__(512) void proj_small_kernel(ProjSmallArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[SM_T * PS_ROW<K>];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (the division runs on the vector ALU: back to SGPRs at once -- the fragment loads below take their base address in SGPRs through inline asm,
    //  and a v_readfirstlane right in front of such a load is a VALU-writes-SGPR -> VMEM hazard the compiler does not see: the first version of
    //  this split faulted on a stale base)
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / g.parts));
    const int t0 = tile * SM_T, part = (int)blockIdx.x - tile * g.parts;
    const int r0 = t0 < g.split_row ? g.a0 : g.b0, r1 = t0 < g.split_row ? g.a1 : g.b1;
    const int cb0 = __builtin_amdgcn_readfirstlane(r0 + part * g.bpp);
    const int cb1 = __builtin_amdgcn_readfirstlane(cb0 + g.bpp < r1 ? cb0 + g.bpp : r1);
    if (cb1 <= cb0) return;
    // ---- x halves of the token rows -> LDS: thread (row tid >> 4, 16-byte column tid & 15 + 16 j), 4K bytes per row ----
    {
        int r = t0 + (tid >> 4);
        if (r > g.M - 1) r = g.M - 1;
        const char* src = reinterpret_cast<const char*>(g.X) + (int64_t)r * g.ld * 2 + (tid & 15) * 16;
        char* dst = smem + (tid >> 4) * PS_ROW<K> + (tid & 15) * 16;
        og_u32x4 v[K / 64];
#pragma unroll
        for (int j = 0; j < K / 64; ++j) v[j] = *reinterpret_cast<const og_u32x4*>(src + j * 256);
#pragma unroll
        for (int j = 0; j < K / 64; ++j) *reinterpret_cast<og_u32x4*>(dst + j * 256) = v[j];
    }
    __syncthreads();
    const int mine = (cb1 - cb0 - wave + 7) / 8;             // blocks cb0 + wave + 8 b < cb1 (wave-uniform)
    if (mine >= 3) proj_small_run<3, K>(g, smem, wave, lane, cb0, t0);
    else if (mine == 2) proj_small_run<2, K>(g, smem, wave, lane, cb0, t0);
    else if (mine == 1) proj_small_run<1, K>(g, smem, wave, lane, cb0, t0);
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------
bool og_mlp_fused_supported(int D) { return D == 256 || D == 128; }

size_t og_mlp_stream_bytes(int D) { return og_mlp_fused_supported(D) ? (size_t)6 * D * D * 4 : 0; }      // 4D^2 (W0') + 2D^2 (W3') (hi, lo) pairs

bool og_mlp_fused_enabled(int D) {
    static const bool on = [] { const char* e = getenv("OG_MLP_FUSED"); return !e || atoi(e) != 0; }();   // experiments: 0 = fc.0 and fc.3 as two GEMM launches
    return on && og_mlp_fused_supported(D);
}

// Fragment-major weight stream of mlp_fused_kernel.  W0 [2D][2D], W3 [D][2D] row-major double (already folded), written as (hi, lo)
// halves of S0 w / S3 w (power-of-two pre-scales, 256 unless a weight would leave binary16).  256-d: 48 stages of 32 fragments (1 KiB = 64 lanes x 8 halves; a lo fragment
// follows its hi fragment), as below; 128-d: 12 stages -- 8 fc.0 stages (hidden block 4 a + i), then one fc.3 stage per hidden block j of a half,
// fragment ((a * 2 + t) * 4 + i) * 2 + part (both k-steps, output block i).  256-d, quarter q:
//   fc.0 stage 24 q + kg (k-group kg):   fragment ((a * 2 + t) * 4 + i) * 2 + part, lane l = (rho = l & 31, h = l >> 5), element e:
//          W0[32 (8a + 4q + i) + rho][32 kg + 16 t + 8 h + e]                                   (a = hidden half, i = block of the quarter)
//   fc.3 stage 24 q + 16 + 2 j + t:      fragment (a * 8 + i) * 2 + part (i = output block):
//          W3[32 i + rho][32 (8a + 4q + j) + 16 t + 8 (e >> 2) + 4 h + (e & 3)]                   (the accumulator-register order, above)
// Returns false when a scaled weight does not fit binary16.
bool og_pack_mlp_stream(int D, const double* W0, const double* W3, void* out, double S0, double S3) {
    if (!og_mlp_fused_supported(D)) return false;
    const int D2 = 2 * D, G0 = D2 / 32, HB2 = G0 / 2, NPASS = HB2 / 4, NOB = D / 32;
    const int SPP = G0 + (D == 256 ? 8 : 4);            // stages per pass: G0 fc.0 stages + the fc.3 stages of its 4 hidden blocks per half
    _Float16* o = (_Float16*)out;
    bool ok = true;
    auto put = [&](int64_t stage, int f, int l, int e, double w, double S) {
        w *= S;
        if (!(fabs(w) <= 65504.0)) { ok = false; w = 0.0; }
        const _Float16 hi = (_Float16)w;
        _Float16* base = o + stage * (32768 / 2) + (int64_t)f * 512 + l * 8 + e;     // fragment f: 1 KiB = 512 halves
        base[0] = hi;
        base[512] = (_Float16)(w - (double)hi);                                        // the lo fragment follows the hi fragment
    };
    for (int q = 0; q < NPASS; ++q) {
        for (int kg = 0; kg < G0; ++kg)
            for (int a = 0; a < 2; ++a)
                for (int t = 0; t < 2; ++t)
                    for (int i = 0; i < 4; ++i)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e)
                                put(SPP * q + kg, ((a * 2 + t) * 4 + i) * 2, l, e,
                                    W0[(int64_t)(32 * (HB2 * a + 4 * q + i) + (l & 31)) * D2 + 32 * kg + 16 * t + 8 * (l >> 5) + e], S0);
        for (int j = 0; j < 4; ++j)
            for (int t = 0; t < 2; ++t)
                for (int a = 0; a < 2; ++a)
                    for (int i = 0; i < NOB; ++i)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                // 256-d: one stage per (hidden block j, k-step t), 8 output blocks per half; 128-d: one stage per hidden block, (t, 4 output blocks) per half
                                const int64_t stage = D == 256 ? SPP * q + G0 + 2 * j + t : SPP * q + G0 + j;
                                const int frag = D == 256 ? (a * 8 + i) * 2 : ((a * 2 + t) * 4 + i) * 2;
                                put(stage, frag, l, e,
                                    W3[(int64_t)(32 * i + (l & 31)) * D2 + 32 * (HB2 * a + 4 * q + j) + 16 * t + 8 * (e >> 2) + 4 * (l >> 5) + (e & 3)], S3);
                            }
    }
    return ok;
}

// Few token rows: 32-token workgroups whose waves split the hidden dimension (mlp_small_kernel).  Up to 8192 rows = 256 workgroups, one
// round of the chip; above that the 128-token tiles of mlp_fused_kernel stream the weights four times less often.  OG_MLP_SMALL=0 / 1 forces.
bool og_mlp_small_wanted(int M) {
    static const int mode = [] { const char* e = getenv("OG_MLP_SMALL"); return e ? atoi(e) : -1; }();
    if (mode >= 0) return mode != 0;
    return M <= 8192;
}

int og_launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t stream) {
    if (!a.XO || !a.wstream || !a.b0 || !a.b3 || a.M <= 0) return OG_E_INVALID;
    if (!og_mlp_fused_supported(D)) return OG_E_SHAPE;
    if (((uintptr_t)a.XO & 15) || ((uintptr_t)a.wstream & 15) || (a.ld & 7) || a.ld < 4 * (int64_t)D) return OG_E_ALIGN;
    if ((int64_t)MT * a.ld * 2 >= (int64_t)1 << 31) return OG_E_SHAPE;              // 32-bit lane offsets are relative to the TILE's first row (baseX is 64-bit)
    if (!(a.scale != 0.f) || !std::isfinite(a.scale)) return OG_E_INVALID;
    if (og_mlp_small_wanted(a.M) && !(((uintptr_t)a.b0 | (uintptr_t)a.b3) & 15)) {      // (mlp_small_kernel reads the biases as 16-byte vectors)
        if (D == 256) hipLaunchKernelGGL(mlp_small_kernel<256>, dim3((a.M + SM_T - 1) / SM_T), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL(mlp_small_kernel<128>, dim3((a.M + SM_T - 1) / SM_T), dim3(512), 0, stream, a);
        return og_launch_status();
    }
    const int tiles = (a.M + MT - 1) / MT;
    if (D == 256) hipLaunchKernelGGL(mlp_fused_kernel<256>, dim3(tiles), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL(mlp_fused_kernel<128>, dim3(tiles), dim3(512), 0, stream, a);
    return og_launch_status();
}

#if OG_MLP_TRACE
extern "C" int og_debug_mlp_trace(void* host_dst, size_t bytes) {
    if (bytes > sizeof(og_mlp_trace_buf)) bytes = sizeof(og_mlp_trace_buf);
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(og_mlp_trace_buf), bytes);
}
#endif

// Stage entry (include/openglue_amd.h): the message MLP of one GNN layer on M token rows of [x | O] hl32 rows, in place.
extern "C" size_t og_mlp_block_stream_bytes(int32_t D) { return og_mlp_stream_bytes(D); }

extern "C" int og_mlp_block_pack(int32_t D, const float* W0, const float* W3, void* stream_host) {
    if (!W0 || !W3 || !stream_host) return OG_E_INVALID;
    if (!og_mlp_fused_supported(D)) return OG_E_SHAPE;
    const int64_t D2 = 2 * D;
    double* w0 = (double*)malloc(sizeof(double) * D2 * D2);
    double* w3 = (double*)malloc(sizeof(double) * D * D2);
    if (!w0 || !w3) { free(w0); free(w3); return OG_E_INVALID; }
    for (int64_t i = 0; i < D2 * D2; ++i) w0[i] = W0[i];
    for (int64_t i = 0; i < D * D2; ++i) w3[i] = W3[i];
    const bool ok = og_pack_mlp_stream(D, w0, w3, stream_host);
    free(w0); free(w3);
    return ok ? 0 : OG_E_RANGE;
}

extern "C" int og_mlp_block(int32_t D, void* xo_rows, int64_t ld, int32_t M, const void* stream_dev, const float* b0, const float* b3, void* stream) {
    og_clear_status();
    MlpFusedArgs a{};
    a.XO = (_Float16*)xo_rows; a.ld = ld; a.M = M; a.wstream = (const char*)stream_dev; a.b0 = b0; a.b3 = b3; a.scale = (float)(1.0 / OG_W_SCALE);
    return og_launch_mlp_fused(a, D, (hipStream_t)stream);
}

// ---- the small-batch projection (proj_small_kernel) ----
// Fragment-major stream of an [N][K] matrix (K = 256 or 128, N a multiple of 32), standard k order: fragment ((i * K / 16 + ks) * 2 + part), lane l =
// (rho = l & 31, h = l >> 5), element e = S w[32 i + rho][16 ks + 8 h + e] as (hi, lo) halves (the lo fragment follows the hi fragment).
size_t og_proj_stream_bytes(int N, int K) { return ((K == 256 || K == 128) && N > 0 && N % 32 == 0) ? (size_t)N * K * 4 : 0; }

bool og_pack_proj_stream(int N, int K, const double* W, void* out, double S) {
    if (!og_proj_stream_bytes(N, K)) return false;
    _Float16* o = (_Float16*)out;
    bool ok = true;
    for (int i = 0; i < N / 32; ++i)
        for (int ks = 0; ks < K / 16; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    double w = W[(int64_t)(32 * i + (l & 31)) * K + 16 * ks + 8 * (l >> 5) + e] * S;
                    if (!(fabs(w) <= 65504.0)) { ok = false; w = 0.0; }
                    const _Float16 hi = (_Float16)w;
                    _Float16* base = o + ((int64_t)(i * (K / 16) + ks) * 2) * 512 + l * 8 + e;
                    base[0] = hi;
                    base[512] = (_Float16)(w - (double)hi);
                }
    return ok;
}

// Rows [0, M) of X (hl32 rows, x half) times columns [32 a0, 32 a1) of the packed matrix for rows below split_row, [32 b0, 32 b1) for the
// others (split_row must be a multiple of 32 unless it is 0 or >= M); planes Ch / Cl [M][ldc].
int og_launch_proj_small(const _Float16* X, int64_t ld, int M, int K, const char* wstream, const float* bias, const float* scale_dev,
                         _Float16* Ch, _Float16* Cl, int64_t ldc, int split_row, int a0, int a1, int b0, int b1, hipStream_t stream) {
    if (!X || !wstream || !bias || !scale_dev || !Ch || !Cl || M <= 0) return OG_E_INVALID;
    if (K != 256 && K != 128) return OG_E_SHAPE;
    if (((uintptr_t)X & 15) || ((uintptr_t)wstream & 15) || ((uintptr_t)bias & 15) || (ld & 7) || (ldc & 3) || ((uintptr_t)Ch & 7) || ((uintptr_t)Cl & 7)) return OG_E_ALIGN;
    if (a0 < 0 || b0 < 0 || a1 < a0 || b1 < b0 || ld < 2 * K) return OG_E_SHAPE;
    if (split_row > 0 && split_row < M && (split_row % SM_T)) return OG_E_SHAPE;
    // One or two pairs: a token tile's output blocks go to several workgroups, 8 blocks (one per wave) each -- nothing to reduce, and a workgroup
    // streams 256 KB of weights instead of the whole matrix (the per-CU weight stream is what bounds these kernels).
    const int tiles = (M + SM_T - 1) / SM_T, nblk = (a1 - a0) > (b1 - b0) ? (a1 - a0) : (b1 - b0);
    int parts = 1, bpp = nblk > 0 ? nblk : 1;
    static const bool split_on = [] { const char* e = getenv("OG_PROJ_PARTS"); return !(e && e[0] == '0'); }();      // OG_PROJ_PARTS=0: one workgroup per token tile
    if (split_on && nblk > 8 && tiles * ((nblk + 7) / 8) <= 256) { parts = (nblk + 7) / 8; bpp = 8; }
    if (nblk > 24) { parts = (nblk + 7) / 8; bpp = 8; }          // a workgroup covers at most 3 blocks per wave: wider ranges are always dealt out
    ProjSmallArgs g{X, ld, M, wstream, bias, scale_dev, Ch, Cl, ldc, split_row, a0, a1, b0, b1, parts, bpp};
    if (K == 256) hipLaunchKernelGGL(proj_small_kernel<256>, dim3(tiles * parts), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL(proj_small_kernel<128>, dim3(tiles * parts), dim3(512), 0, stream, g);
    return og_launch_status();
}

// ---- the batch projection (proj_stream_kernel) ----
// Fragment-major stream of an [N][K] matrix, N a multiple of 128: stage (sp, kq) = sp * (K / 64) + kq covers the 4 output blocks of super-pair sp
// and the k-steps 4 kq .. 4 kq + 3; fragment a * 16 + (g * 2 + j) * 2 + part (wave half a owns blocks 2a + j of the super-pair, g = k-step of the
// stage), lane l = (rho = l & 31, h = l >> 5), element e = S w[32 (4 sp + 2 a + j) + rho][16 (4 kq + g) + 8 h + e] as (hi, lo) halves.
size_t og_proj_stream_big_bytes(int N, int K) { return ((K == 256 || K == 128) && N > 0 && N % 128 == 0) ? (size_t)N * K * 4 : 0; }

bool og_pack_proj_stream_big(int N, int K, const double* W, void* out, double S) {
    if (!og_proj_stream_big_bytes(N, K)) return false;
    _Float16* o = (_Float16*)out;
    const int SPS = K / 64;
    bool ok = true;
    for (int sp = 0; sp < N / 128; ++sp)
        for (int kq = 0; kq < SPS; ++kq)
            for (int a = 0; a < 2; ++a)
                for (int gq = 0; gq < 4; ++gq)
                    for (int j = 0; j < 2; ++j)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                double w = W[(int64_t)(32 * (4 * sp + 2 * a + j) + (l & 31)) * K + 16 * (4 * kq + gq) + 8 * (l >> 5) + e] * S;
                                if (!(fabs(w) <= 65504.0)) { ok = false; w = 0.0; }
                                const _Float16 hi = (_Float16)w;
                                _Float16* base = o + (int64_t)(sp * SPS + kq) * (WSTAGE / 2) + (int64_t)(a * 16 + (gq * 2 + j) * 2) * 512 + l * 8 + e;
                                base[0] = hi;
                                base[512] = (_Float16)(w - (double)hi);
                            }
    return ok;
}

// og_forward: launches of more than 8192 rows take proj_stream_kernel at K = 128 (OG_PROJ_STREAM=0 / 1 forces for both widths).  Measured in one
// gpurun call (profiles/r05_c_proj_micro.log, r05_b_bench_proj_stream_ab.jsonl): K = 128 -- self 37.8 us against 52.6 for the 128-token tile GEMM,
// cross 29.2 / 40.7, kv 16.8 / 21.3; C4 13.29 -> 13.12 ms, S128 12.27 -> 12.12 ms per step.  K = 256 -- self 102 us against 104.5 for the 256-tile
// GEMM, but kv 42.6 / 33.5 and the cross launch 87.6 / ~56: C2 8.92 -> 8.99 ms, so the tile GEMMs keep the 256-d batches.
// `full`: the launch produces every column of the matrix for all its rows (a self layer's q | k | v).  OG_PROJ_STREAM=2 (experiment): at K = 256 the stream
// kernel takes those launches only (95 against 104.5 us in the micro-benchmark), the tile GEMMs the cross / kv forms.
// (og_forward packs the batch stream for K = 128 only -- api.hip: packed_layout -- so forcing the K = 256 form reaches the stage entry og_proj_block alone)
bool og_proj_stream_wanted(int M, int K, bool full) {
    static const int mode = [] { const char* e = getenv("OG_PROJ_STREAM"); return e ? atoi(e) : -1; }();
    if (mode == 2) return M > 8192 && (K == 128 || full);
    if (mode >= 0) return mode != 0 && M > 0;
    return M > 8192 && K == 128;
}

// Rows [0, M) of X (hl32 rows, x half) times columns [128 a0, 128 a1) of the packed matrix for rows below split_row, [128 b0, 128 b1) for the
// others (split_row a multiple of 128 unless it is 0 or >= M); planes Ch / Cl [M][ldc].
int og_launch_proj_stream(const _Float16* X, int64_t ld, int M, int K, const char* wstream, const float* bias, const float* scale_dev,
                          _Float16* Ch, _Float16* Cl, int64_t ldc, int split_row, int a0, int a1, int b0, int b1, hipStream_t stream) {
    if (!X || !wstream || !bias || !scale_dev || !Ch || !Cl || M <= 0) return OG_E_INVALID;
    if (K != 256 && K != 128) return OG_E_SHAPE;
    if (((uintptr_t)X & 15) || ((uintptr_t)wstream & 15) || (ld & 7) || (ldc & 63) || ((uintptr_t)Ch & 127) || ((uintptr_t)Cl & 127)) return OG_E_ALIGN;
    if (a0 < 0 || b0 < 0 || a1 < a0 || b1 < b0 || a1 - a0 > 8 || b1 - b0 > 8 || ld < 2 * K) return OG_E_SHAPE;      // (the bias area holds 1024 columns)
    if (split_row > 0 && split_row < M && (split_row % MT)) return OG_E_SHAPE;
    if ((int64_t)M * ld * 2 >= (int64_t)1 << 40) return OG_E_SHAPE;
    ProjStreamArgs g{X, ld, M, wstream, bias, scale_dev, Ch, Cl, ldc, split_row, a0, a1, b0, b1};
    const int tiles = (M + MT - 1) / MT;
    if (K == 256) hipLaunchKernelGGL(proj_stream_kernel<256>, dim3(tiles), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL(proj_stream_kernel<128>, dim3(tiles), dim3(512), 0, stream, g);
    return og_launch_status();
}

// og_proj_block_pack writes the small-batch stream and, when N is a multiple of 128, the batch stream right behind it
extern "C" size_t og_proj_block_stream_bytes(int32_t N, int32_t K) { return og_proj_stream_bytes(N, K) ? og_proj_stream_bytes(N, K) + og_proj_stream_big_bytes(N, K) : 0; }

extern "C" int og_proj_block_pack(int32_t N, int32_t K, const float* W, void* stream_host) {
    if (!W || !stream_host) return OG_E_INVALID;
    if (!og_proj_stream_bytes(N, K)) return OG_E_SHAPE;
    double* w = (double*)malloc(sizeof(double) * (size_t)N * K);
    if (!w) return OG_E_INVALID;
    for (int64_t i = 0; i < (int64_t)N * K; ++i) w[i] = W[i];
    bool ok = og_pack_proj_stream(N, K, w, stream_host, OG_W_SCALE);
    if (og_proj_stream_big_bytes(N, K)) ok &= og_pack_proj_stream_big(N, K, w, (char*)stream_host + og_proj_stream_bytes(N, K), OG_W_SCALE);
    free(w);
    return ok ? 0 : OG_E_RANGE;
}

extern "C" int og_proj_block(const void* x_rows, int64_t ld, int32_t M, int32_t K, int32_t N, const void* stream_dev, const float* bias, const float* inv_scale_dev,
                             void* yh, void* yl, int64_t ldy, int32_t split_row, int32_t a0, int32_t a1, int32_t b0, int32_t b1, void* stream) {
    og_clear_status();
    if (N <= 0 || (N % 32) || a0 < 0 || b0 < 0 || a1 < a0 || b1 < b0 || 32 * (int64_t)(a1 > b1 ? a1 : b1) > N || ldy < N) return OG_E_SHAPE;
    if (!og_proj_stream_bytes(N, K)) return OG_E_SHAPE;
    // batches: the 128-token kernel (its stream sits behind the small one, located by N: ABI v10) when the ranges are whole 128-channel groups
    // (the stage entry of BOTH kernels: above 8192 rows it runs proj_stream_kernel at either width, whatever og_forward prefers)
    static const int ps_mode = [] { const char* e = getenv("OG_PROJ_STREAM"); return e ? atoi(e) : -1; }();
    if ((ps_mode >= 0 ? ps_mode != 0 : M > 8192) && og_proj_stream_big_bytes(N, K) && !(a0 % 4) && !(a1 % 4) && !(b0 % 4) && !(b1 % 4) && a1 - a0 <= 32 && b1 - b0 <= 32 &&
        !(ldy & 63) && !(split_row > 0 && split_row < M && (split_row % MT)) && !((uintptr_t)yh & 127) && !((uintptr_t)yl & 127))
        return og_launch_proj_stream((const _Float16*)x_rows, ld, M, K, (const char*)stream_dev + og_proj_stream_bytes(N, K), bias, inv_scale_dev,
                                     (_Float16*)yh, (_Float16*)yl, ldy, split_row, a0 / 4, a1 / 4, b0 / 4, b1 / 4, (hipStream_t)stream);
    return og_launch_proj_small((const _Float16*)x_rows, ld, M, K, (const char*)stream_dev, bias, inv_scale_dev, (_Float16*)yh, (_Float16*)yl, ldy,
                                split_row, a0, a1, b0, b1, (hipStream_t)stream);
}

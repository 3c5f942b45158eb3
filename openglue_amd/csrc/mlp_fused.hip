// The message MLP of one attentional-GNN layer as ONE kernel (reference attention_gnn.py:43-55 + models/utils.py:48-58):
//
//        x  <-  x + W3' · relu(W0' · [x ; O] + b0') + b3'
//
// (out_proj folded into W0', BatchNorm folded into W3': og_pack_weights).  Round 2 ran fc.0 and fc.3 as two split-f16 GEMM launches
// with the 2D-wide hidden activation H written to and re-read from memory as hl32 rows (134 MB each way per launch at C2, 36 launches
// per step) and two prologue / epilogue rounds per layer.  Here the hidden activation never leaves the register file:
//
//   * a workgroup = 128 tokens, 4 waves, ONE wave per SIMD (512 registers per lane); a wave owns 32 tokens and ALL channels;
//   * MFMA orientation D[channel][token] = W · Xᵀ as in gemm_f16x3.hip.  In the 32x32 accumulator layout a lane owns one token and,
//     per register, one channel -- which is exactly the B-operand layout of the NEXT contraction over those channels (the order of
//     the k index inside an MFMA is free as long as both operands agree): after bias / ReLU / (hi, lo) split the fc.0 accumulators
//     ARE the B fragments of fc.3.  The channel permutation this implies is baked into the packed W3' fragments.
//   * the hidden dimension is processed in two halves of 8 channel blocks (128 accumulator registers) so that fc.0's accumulators
//     (128) plus fc.3's (128) fill the 256 AGPRs and every fragment / address lives in the VGPRs; the token tile is streamed twice
//     (its second read hits the Infinity Cache);
//   * weights are packed FRAGMENT-MAJOR (og_pack_mlp_stream): the 16 bytes lane l feeds to an MFMA sit at fragment base + 16 l, so
//     a fragment is one contiguous 1 KiB both in memory (LDS-DMA source: full lines) and in LDS (ds_read_b128: conflict-free, no
//     swizzle); the whole kernel consumes ONE linear stream of 48 stages x 32 KiB (32 fragments = 48 MFMAs per wave each):
//         pass a in {0, 1}:  16 fc.0 stages (one 32-channel k-group of [x ; O] each)  then  8 fc.3 stages (one hidden block each)
//   * 3-slot weight ring + 3-slot token ring in LDS, LDS-DMA two stages ahead, one s_barrier per stage; waits counted by hand.
//
// Per 128-token tile and wave: 2304 MFMAs (73.7k matrix-pipe cycles); traffic per tile 1.5 MB of weights (L2) + 2 x 256 KB of tokens.
#include <stdlib.h>
#include <string.h>
#include <cmath>

#include "og_common.h"

namespace {

typedef unsigned og_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void og_lds_void;
typedef __attribute__((address_space(1))) const void og_glb_void;

constexpr int MT = 128;                       // tokens per workgroup
constexpr int WSTAGE = 32768;                 // one weight stage: 32 fragments of 1 KiB
constexpr int XSTAGE = MT * 128;              // one 32-channel k-group of the token tile: 128 rows x (64 B hi | 64 B lo)
constexpr int XOFF = 3 * WSTAGE;
constexpr int BOFF = XOFF + 3 * XSTAGE;       // biases * 256 (fp32): b0' [2D] then b3' [D]
constexpr int EPI_SLAB = 32 * 144;            // epilogue scratch: one 32-token slice, rows of (128 B + 16 B pad)

// Compile-time ablations (scripts/build_mlp_ablation.sh; results wrong by construction): 1 = no global stores, 4 = no MFMA,
// 8 = no LDS-DMA after the prologue, 32 = no token LDS-DMA after the prologue.
#ifndef OG_MLP_ABL
#define OG_MLP_ABL 0
#endif
// Experiment builds only (-DOG_MLP_TRACE=1): shader-cycle stamps of every wave at every stage hand-over (before the DMA wait, after
// it, after the barrier), kept in the lanes of three VGPRs (stamp of stage s in lane s) and written out at the end
// (og_debug_mlp_trace, scripts/trace_mlp.py).
#ifndef OG_MLP_TRACE
#define OG_MLP_TRACE 0
#endif
#if OG_MLP_TRACE
constexpr int OG_MT_BLOCKS = 512;
__device__ unsigned og_mlp_trace_buf[OG_MT_BLOCKS][4][4][64];
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
__global__ __launch_bounds__(256) void mlp_fused_kernel(MlpFusedArgs g) {
    static_assert(D == 256, "instantiated for 256-d descriptors (8 output blocks, 2 x 8 hidden blocks, 16 k-groups)");
    constexpr int G0 = 2 * D / 32;            // k-groups of fc.0 (K = 2D)
    constexpr int NJ = 8;                     // hidden blocks per pass = fc.3 stages per pass
    constexpr int NPASS = 2 * D / 256;
    constexpr int STAGES = NPASS * (G0 + NJ);
    constexpr int XSTAGES = NPASS * G0;
    constexpr int SMEM = BOFF + (2 * D + D) * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    static_assert(SMEM <= 163840 && 4 * 2 * EPI_SLAB <= XOFF, "LDS budget");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    auto launder = [](unsigned o) { asm volatile("" : "+v"(o)); return o; };

    // ---- LDS-DMA pieces (1 KiB each).  Tokens: wave w fills ITS OWN rows [32w, 32w+32) (4 pieces of 8 rows x 128 B) and is the only
    //      reader of them; weights: wave w fills fragments [8w, 8w+8) of a stage, every wave reads all 32. ----
    unsigned xoff[4];
    const char* const baseX = reinterpret_cast<const char*>(g.XO + (int64_t)t0 * g.ld);
    {
        const int rl = lane >> 3, pc = lane & 7;
        const int last = g.M - 1 - t0;                   // rows past the matrix are clamped (computed, never stored)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int rt = wave * 32 + h * 8 + rl;
            xoff[h] = (unsigned)((rt < last ? rt : last) * (int)g.ld * 2) + (unsigned)(pc ^ ((rt >> 1) & 7)) * 16u;
            asm volatile("" : "+v"(xoff[h]));
        }
    }
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const char* const baseW = g.wstream + wave * 8 * 1024;
    auto issue_w = [&](int s, int slot, int p) {         // piece p (0..7) of weight stage s into ring slot `slot`
        __builtin_amdgcn_global_load_lds((og_glb_void*)(scalar_ptr(baseW + (int64_t)s * WSTAGE + p * 1024) + launder(lane16)),
                                         (og_lds_void*)(smem + slot * WSTAGE + (wave * 8 + p) * 1024), 16, 0, 0);
    };
    auto issue_x = [&](int xs, int slot, int h) {        // piece h (0..3) of token stage xs (k-group xs % G0)
        __builtin_amdgcn_global_load_lds((og_glb_void*)(scalar_ptr(baseX + (int64_t)(xs % G0) * 128) + launder(xoff[h])),
                                         (og_lds_void*)(smem + XOFF + slot * XSTAGE + (wave * 32 + h * 8) * 128), 16, 0, 0);
    };

    // ---- prologue: W(0), X(0), W(1), X(1) in flight; biases * 256 into LDS in their shadow (the bias loads are issued FIRST: the
    //      counter is in order, so waiting for them does not wait for the DMA pieces behind them) ----
    static_assert(D == 256, "one b0 pair and one b3 value per thread");
    float bv0 = g.b0[tid], bv1 = g.b0[tid + 256], bv2 = g.b3[tid];
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_w(0, 0, p);
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_x(0, 0, h);
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_w(1, 1, p);
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_x(1, 1, h);
    {
        // inline asm: a compiler-visible LDS store would be ordered behind every LDS-DMA in flight (vmcnt(0))
        const float is = 1.f / g.scale;
        const unsigned ba = lds0 + BOFF + (unsigned)tid * 4u;
        asm volatile("s_waitcnt vmcnt(24)\n\t"
                     "v_mul_f32 %0, %0, %4\n\t"
                     "v_mul_f32 %1, %1, %4\n\t"
                     "v_mul_f32 %2, %2, %4\n\t"
                     "ds_write_b32 %3, %0\n\t"
                     "ds_write_b32 %3, %1 offset:1024\n\t"
                     "ds_write_b32 %3, %2 offset:2048"
                     : "+v"(bv0), "+v"(bv1), "+v"(bv2) : "v"(ba), "s"(is) : "memory");
    }

    f32x16 acc0[8], acc3[8];
    f16x8 wh[2][2], wl[2][2], xh[2], xl[2];
    auto lds_read = [&](f16x8& dst, unsigned addr, int imm) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); };
    // fragment (t, i, part) of the current weight stage: ((t * 8 + i) * 2 + part) KiB behind the stage base (part 0 = hi, 1 = lo)
    unsigned wa = 0;                                       // lds address of this lane's 16 bytes of fragment 0 of the stage being read
    // token fragments: row 32w + l31 of the tile, logical chunk 2t + hi (hi part), + 4 (lo part), XOR-swizzled like the DMA source
    const int swz = (l31 >> 1) & 7;
    unsigned xk[2][2], xa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
            xk[t][hl] = (unsigned)((wave * 32 + l31) * 128) + ((unsigned)(((2 * t + hi) ^ swz) * 16) ^ (hl ? 64u : 0u));
    auto set_x = [&](int slot) {
        const unsigned b = lds0 + XOFF + slot * XSTAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) { xa[t][0] = b + xk[t][0]; xa[t][1] = b + xk[t][1]; }
    };
    auto read_x = [&](int t) { lds_read(xh[t], xa[t][0], 0); lds_read(xl[t], xa[t][1], 0); };
    auto set_w = [&](int slot) { wa = lds0 + slot * WSTAGE + lane16; };
    // One weight fragment of group grp (t = grp >> 2, channel blocks 2 (grp & 3) + {0, 1}) in the order the group's MFMAs first use
    // them: k = 0: lo of block 0, 1: lo of block 1, 2: hi of block 0, 3: hi of block 1.
    auto read_wk = [&](int grp, int k, int buf) {
        const int f = ((grp >> 2) * 8 + 2 * (grp & 3) + (k & 1)) * 2 + (k < 2 ? 1 : 0);
        lds_read(k < 2 ? wl[buf][k & 1] : wh[buf][k & 1], wa, f * 1024);
    };
    // LDS returns in order: "all but the N newest reads have landed".  The "+v" operand ties the wait to the register it releases.
    auto wait1 = [&](int newer, f16x8& r) {
        if (newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
        else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r));
        else if (newer == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r));
        else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r));
    };
    auto tie2 = [&](f16x8& a, f16x8& b) { asm volatile("" : "+v"(a), "+v"(b)); };

    // accumulator initialisation: bias * 256 of 8 consecutive channel blocks from LDS (float offset `off`).  Register r of a 32x32
    // accumulator holds channel (r & 3) + 8 (r >> 2) + 4 hi of the block.  Inline-asm reads with their own full wait: LDS reads the
    // compiler knows about would be ordered behind the LDS-DMA in flight (vmcnt(0)) and could slip between the counted fragment reads.
    auto init_acc8 = [&](f32x16 (&a)[8], int off) {
        const unsigned ad = lds0 + BOFF + (unsigned)(off + 4 * hi) * 4u;
#pragma unroll
        for (int i2 = 0; i2 < 8; i2 += 2) {
            f32x4 b[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[i][q]) : "v"(ad), "i"(((i2 + i) * 32 + 8 * q) * 4));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i2 + i][4 * q + e] = b[i][q][e];
        }
    };

    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");      // my pieces of W(0), X(0) landed (W(1), X(1) may still fly), my bias stores too
    __builtin_amdgcn_s_barrier();                                     // ... everybody's
    init_acc8(acc3, 2 * D);

    int s = 0, wslot = 0;           // weight stage being consumed and its ring slot
    int xs = 0, xslot = 0;          // token stage being consumed (fc.0 stages only) and its ring slot
    auto next3 = [](int v) { return v == 2 ? 0 : v + 1; };
    auto prev3 = [](int v) { return v == 0 ? 2 : v - 1; };
    const float sc = g.scale;

    // ---- one stage = 8 groups of 6 MFMAs on ONE wave per SIMD.  With a single wave nothing else hides what sits between two MFMAs:
    //      the first generation issued a group's four fragment reads, its wait and three LDS-DMA pieces in front of / inside the
    //      group and ran at 2520 cycles per stage against 1536 of matrix-pipe time (profiles/r03_a_mlp_fused_trace.log: a wave issues
    //      an MFMA only when the pipe is free, so everything between two MFMAs beyond the 32 cycles the previous one executes is
    //      exposed).  Here every MFMA is followed by ONE small item that fits its shadow:
    //        slots 0-3  the next group's fragment reads, one each, in first-use order (lo 0, lo 1, hi 0, hi 1), waited for one by one
    //                   (lgkmcnt(3): a read has six MFMA slots = 192 cycles to land)
    //        slots 0-3  of groups 0-5 also the LDS-DMA pieces, wave w in slots w and (w + 2) & 3 -- the four waves of a CU run in
    //                   lock-step and the CU accepts one 1 KiB piece per >= 16 cycles: spread over different slots they do not queue
    //                   behind each other
    //        slots 4-5  fc.0: the token fragments of the second k-step; fc.3: the (hi, lo) conversion of the next hidden block, 12 small steps
    //      The hand-over to the next stage (DMA wait + barrier) sits between MFMA 1 and 2 of the last group; the next stage's first
    //      fragments are read in that group's remaining slots.
    auto hand_over = [&](bool next_exists, bool next_is_fc0, int next_xslot, int issued) {
        OG_MT(0, s);
        // everything issued before this stage has landed (only this stage's own pieces may still fly)
        if (issued == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (issued == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OG_MT(1, s);
        if (next_exists) {
            __builtin_amdgcn_s_barrier();
            OG_MT(2, s);
            set_w(next3(wslot));
            if (next_is_fc0) set_x(next_xslot);
        }
    };
#define OG_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, acc_, 0, 0, 0)
#define OG_SB() __builtin_amdgcn_sched_barrier(0)
    // the six MFMAs of group grp of a stage with their slots; ACC = accumulator array, BH / BL = the B fragments of k-step t
#if OG_MLP_ABL & 4
#define OG_MM(acc_, a_, b_) asm volatile("" ::"v"(a_), "v"(b_))
#else
#define OG_MM(acc_, a_, b_) OG_MFMA(acc_, a_, b_)
#endif
#define OG_GROUP(ACC, BH, BL, GRP, CNT, SLOT, HANDOVER, NEXT_EXISTS, NEXT_FC0)                                                  \
    {                                                                                                                          \
        constexpr int grp_ = GRP, ip_ = grp_ & 3, b_ = grp_ & 1, nb_ = b_ ^ 1;                                                  \
        if constexpr (grp_ < 7) {                                                                                              \
            wait1(CNT, wl[b_][0]); OG_SB(); OG_MM(ACC[2 * ip_], wl[b_][0], BH); OG_SB(); SLOT(grp_, 0); read_wk(grp_ + 1, 0, nb_); OG_SB();      \
            wait1(CNT, wl[b_][1]); OG_SB(); OG_MM(ACC[2 * ip_ + 1], wl[b_][1], BH); OG_SB(); SLOT(grp_, 1); read_wk(grp_ + 1, 1, nb_); OG_SB();  \
            wait1(CNT, wh[b_][0]); OG_SB(); OG_MM(ACC[2 * ip_], wh[b_][0], BL); OG_SB(); SLOT(grp_, 2); read_wk(grp_ + 1, 2, nb_); OG_SB();      \
            wait1(CNT, wh[b_][1]); OG_SB(); OG_MM(ACC[2 * ip_ + 1], wh[b_][1], BL); OG_SB(); SLOT(grp_, 3); read_wk(grp_ + 1, 3, nb_); OG_SB();  \
            OG_MM(ACC[2 * ip_], wh[b_][0], BH); OG_SB(); SLOT(grp_, 4); OG_SB();                                                \
            OG_MM(ACC[2 * ip_ + 1], wh[b_][1], BH); OG_SB(); SLOT(grp_, 5); OG_SB();                                            \
        } else {                                                                                                               \
            wait1(3, wl[b_][0]); OG_SB(); OG_MM(ACC[2 * ip_], wl[b_][0], BH); OG_SB(); SLOT(grp_, 0); OG_SB();                   \
            wait1(2, wl[b_][1]); OG_SB(); OG_MM(ACC[2 * ip_ + 1], wl[b_][1], BH); OG_SB(); SLOT(grp_, 1); OG_SB();               \
            wait1(0, wh[b_][0]); tie2(wh[b_][0], wh[b_][1]); OG_SB();      /* all my LDS reads of this stage are done */         \
            HANDOVER; OG_SB();                                                                                                 \
            OG_MM(ACC[2 * ip_], wh[b_][0], BL); OG_SB(); if (NEXT_FC0) read_x(0); OG_SB();                                       \
            OG_MM(ACC[2 * ip_ + 1], wh[b_][1], BL); OG_SB(); if (NEXT_EXISTS) read_wk(0, 0, 0); OG_SB();                         \
            OG_MM(ACC[2 * ip_], wh[b_][0], BH); OG_SB(); if (NEXT_EXISTS) read_wk(0, 1, 0); OG_SB();                             \
            OG_MM(ACC[2 * ip_ + 1], wh[b_][1], BH); OG_SB(); if (NEXT_EXISTS) { read_wk(0, 2, 0); read_wk(0, 3, 0); } OG_SB();   \
        }                                                                                                                      \
    }

    set_w(0); set_x(0);
    read_x(0);
    read_wk(0, 0, 0); read_wk(0, 1, 0); read_wk(0, 2, 0); read_wk(0, 3, 0);

#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        init_acc8(acc0, pass * 256);

        // ================= fc.0: acc0[i] += W0'[pass half, block i][k-group] · [x ; O][k-group], 16 stages =================
#pragma unroll 1
        for (int kg = 0; kg < G0; ++kg) {
            const bool ix = xs + 2 < XSTAGES;                           // X(xs+2) exists (W(s+2) always does during fc.0)
            const int wslot2 = prev3(wslot), xslot2 = prev3(xslot);     // slots of W(s+2), X(xs+2): (s + 2) % 3 = (s - 1) % 3
            const bool next_x = kg + 1 < G0;                            // the next stage is an fc.0 stage (reads token fragments)
            // piece p of this wave and stage: 0-7 weights, 8-11 tokens
            auto piece = [&](int p) {
                if (OG_MLP_ABL & 8) return;
                if (p < 8) issue_w(s + 2, wslot2, p);
                else if (ix && !(OG_MLP_ABL & 32)) issue_x(xs + 2, xslot2, p - 8);
            };
            auto slot = [&](int grp, int k) {
                if (k < 4 && grp < 6) {
                    if (wave == k) piece(2 * grp);
                    else if (wave == ((k + 2) & 3)) piece(2 * grp + 1);
                }
                if (grp == 1 && k == 4) read_x(1);                      // the second k-step's token fragments (first used by group 4)
            };
            tie2(xh[0], xl[0]);
            OG_GROUP(acc0, xh[0], xl[0], 0, 3, slot, , true, false)
            OG_GROUP(acc0, xh[0], xl[0], 1, 3, slot, , true, false)
            OG_GROUP(acc0, xh[0], xl[0], 2, 5, slot, , true, false)
            OG_GROUP(acc0, xh[0], xl[0], 3, 3, slot, , true, false)
            tie2(xh[1], xl[1]);
            OG_GROUP(acc0, xh[1], xl[1], 4, 3, slot, , true, false)
            OG_GROUP(acc0, xh[1], xl[1], 5, 3, slot, , true, false)
            OG_GROUP(acc0, xh[1], xl[1], 6, 3, slot, , true, false)
            OG_GROUP(acc0, xh[1], xl[1], 7, 3, slot, hand_over(true, next_x, next3(xslot), ix ? 12 : 8), true, next_x)
            ++s; wslot = next3(wslot);
            ++xs; xslot = next3(xslot);
        }

        // ================= fc.3: acc3[i] += W3'[block i][hidden block j of this half] · relu(acc0[j] / 256), 8 stages =================
        // hidden block j as B fragments: element e of k-step t is accumulator register 8t + e (og_pack_mlp_stream permutes W3' to match)
        unsigned hh[2][2][4], hl[2][2][4];             // [buffer][t][dword]
        float cv[4];
        // (hi, lo) conversion of quarter q (accumulator registers 4q .. 4q+3) of a hidden block in three small steps
        auto convert_step = [&](const f32x16& a, int step, int buf) {
#pragma clang fp contract(off)
            const int q = step / 3, k = step % 3;
            if (k == 0) { cv[0] = a[4 * q] * sc; cv[1] = a[4 * q + 1] * sc; cv[2] = a[4 * q + 2] * sc; cv[3] = a[4 * q + 3] * sc; }
            else if (k == 1) { cv[0] = fmaxf(cv[0], 0.f); cv[1] = fmaxf(cv[1], 0.f); cv[2] = fmaxf(cv[2], 0.f); cv[3] = fmaxf(cv[3], 0.f); }
            else {
                const int t = q >> 1, d = 2 * (q & 1);
                og_split4(cv[0], cv[1], cv[2], cv[3], hh[buf][t][d], hl[buf][t][d], hh[buf][t][d + 1], hl[buf][t][d + 1]);
            }
        };
#pragma unroll
        for (int st = 0; st < 12; ++st) convert_step(acc0[0], st, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool iw = s + 2 < STAGES;
            const int wslot2 = prev3(wslot);
            const bool next_exists = s + 1 < STAGES;
            const bool next_x = j + 1 == NJ && pass + 1 < NPASS;        // the next stage is the first fc.0 stage of the next pass
            const int hb = j & 1;
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = __builtin_bit_cast(f16x8, og_u32x4{hh[hb][t][0], hh[hb][t][1], hh[hb][t][2], hh[hb][t][3]});
                bl[t] = __builtin_bit_cast(f16x8, og_u32x4{hl[hb][t][0], hl[hb][t][1], hl[hb][t][2], hl[hb][t][3]});
            }
            auto slot = [&](int grp, int k) {
                if (k < 4 && grp < 4 && iw && !(OG_MLP_ABL & 8)) {
                    if (wave == k) issue_w(s + 2, wslot2, 2 * grp);
                    else if (wave == ((k + 2) & 3)) issue_w(s + 2, wslot2, 2 * grp + 1);
                }
                if (k >= 4 && grp < 6 && j + 1 < NJ) convert_step(acc0[j + 1 < NJ ? j + 1 : j], 2 * grp + (k - 4), hb ^ 1);
            };
            OG_GROUP(acc3, bh[0], bl[0], 0, 3, slot, , true, false)
            OG_GROUP(acc3, bh[0], bl[0], 1, 3, slot, , true, false)
            OG_GROUP(acc3, bh[0], bl[0], 2, 3, slot, , true, false)
            OG_GROUP(acc3, bh[0], bl[0], 3, 3, slot, , true, false)
            OG_GROUP(acc3, bh[1], bl[1], 4, 3, slot, , true, false)
            OG_GROUP(acc3, bh[1], bl[1], 5, 3, slot, , true, false)
            OG_GROUP(acc3, bh[1], bl[1], 6, 3, slot, , true, false)
            OG_GROUP(acc3, bh[1], bl[1], 7, 3, slot, hand_over(next_exists, next_x, xslot, iw ? 8 : 0), next_exists, next_x)
            ++s; wslot = next3(wslot);
        }
    }
#undef OG_GROUP
#undef OG_MM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OG_MT(3, 1);
    __builtin_amdgcn_s_barrier();          // every wave is past its last fragment reads, no DMA in flight: the rings are free

    // ================= epilogue: x <- acc3 / 256 + (x_hi + x_lo), written back as hl32 rows =================
    // A lane owns ONE token and 4 consecutive channels per register group.  Residual rows come in and result rows go out as whole
    // 128-byte lines (one channel block of one token: 64 B hi | 64 B lo), 8 rows per instruction, and change layout through two
    // per-wave LDS slabs (as gemm_f16x3_epilogue_fast).  Channel blocks are handled in pairs; the next pair's residual is in flight
    // while the current one is finished.
    {
#pragma clang fp contract(off)
        constexpr int ROWB = 128 + 16;
        char* const slab2 = smem + wave * 2 * EPI_SLAB;
        const int tok0 = t0 + wave * 32;
        char* const rows = reinterpret_cast<char*>(g.XO);
        const unsigned rd_off = (unsigned)((lane >> 3) * ROWB + (lane & 7) * 16);
        const bool full = t0 + MT <= g.M;                    // block-uniform
        int64_t rowb[4];                                      // byte offset of this lane's row per store / load instruction
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int r = tok0 + it * 8 + (lane >> 3);
            if (r > g.M - 1) r = g.M - 1;
            rowb[it] = (int64_t)r * g.ld * 2 + (lane & 7) * 16;
        }
        og_u32x4 rrow[2][4];
        auto load_res = [&](int ip) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int it = 0; it < 4; ++it) rrow[i][it] = *reinterpret_cast<const og_u32x4*>(rows + rowb[it] + (2 * ip + i) * 128);
        };
        load_res(0);
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            f32x16 a[2];
            a[0] = acc3[2 * ip]; a[1] = acc3[2 * ip + 1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int it = 0; it < 4; ++it) *reinterpret_cast<og_u32x4*>(slab2 + i * EPI_SLAB + it * 8 * ROWB + rd_off) = rrow[i][it];
            if (ip + 1 < 4) load_res(ip + 1);
            og_u32x4 raw[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const char* d = slab2 + i * EPI_SLAB + l31 * ROWB + (8 * q + 4 * hi) * 2;
                    const uint2 h2 = *reinterpret_cast<const uint2*>(d), l2 = *reinterpret_cast<const uint2*>(d + 64);
                    raw[i][q] = og_u32x4{h2.x, h2.y, l2.x, l2.y};
                }
            // v = acc * scale + hi + lo: two mixed-precision FMAs per element (gemm_f16x3_epilogue_finish_spec<OG_EM_RES_HL>)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x0 = a[i][4 * q], x1 = a[i][4 * q + 1], x2 = a[i][4 * q + 2], x3 = a[i][4 * q + 3];
                    asm("s_nop 1\n\t"
                        "v_fma_mix_f32 %0, %0, %4, %5 op_sel_hi:[0,0,1]\n\t"
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
            // registers -> slabs: slab i = [32 tok][hi 64 B | lo 64 B] of channel block 2 ip + i
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned ha, la, hb2, lb;
                    og_split4(a[i][4 * q], a[i][4 * q + 1], a[i][4 * q + 2], a[i][4 * q + 3], ha, la, hb2, lb);
                    char* d = slab2 + i * EPI_SLAB + l31 * ROWB + (8 * q + 4 * hi) * 2;
                    *reinterpret_cast<uint2*>(d) = make_uint2(ha, hb2);
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

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------
bool og_mlp_fused_supported(int D) { return D == 256; }

size_t og_mlp_stream_bytes(int D) { return og_mlp_fused_supported(D) ? (size_t)6 * D * D * 4 : 0; }      // 4D^2 (W0') + 2D^2 (W3') (hi, lo) pairs

bool og_mlp_fused_enabled(int D) {
    static const bool on = [] { const char* e = getenv("OG_MLP_FUSED"); return !e || atoi(e) != 0; }();   // experiments: 0 = fc.0 and fc.3 as two GEMM launches
    return on && og_mlp_fused_supported(D);
}

// Fragment-major weight stream of mlp_fused_kernel.  W0 [2D][2D], W3 [D][2D] row-major double (already folded), written as (hi, lo)
// halves of 256 w.  Stage order: pass a: fc.0 k-groups 0..15, then fc.3 hidden blocks 8a..8a+7.  Inside a stage fragment
// f = (t * 8 + i) * 2 + part (part 0 = hi, 1 = lo) holds for lane l = (rho = l & 31, h = l >> 5) eight halves e = 0..7:
//   fc.0 stage (a, kg):  W0[32 (8a + i) + rho][32 kg + 16 t + 8 h + e]
//   fc.3 stage (a, j):   W3[32 i + rho][32 (8a + j) + 16 t + 8 (e >> 2) + 4 h + (e & 3)]     (the accumulator-register order, above)
// Returns false when a scaled weight does not fit binary16.
bool og_pack_mlp_stream(int D, const double* W0, const double* W3, void* out) {
    if (!og_mlp_fused_supported(D)) return false;
    const int D2 = 2 * D, G0 = D2 / 32, NJ = 8, NPASS = D2 / 256;
    _Float16* o = (_Float16*)out;
    bool ok = true;
    auto put = [&](int64_t stage, int f, int l, int e, double w) {
        w *= OG_W_SCALE;
        if (!(fabs(w) <= 65504.0)) { ok = false; w = 0.0; }
        const _Float16 hi = (_Float16)w;
        _Float16* base = o + stage * (32768 / 2) + (int64_t)f * 512 + l * 8 + e;     // fragment f: 1 KiB = 512 halves
        base[0] = hi;
        base[512] = (_Float16)(w - (double)hi);                                        // the lo fragment follows the hi fragment
    };
    int64_t stage = 0;
    for (int a = 0; a < NPASS; ++a) {
        for (int kg = 0; kg < G0; ++kg, ++stage)
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < 8; ++i)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e)
                            put(stage, (t * 8 + i) * 2, l, e, W0[(int64_t)(32 * (8 * a + i) + (l & 31)) * D2 + 32 * kg + 16 * t + 8 * (l >> 5) + e]);
        for (int j = 0; j < NJ; ++j, ++stage)
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < D / 32; ++i)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e)
                            put(stage, (t * 8 + i) * 2, l, e,
                                W3[(int64_t)(32 * i + (l & 31)) * D2 + 32 * (8 * a + j) + 16 * t + 8 * (e >> 2) + 4 * (l >> 5) + (e & 3)]);
    }
    return ok;
}

int og_launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t stream) {
    if (!a.XO || !a.wstream || !a.b0 || !a.b3 || a.M <= 0) return OG_E_INVALID;
    if (!og_mlp_fused_supported(D)) return OG_E_SHAPE;
    if (((uintptr_t)a.XO & 15) || ((uintptr_t)a.wstream & 15) || (a.ld & 7) || a.ld < 4 * (int64_t)D) return OG_E_ALIGN;
    if ((int64_t)a.M * a.ld * 2 >= (int64_t)1 << 32) return OG_E_SHAPE;             // 32-bit lane offsets
    if (!(a.scale != 0.f) || !std::isfinite(a.scale)) return OG_E_INVALID;
    const int tiles = (a.M + MT - 1) / MT;
    hipLaunchKernelGGL(mlp_fused_kernel<256>, dim3(tiles), dim3(256), 0, stream, a);
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

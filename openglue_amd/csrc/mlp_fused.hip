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

#ifndef OG_MLP_ABL
#define OG_MLP_ABL 0                          // experiments: 1 = no stores, 4 = no MFMA (results wrong by construction)
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
    // fragment (t, i, part) of the current weight stage: ((t * 8 + i) * 2 + part) KiB behind the stage base
    unsigned wa = 0;                                       // lds address of this lane's 16 bytes of fragment 0 of the stage being read
    auto read_w = [&](int t, int ip, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_read(wh[buf][i], wa, ((t * 8 + 2 * ip + i) * 2) * 1024);
            lds_read(wl[buf][i], wa, ((t * 8 + 2 * ip + i) * 2 + 1) * 1024);
        }
    };
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
    auto wait_w = [&](int b, int newer) {                 // frees fragment buffer b while `newer` younger LDS reads may stay in flight
        if (newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[b][0]), "+v"(wl[b][0]), "+v"(wh[b][1]), "+v"(wl[b][1]));
        else if (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wh[b][0]), "+v"(wl[b][0]), "+v"(wh[b][1]), "+v"(wl[b][1]));
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wh[b][0]), "+v"(wl[b][0]), "+v"(wh[b][1]), "+v"(wl[b][1]));
    };
    auto tie_x = [&](int t) { asm volatile("" : "+v"(xh[t]), "+v"(xl[t])); };

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
    set_w(0); set_x(0);
    read_x(0);
    read_w(0, 0, 0);

    const float sc = g.scale;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        init_acc8(acc0, pass * 256);

        // ================= fc.0: acc0[i] += W0'[pass half, block i][k-group] · [x ; O][k-group], 16 stages =================
#pragma unroll 1
        for (int kg = 0; kg < G0; ++kg) {
            const bool iw = s + 2 < STAGES, ix = xs + 2 < XSTAGES;      // W(s+2) / X(xs+2) exist
            const int wslot2 = prev3(wslot), xslot2 = prev3(xslot);     // their slots: (s + 2) % 3 = (s - 1) % 3
            const bool next_x = kg + 1 < G0;                            // the next stage is an fc.0 stage (reads token fragments)
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                const int t = grp >> 2, ip = grp & 3, b = grp & 1;
                if (grp < 7) {
                    if (grp == 0) read_x(1);
                    read_w((grp + 1) >> 2, (grp + 1) & 3, (grp + 1) & 1);
                    wait_w(b, grp == 0 ? 6 : 4);
                    if (grp == 0) tie_x(0);
                    if (grp == 1) tie_x(1);
                } else {
                    wait_w(b, 0);                                       // all my LDS reads of this stage are done
                    // hand-over: everything issued before this stage has landed (only this stage's own pieces may still fly)
                    if (iw && ix) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (iw) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    set_w(next3(wslot));
                    if (next_x) { set_x(next3(xslot)); read_x(0); }
                    read_w(0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#if !(OG_MLP_ABL & 4)
                acc0[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][0], xh[t], acc0[2 * ip], 0, 0, 0);
                acc0[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][1], xh[t], acc0[2 * ip + 1], 0, 0, 0);
#endif
                if (grp < 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (iw) { issue_w(s + 2, wslot2, 2 * grp); issue_w(s + 2, wslot2, 2 * grp + 1); }
                    if (ix) issue_x(xs + 2, xslot2, grp);
                    __builtin_amdgcn_sched_barrier(0);
                }
#if !(OG_MLP_ABL & 4)
                acc0[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][0], xl[t], acc0[2 * ip], 0, 0, 0);
                acc0[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][1], xl[t], acc0[2 * ip + 1], 0, 0, 0);
                acc0[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][0], xh[t], acc0[2 * ip], 0, 0, 0);
                acc0[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][1], xh[t], acc0[2 * ip + 1], 0, 0, 0);
#else
                asm volatile("" ::"v"(wh[b][0]), "v"(wl[b][0]), "v"(wh[b][1]), "v"(wl[b][1]), "v"(xh[t]), "v"(xl[t]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            ++s; wslot = next3(wslot);
            ++xs; xslot = next3(xslot);
        }

        // ================= fc.3: acc3[i] += W3'[block i][hidden block j of this half] · relu(acc0[j] / 256), 8 stages =================
        // hidden block j as B fragments: element e of k-step t is accumulator register 8t + e (og_pack_mlp_stream permutes W3' to match)
        unsigned hh[2][2][4], hl[2][2][4];             // [buffer][t][dword]
        auto convert_quarter = [&](const f32x16& a, int q, int buf) {
#pragma clang fp contract(off)
            float v0 = fmaxf(a[4 * q] * sc, 0.f), v1 = fmaxf(a[4 * q + 1] * sc, 0.f), v2 = fmaxf(a[4 * q + 2] * sc, 0.f), v3 = fmaxf(a[4 * q + 3] * sc, 0.f);
            const int t = q >> 1, d = 2 * (q & 1);
            og_split4(v0, v1, v2, v3, hh[buf][t][d], hl[buf][t][d], hh[buf][t][d + 1], hl[buf][t][d + 1]);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) convert_quarter(acc0[0], q, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool iw = s + 2 < STAGES;
            const int wslot2 = prev3(wslot);
            const bool next_x = j + 1 == NJ && pass + 1 < NPASS;        // the next stage is the first fc.0 stage of the next pass
            const int hb = j & 1;
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = __builtin_bit_cast(f16x8, og_u32x4{hh[hb][t][0], hh[hb][t][1], hh[hb][t][2], hh[hb][t][3]});
                bl[t] = __builtin_bit_cast(f16x8, og_u32x4{hl[hb][t][0], hl[hb][t][1], hl[hb][t][2], hl[hb][t][3]});
            }
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                const int t = grp >> 2, ip = grp & 3, b = grp & 1;
                if (grp < 7) {
                    read_w((grp + 1) >> 2, (grp + 1) & 3, (grp + 1) & 1);
                    wait_w(b, 4);
                } else {
                    wait_w(b, 0);
                    if (iw) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (s + 1 < STAGES) {
                        __builtin_amdgcn_s_barrier();
                        set_w(next3(wslot));
                        if (next_x) { set_x(xslot); read_x(0); }
                        read_w(0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#if !(OG_MLP_ABL & 4)
                acc3[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][0], bh[t], acc3[2 * ip], 0, 0, 0);
                acc3[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][1], bh[t], acc3[2 * ip + 1], 0, 0, 0);
#endif
                if (grp < 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (iw) { issue_w(s + 2, wslot2, 2 * grp); issue_w(s + 2, wslot2, 2 * grp + 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j + 1 < NJ && (grp & 1) == 0) {        // the next hidden block, a quarter at a time, between the MFMAs
                    convert_quarter(acc0[j + 1 < NJ ? j + 1 : j], grp >> 1, hb ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#if !(OG_MLP_ABL & 4)
                acc3[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][0], bl[t], acc3[2 * ip], 0, 0, 0);
                acc3[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][1], bl[t], acc3[2 * ip + 1], 0, 0, 0);
                acc3[2 * ip] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][0], bh[t], acc3[2 * ip], 0, 0, 0);
                acc3[2 * ip + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][1], bh[t], acc3[2 * ip + 1], 0, 0, 0);
#else
                asm volatile("" ::"v"(wh[b][0]), "v"(wl[b][0]), "v"(wh[b][1]), "v"(wl[b][1]), "v"(bh[t]), "v"(bl[t]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            ++s; wslot = next3(wslot);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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

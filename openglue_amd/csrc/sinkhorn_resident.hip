// Log-domain Sinkhorn with the score matrix RESIDENT ON CHIP for all iterations (gfx950).
//
// Replaces the iteration loop of log_otp_solver (reference optimal_transport.py:24-26) for batches whose score
// matrices fit the register files + LDS of the chip: 256 CUs x (512 KB of VGPRs + 160 KB of LDS) = 172 MB, against
// 134 MB of fp32 scores at BASELINE config 2 (32 pairs x 1024 x 1024).  The streaming kernels (sinkhorn.hip) read S
// from HBM once per iteration -- 134 MB x 100 iterations at ~5.6 TB/s = 2.4 ms, plus 200 launch boundaries; here S is
// read ONCE, every workgroup keeps 128 rows of one pair (8 of the 16 rows of each wave in registers, 4 in LDS, four
// re-read from memory per iteration under the cross-workgroup exchange: 75 % resident) and runs all dual-stabilised iterations in ONE launch.
//
// Per iteration and pair (same recursion and the same arithmetic as sinkhorn_sweep_fast / sinkhorn_combine_fast):
//     P_ij = 2^(s_ij + v_j + u_i)            s = S/reg * log2 e, duals in base 2
//     rowsum_i = sum_j P_ij + 2^(z + v_N + u_i)     u_i += log2 a_i - log2 rowsum_i       f_i = 2^(u_i' - u_i)
//     colsum_j = sum_i P_ij f_i + 2^(z + v_j + u_M')  v_j += log2 b_j - log2 colsum_j
// Rows are wave-local (one wave owns a row: lane-local exponentials + one DPP reduction), columns cross the G
// workgroups of the pair: every workgroup publishes its 1024 column partials as 8-byte {epoch, value} granules
// (cdna_hip_programming.md Guideline 16, form R2: the data is the flag -- agent-scope relaxed atomics on both sides, no
// fence), sweeps the 8 x 1025 granules of its pair (four 16-byte loads per column, see rs_granule_offset) and computes ALL
// new v_j redundantly but bit-identically (fixed summation order), so no second exchange is needed.  Slots without a workgroup (G < 8) and columns >= n carry published zeros: the
// sweep has no predicates.  Granules are double-buffered by epoch parity: a workgroup can
// only reach epoch t+2 after it has seen every epoch-t+1 granule, i.e. after every peer has finished reading epoch t.
// Every spin is bounded (status word: 0 ok, 1 = a wait timed out -> results invalid); all G x B workgroups must be
// co-resident: one 512-thread workgroup per CU (148 KB of LDS), G x B <= number of CUs, checked by the launcher.
#include <stdlib.h>
#include <type_traits>

#include "og_common.h"

namespace {

constexpr int RS_RW = 16;              // row slots per wave
constexpr int RS_RR = 8;               //   slots 0..7: rows held in registers (the 128 accumulator registers of the wave)
constexpr int RS_LR = 4;               //   slots 8..11: rows held in LDS
constexpr int RS_SR = 4;               //   slots 12..15: re-read from memory every iteration (prefetched under the exchange)
static_assert(RS_RR + RS_LR + RS_SR == RS_RW, "sixteen row slots per wave");
constexpr int RS_NW = 8;               // waves per workgroup
constexpr int RS_ROWS = RS_NW * RS_RW; // 128 rows per workgroup
constexpr int RS_NCOL = 1024;          // columns per row held on chip (16 per lane)
constexpr int RS_NG = RS_NCOL + 16;    // columns with granules per pair: 1024 column partials + the dustbin-column term at index 1024
constexpr int RS_GMAX = 8;             // slots per column = workgroups per pair at most (m <= 1024 rows on chip)
constexpr unsigned RS_SPIN_LIMIT = 1u << 21;
constexpr float RS_LOG2E = 1.4426950408889634f;
constexpr float RS_LN2 = 0.6931471805599453f;

typedef unsigned long long rs_u64;
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) rs_u64 rs_gu64;
typedef __attribute__((address_space(1))) unsigned rs_gu32;
typedef unsigned rs_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) char rs_gchar;
typedef __attribute__((address_space(1))) f32x4 rs_gf32x4;

// ---- exchange area of one (parity, pair): 8-byte {epoch, value} granules, RS_GMAX slots (one per workgroup of the pair) for each
// of the 1024 columns + the dustbin column.  Layout: blocks of 64 columns; inside a block four 1-KB rows, row k = slots 2k and
// 2k+1 of the 64 columns (16 bytes per column).  A wave reads the eight slots of its 64 columns with four 16-byte loads per lane
// whose lanes are contiguous (1 KB = 8 cache lines per instruction, one address + immediates 0/1024/2048/3072); laid out
// [column][slot] the same loads touch 32 lines each and the sweep gets slower, laid out [slot][column] they are sixteen 8-byte
// loads with their own addresses. ----
constexpr int RS_XBLOCK = 4096;                              // bytes per block of 64 columns
constexpr int RS_XPAIR = (RS_NCOL / 64 + 1) * RS_XBLOCK;     // bytes per (parity, pair): 16 column blocks + the dustbin column's block
__device__ __forceinline__ unsigned rs_granule_offset(int col, int slot) {      // byte offset inside the exchange area of a (parity, pair)
    return (unsigned)((col >> 6) * RS_XBLOCK + (slot >> 1) * 1024 + (col & 63) * 16 + (slot & 1) * 8);
}

// The eight granules of two columns + one granule of the dustbin column, read from the L2 -- never from this CU's vector L1 (an
// aligned 8-byte granule cannot tear).  The wait is part of the block: the compiler does not count the loads of an asm statement.
//   agent scope (sc1): coherent over the whole device -- on this multi-XCD part every such load goes to the fabric behind the
//   per-XCD L2s (which are not coherent with each other): ~3k cycles per round trip, 11-12k cycles per sweep;
//   XCD-local (LOCAL): the workgroups of a pair share one XCD = one L2 (verified at run time, kernel prologue).  Workgroup-scope
//   streaming loads (sc0 nt) do not keep their line in the vector L1, so a re-poll reads the L2 again: 5-6k cycles per sweep.
//   (sc0 alone may hit a stale L1 line for ever; buffer_inv sc0 does not help in non-tgsplit mode; buffer_inv sc1 is correct but
//   costs 8k cycles per iteration -- it is the guaranteed-progress fallback of the poll loop, every 64th poll.)
template <bool LOCAL>
__device__ __forceinline__ void rs_load_columns(const rs_gu64* base0, const rs_gu64* base1, unsigned off, unsigned offd, rs_u32x4 (&a)[4],
                                                rs_u32x4 (&b)[4], rs_u64& d) {
#define RS_LOADS(M)                                                                                                                       \
    asm volatile("global_load_dwordx4 %0, %9, %11 " M "\n\t"                                                                              \
                 "global_load_dwordx4 %1, %9, %11 offset:1024 " M "\n\t"                                                                  \
                 "global_load_dwordx4 %2, %9, %11 offset:2048 " M "\n\t"                                                                  \
                 "global_load_dwordx4 %3, %9, %11 offset:3072 " M "\n\t"                                                                  \
                 "global_load_dwordx4 %4, %9, %12 " M "\n\t"                                                                              \
                 "global_load_dwordx4 %5, %9, %12 offset:1024 " M "\n\t"                                                                  \
                 "global_load_dwordx4 %6, %9, %12 offset:2048 " M "\n\t"                                                                  \
                 "global_load_dwordx4 %7, %9, %12 offset:3072 " M "\n\t"                                                                  \
                 "global_load_dwordx2 %8, %10, %11 " M "\n\t"                                                                             \
                 "s_waitcnt vmcnt(0)"                                                                                                     \
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(d)       \
                 : "v"(off), "v"(offd), "s"(base0), "s"(base1) : "memory")
    if constexpr (LOCAL) RS_LOADS("sc0 nt"); else RS_LOADS("sc1");
#undef RS_LOADS
}

// Experiment builds only (-DOG_SK_TRACE=1): shader-cycle stamps of the phases of iterations 8..15 of every wave of workgroups
// (0,0) and (B/2, G-1), read back by og_debug_sk_trace (scripts/trace_sinkhorn.py)
#ifndef OG_SK_TRACE
#define OG_SK_TRACE 0
#endif
#if OG_SK_TRACE
__device__ unsigned og_sk_trace_buf[2][8][8][16];
#define RS_TP(i) do { if (tsel >= 0 && it >= 8 && it < 16) { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
                      if (lane == 0) og_sk_trace_buf[tsel][wave][it - 8][i] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RS_TP(i) do {} while (0)
#endif

struct SkResArgs {
    const float* S; int64_t lds, strideS;      // raw scores [B][m][lds]
    float* u; int ldu;                         // [B][ldu]: duals of the rows after the first (max-subtracted) iteration, natural units; updated in place
    const float* v_in; float* v_out; int ldv;  // [B][ldv]
    rs_u64* xg;                                // [2][B] exchange areas of RS_XPAIR bytes (parity, pair), zeroed before the launch
    unsigned* status;                          // zeroed before the launch
    unsigned* xcc;                             // [B][8] XCC id + 1 of every workgroup, zeroed before the launch
    int force_agent_scope;                     // experiments / tests: never take the XCD-local path
    int sanitize_pad;                          // n % 4 != 0 and the padding columns [n, lds) of S are the CALLER's (og_sinkhorn): they may
                                               // hold NaN / Inf, which -inf duals do not neutralise -- zero them as the rows are loaded
    const float* zdev; float zhost;
    float inv_reg, la, la_bin, lb, lb_bin;     // natural units (see og_launch_sinkhorn)
    int m, n, mb, iters;                       // mb = rows per workgroup (<= 128), iters = dual-stabilised iterations to run (>= 1)
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float rs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes, result wave-uniform (an SGPR): quad_perm, row_half_mirror, row_mirror, row_bcast15, row_bcast31
__device__ __forceinline__ float rs_wave_sum(float v) {
    v += rs_dpp<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v += rs_dpp<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v += rs_dpp<0x141, 0xF>(v);     // row_half_mirror
    v += rs_dpp<0x140, 0xF>(v);     // row_mirror: every lane of a 16-lane row holds the row total
    v += rs_dpp<0x142, 0xA>(v);     // row_bcast15 into rows 1, 3
    v += rs_dpp<0x143, 0xC>(v);     // row_bcast31 into rows 2, 3: lane 63 holds the wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float rs_dpp_keep(float v) {     // lanes without a source keep their own value (identity for max)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float rs_wave_max(float v) {
    v = fmaxf(v, rs_dpp_keep<0xB1, 0xF>(v));
    v = fmaxf(v, rs_dpp_keep<0x4E, 0xF>(v));
    v = fmaxf(v, rs_dpp_keep<0x141, 0xF>(v));
    v = fmaxf(v, rs_dpp_keep<0x140, 0xF>(v));
    v = fmaxf(v, rs_dpp_keep<0x142, 0xA>(v));
    v = fmaxf(v, rs_dpp_keep<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The 8 register-resident rows of a wave are an ordinary array of 128 floats (fully unrolled, static indices only).  More
// does not work with hipcc: with 10-11 rows its allocator spills 4-5 of them to scratch (= HBM; those rows then cost 3.4x
// the others, scripts/trace_sinkhorn.py), and parking rows in the accumulator registers through "a"-constrained inline asm
// fails the same way (the allocator wants AGPRs for its own overflow and spills the parked values).  Audit after every edit:
// no scratch traffic inside the iteration loop.
__global__ __launch_bounds__(512) void sinkhorn_resident_kernel(SkResArgs a) {
    // The small, hot arrays sit at LOW LDS addresses (ds_read/ds_write immediates are 16 bits: an array beyond 64 KB needs
    // its own address register per access, and those are loop-invariant = live across the whole iteration loop).
    __shared__ __attribute__((aligned(16))) float smem[64 + RS_NG + 4 * RS_NCOL + RS_NW * RS_LR * RS_NCOL];
    float* red = smem;                                     // [64] block reductions
    float* vL = smem + 64;                                 // [RS_NG] current v (base 2); index n = the dustbin column
    float* Pbuf = vL + RS_NG;                              // [4][1024] per-wave column partials, two rounds of four waves
    float* Srows = Pbuf + 4 * RS_NCOL;                     // [wave][RS_LR][1024]

    // grid (B, G): workgroup id = g * B + b, so with B a multiple of 8 the G workgroups of a pair share one XCD (= one L2)
    const int b = blockIdx.x, g = blockIdx.y, B = gridDim.x, G = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.m, N = a.n;
    const float c2 = a.inv_reg * RS_LOG2E;
    const float zr2 = (a.zdev ? a.zdev[0] : a.zhost) * c2;
    const float la2 = a.la * RS_LOG2E, la_bin2 = a.la_bin * RS_LOG2E, lb2 = a.lb * RS_LOG2E, lb_bin2 = a.lb_bin * RS_LOG2E;
    const float* Sb = a.S + (int64_t)b * a.strideS;
    float* ub = a.u + (int64_t)b * a.ldu;
    const int row0 = g * a.mb + wave * RS_RW;              // global row of this wave's slot 0
    const int row_end = min(M, (g + 1) * a.mb);            // rows >= row_end belong to the next workgroup (or do not exist)
    rs_gu32* status = (rs_gu32*)a.status;
    // rows of this wave that exist: slot s is live iff s < nvalid (ONE scalar; sixteen hoisted lane masks cost 32 SGPRs)
    const int nvalid = __builtin_amdgcn_readfirstlane(min(max(row_end - row0, 0), RS_RW));

    // ---- v (natural units, after the max-subtracted first iteration) -> base 2 in LDS.  Columns n..1023 do not exist: their
    //      v is -inf, so their plan entries 2^(x + v + u) vanish whatever (finite) bytes the loads fetched -- the data path
    //      needs no masks.  The dual of the dustbin COLUMN lives at the fixed index 1024. ----
    for (int j = tid; j < RS_NG; j += 512) {
        float t = 0.f;
        if (j < N) t = a.v_in[(int64_t)b * a.ldv + j] * RS_LOG2E;
        else if (j < RS_NCOL) t = OG_NEG_INF;
        else if (j == RS_NCOL) t = a.v_in[(int64_t)b * a.ldv + N] * RS_LOG2E;
        vL[j] = t;
    }

    // ---- my rows: duals and scores, scaled to base 2 once; rows / columns outside the matrix hold -inf (2^-inf = 0) ----
    float ur[RS_RW];
#pragma unroll
    for (int s = 0; s < RS_RW; ++s) {
        const int row = row0 + s;
        const float u0 = row < row_end ? ub[row] * RS_LOG2E : 0.f;
        ur[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u0)));     // wave-uniform: SGPRs
    }
    unsigned ckb[4];                                       // this lane's column chunks (byte offsets), clamped to valid addresses
#pragma unroll
    for (int k = 0; k < 4; ++k) ckb[k] = (unsigned)(4 * lane + 256 * k < N ? 4 * lane + 256 * k : 0) * 4u;
    // raw scores of one row (rows past the end: row 0's bytes, never used).  (scalar row base + 32-bit lane offset) addressing: as
    // 64-bit per-lane addresses the sixteen loop-invariant chunk addresses of the streamed rows took 32 registers and were spilled
    auto load_row_raw = [&](int row, f32x4 (&x)[4]) {
        const uint64_t v = (uint64_t)(uintptr_t)(Sb + (int64_t)(row < row_end ? row : 0) * a.lds);      // wave-uniform: pinned to an SGPR pair
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        const rs_gchar* rp = (const rs_gchar*)(uintptr_t)(((uint64_t)hi32 << 32) | lo);                    // global address space: global_load, not flat_load
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned o = ckb[k];
            asm volatile("" : "+v"(o));                    // re-launder per use: keeps the zero-extension next to the address add
            x[k] = *(const rs_gf32x4*)(rp + o);
        }
        if (a.sanitize_pad) {                              // wave-uniform; only the one chunk that straddles column N has anything to clear
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * lane + 256 * k + e >= N) x[k][e] = 0.f;
        }
    };
    auto load_row = [&](int row, f32x4 (&x)[4]) {          // -> s2 = S * c2 (waits for the data: resident rows only)
        load_row_raw(row, x);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = x[k] * c2;
    };
    f32x4 sr[RS_RR][4];
#pragma unroll
    for (int s = 0; s < RS_RR; ++s) {
        load_row(row0 + s, sr[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < RS_LR; ++s) {
        f32x4 x[4];
        __builtin_amdgcn_sched_barrier(0);                 // one row in flight at a time: the registers are full of S
        load_row(row0 + RS_RR + s, x);
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(Srows + (wave * RS_LR + s) * RS_NCOL + 4 * lane + 256 * k) = x[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    const float* vLl = vL + 4 * lane;                      // this lane's columns: + 256 k (immediate offsets)
    float* Sw = Srows + wave * RS_LR * RS_NCOL + 4 * lane; // this wave's LDS rows: + 1024 s + 256 k (immediates within 16 KB)
#if OG_SK_TRACE
    const int tsel = (b == 0 && g == 0) ? 0 : (b == B / 2 && g == G - 1) ? 1 : -1;
#endif
    // ---- do the G workgroups of this pair sit on ONE XCD (one L2)?  Workgroups are dealt to the XCDs round-robin by linear id, so
    //      with B a multiple of 8 they do -- but that is dispatcher behaviour, not a contract: every workgroup publishes its
    //      XCC id (agent scope) and reads its peers'; all of them see the same table and take the same decision.  A pair that is
    //      spread over XCDs exchanges its granules at agent scope (slower, always correct). ----
    bool failed = false;
    if (tid == 0) {
        const unsigned mine = (__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0x00F0000Fu) + 1u;      // HW_REG_XCC_ID: XCC_ID [3:0], DIE_ID [23:20]
        rs_gu32* tab = (rs_gu32*)a.xcc + (int64_t)b * RS_GMAX;
        __hip_atomic_store(tab + g, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned same = 1u;
        for (int q = 0; q < G; ++q) {
            unsigned x = 0u, spins = 0u;
            while ((x = __hip_atomic_load(tab + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                if (++spins > RS_SPIN_LIMIT) { same = 2u; __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (x != mine && same == 1u) same = 0u;
        }
        red[40] = __builtin_bit_cast(float, same);
    }
    __syncthreads();
    const unsigned xcd_word = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, red[40]));
    const bool xcd_local = xcd_word == 1u && !a.force_agent_scope;
    failed = xcd_word == 2u;
    // the first two streamed rows of an iteration are fetched before the exchange of the previous one (nothing they need
    // depends on it), the other two while the LDS rows are processed
    // ONE prefetch buffer: with two of them live over the register rows the allocator spilled part of a resident row, and every
    // reload (s_waitcnt vmcnt(0)) also waited for the prefetches in flight -- thousands of cycles per iteration
    f32x4 xs0[4];                                          // RAW scores: scaled by c2 when consumed (a multiply here would wait for the data)
    load_row_raw(row0 + RS_RR + RS_LR, xs0);
    // ---- dustbin row dual from the INITIAL v: u_M' = log2 a_M - (z + LSE2_{j<=N} v_j)   (every workgroup, identically).  Inside the
    //      loop the same quantity for the next iteration falls out of the new-v phase, where the new v are still in registers ----
    float uM2;
    {
        const float vN2 = vL[RS_NCOL];
        float mx = vN2;
        for (int j = tid; j < N; j += 512) mx = fmaxf(mx, vL[j]);
        mx = rs_wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = red[0];
#pragma unroll
        for (int w = 1; w < RS_NW; ++w) mx = fmaxf(mx, red[w]);
        float sv = tid == 0 ? __builtin_amdgcn_exp2f(vN2 - mx) : 0.f;
        for (int j = tid; j < N; j += 512) sv += __builtin_amdgcn_exp2f(vL[j] - mx);
        sv = rs_wave_sum(sv);
        if (lane == 0) red[8 + wave] = sv;
        __syncthreads();
        float svt = red[8];
#pragma unroll
        for (int w = 1; w < RS_NW; ++w) svt += red[8 + w];
        uM2 = la_bin2 - (zr2 + mx + __builtin_amdgcn_logf(svt));
        __syncthreads();                                   // red[] is reused by the loop
    }
#pragma unroll 1
    for (int it = 0; it < a.iters; ++it) {
        const unsigned epoch = (unsigned)it + 1u;
        char* xg = (char*)a.xg + ((int64_t)(it & 1) * B + b) * RS_XPAIR;                 // this pair's exchange area of this parity

        // (v is re-read from LDS chunk by chunk inside the rows: 16 more live registers would not fit beside the 176 of S)
        const float vN2 = vL[RS_NCOL];
        const float dcol2 = zr2 + vN2;

        RS_TP(0);
        RS_TP(1);
        // ---- (2) row pass over my 16 rows: new u, column partials with the new u, dustbin-column partial ----
        f32x4 cs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        float usum = 0.f;
        // p = the plan entries of the row; rows held in registers keep their scores (p is a temporary), rows fetched from
        // LDS / memory are overwritten in place
        auto finish_row = [&](f32x4 (&p)[4], float sum, int s) {
            const float u = ur[s];
            const float pd = __builtin_amdgcn_exp2f(dcol2 + u);            // the row's dustbin-column entry with the old u
            const float rowsum = rs_wave_sum(sum) + pd;
            const float un = u + la2 - __builtin_amdgcn_logf(rowsum);
            const float f = __builtin_amdgcn_exp2f(un - u);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[k][e] = __builtin_fmaf(p[k][e], f, cs[k][e]);
            usum += pd * f;                                                 // = 2^(z + v_N + u'), one transcendental less per row
            ur[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, un)));   // wave-uniform: an SGPR
        };
        auto mem_row = [&](f32x4 (&x)[4], int slot, auto RAW_) {      // a row fetched from LDS (scaled) / memory (RAW): its plan entries overwrite it
            constexpr bool RAW = decltype(RAW_)::value;
            if (slot < nvalid) {                           // wave-uniform: rows past the end of the workgroup's range are skipped
                const float u = ur[slot];
                // packed fp32 adds (two elements per VALU instruction); the association (s + v) + u and the pairing of the row sum
                // are part of the arithmetic contract with the streaming kernels' tests (fixed, lane-local)
                const rs_f32x2 uu = {u, u};
                const rs_f32x2 cc = {c2, c2};
                rs_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 vk = *reinterpret_cast<const f32x4*>(vLl + 256 * k);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const rs_f32x2 xe = {x[k][e], x[k][e + 1]}, ve = {vk[e], vk[e + 1]};
                        const rs_f32x2 t = (RAW ? xe * cc + ve : xe + ve) + uu;
                        x[k][e] = __builtin_amdgcn_exp2f(t[0]);
                        x[k][e + 1] = __builtin_amdgcn_exp2f(t[1]);
                        sum2 += rs_f32x2{x[k][e], x[k][e + 1]};
                    }
                }
                finish_row(x, sum2[0] + sum2[1], slot);
            }
            __builtin_amdgcn_sched_barrier(0);             // one row at a time
        };
        // streamed rows 12..15 through the one buffer, each requested three resident rows (~3-4k cycles) before it is consumed:
        //   row 12 (fetched under the previous exchange) | load 13 | reg 0-2 | row 13 | load 14 | reg 3-5 | row 14 | load 15 |
        //   reg 6-7, LDS 0-1 | row 15 | load 12 of the next iteration | LDS 2-3
        auto reg_row = [&](auto R) {
            constexpr int r = decltype(R)::value;
            if (r < nvalid) {
                const float u = ur[r];
                f32x4 x[4];
                const rs_f32x2 uu = {u, u};
                rs_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 vk = *reinterpret_cast<const f32x4*>(vLl + 256 * k);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const rs_f32x2 t = (rs_f32x2{sr[r][k][e], sr[r][k][e + 1]} + rs_f32x2{vk[e], vk[e + 1]}) + uu;
                        x[k][e] = __builtin_amdgcn_exp2f(t[0]);
                        x[k][e + 1] = __builtin_amdgcn_exp2f(t[1]);
                        sum2 += rs_f32x2{x[k][e], x[k][e + 1]};
                    }
                }
                finish_row(x, sum2[0] + sum2[1], r);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto lds_row = [&](auto Sl) {
            constexpr int sl = decltype(Sl)::value;
            f32x4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const f32x4*>(Sw + sl * RS_NCOL + 256 * k);
            mem_row(x, RS_RR + sl, std::false_type{});
        };
        static_assert(RS_RR == 8 && RS_LR == 4 && RS_SR == 4, "the row schedule below is written out for 8 + 4 + 4 rows");
        using std::integral_constant;
        mem_row(xs0, RS_RR + RS_LR, std::true_type{});
        load_row_raw(row0 + RS_RR + RS_LR + 1, xs0);
        __builtin_amdgcn_sched_barrier(0);
        reg_row(integral_constant<int, 0>{}); reg_row(integral_constant<int, 1>{}); reg_row(integral_constant<int, 2>{});
        mem_row(xs0, RS_RR + RS_LR + 1, std::true_type{});
        load_row_raw(row0 + RS_RR + RS_LR + 2, xs0);
        __builtin_amdgcn_sched_barrier(0);
        reg_row(integral_constant<int, 3>{}); reg_row(integral_constant<int, 4>{}); reg_row(integral_constant<int, 5>{});
        mem_row(xs0, RS_RR + RS_LR + 2, std::true_type{});
        load_row_raw(row0 + RS_RR + RS_LR + 3, xs0);
        __builtin_amdgcn_sched_barrier(0);
        reg_row(integral_constant<int, 6>{}); reg_row(integral_constant<int, 7>{});
        RS_TP(2);
        lds_row(integral_constant<int, 0>{}); lds_row(integral_constant<int, 1>{});
        mem_row(xs0, RS_RR + RS_LR + 3, std::true_type{});
        if (it + 1 < a.iters) load_row_raw(row0 + RS_RR + RS_LR, xs0);      // next iteration's first streamed row: under the LDS rows and the exchange
        __builtin_amdgcn_sched_barrier(0);
        lds_row(integral_constant<int, 2>{}); lds_row(integral_constant<int, 3>{});

        RS_TP(3);
        // ---- (3) workgroup column partials: two rounds of four waves through LDS, fixed summation order ----
        float tot[3] = {0.f, 0.f, 0.f};                   // this thread's columns: tid, tid + 512, (tid == 0: index n = the dustbin column)
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if ((wave >> 2) == round) {
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(Pbuf + (wave & 3) * RS_NCOL + 4 * lane + 256 * k) = cs[k];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int w = 0; w < 4; ++w) tot[c] += Pbuf[w * RS_NCOL + tid + 512 * c];
            __syncthreads();
        }
        if (lane == 0) red[16 + wave] = usum;
        __syncthreads();
        RS_TP(4);
        // ---- (4) publish: 8-byte {epoch, value} granules, relaxed stores (the vector L1 is write-through: they land in the L2);
        //      slot g of every column.  Columns >= n publish their (zero) partials too, and workgroup 0 fills the slots G..7
        //      nobody owns with zeros.
        //      (5) sweep the granules of the pair (mine included: same code path, same rounding).  Both columns of a thread in ONE
        //      batch of eight 16-byte loads; the batch is polled as a whole until every granule carries this epoch, so a late peer
        //      costs one more round trip (polled granule by granule, the stale ones of a batch cost up to G sequential round
        //      trips: 7.7k cycles per iteration in the first trace of round 2).  Fixed summation order: slot 0..7. ----
        float colsum[3] = {0.f, 0.f, 0.f};
        auto exchange = [&](auto LOCAL_) __attribute__((always_inline)) {
            constexpr bool LOCAL = decltype(LOCAL_)::value;
            constexpr int SCOPE = LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
            const rs_u64 tag = (rs_u64)epoch << 32;
            auto slot_ptr = [&](int col, int slot) { return (rs_gu64*)(xg + rs_granule_offset(col, slot)); };
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                __hip_atomic_store(slot_ptr(tid + 512 * c, g), tag | __builtin_bit_cast(unsigned, tot[c]), __ATOMIC_RELAXED, SCOPE);
                if (g == 0)
                    for (int q = G; q < RS_GMAX; ++q) __hip_atomic_store(slot_ptr(tid + 512 * c, q), tag, __ATOMIC_RELAXED, SCOPE);
            }
            if (tid == 0) {
                float us = red[16];
#pragma unroll
                for (int w = 1; w < RS_NW; ++w) us += red[16 + w];
                __hip_atomic_store(slot_ptr(RS_NCOL, g), tag | __builtin_bit_cast(unsigned, us), __ATOMIC_RELAXED, SCOPE);
                if (g == 0)
                    for (int q = G; q < RS_GMAX; ++q) __hip_atomic_store(slot_ptr(RS_NCOL, q), tag, __ATOMIC_RELAXED, SCOPE);
            }
            RS_TP(5);
            const rs_gu64* base0 = (const rs_gu64*)xg;                                     // columns tid ...
            const rs_gu64* base1 = (const rs_gu64*)(xg + (512 / 64) * RS_XBLOCK);          // ... and tid + 512: eight blocks further
            const unsigned off = rs_granule_offset(tid, 0);
            const unsigned offd = rs_granule_offset(RS_NCOL, lane & (RS_GMAX - 1));         // lane q (mod 8) holds slot q of the dustbin column
            rs_u32x4 ga[4], gb[4];
            rs_u64 xd;
            unsigned spins = 0;
            for (;;) {
                rs_load_columns<LOCAL>(base0, base1, off, offd, ga, gb, xd);
                unsigned bad = (unsigned)(xd >> 32) ^ epoch;
#pragma unroll
                for (int k = 0; k < 4; ++k) bad |= (ga[k][1] ^ epoch) | (ga[k][3] ^ epoch) | (gb[k][1] ^ epoch) | (gb[k][3] ^ epoch);
                if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0) break;
                if (++spins > RS_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    failed = true;
                    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (LOCAL && (spins & 63u) == 0u) asm volatile("buffer_inv sc1" ::: "memory");     // guaranteed progress: drop the L1 for real
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index, hipcc 7.2)
                const unsigned a0 = ga[k][0], a1 = ga[k][2], b0 = gb[k][0], b1 = gb[k][2];
                colsum[0] += __builtin_bit_cast(float, a0);
                colsum[0] += __builtin_bit_cast(float, a1);
                colsum[1] += __builtin_bit_cast(float, b0);
                colsum[1] += __builtin_bit_cast(float, b1);
            }
            const int dval = (int)(unsigned)xd;                // ordered sum of the eight dustbin terms (used by thread 0)
#pragma unroll
            for (int q = 0; q < RS_GMAX; ++q) colsum[2] += __builtin_bit_cast(float, __builtin_amdgcn_readlane(dval, q));
        };
        if (xcd_local) exchange(std::true_type{}); else exchange(std::false_type{});
        RS_TP(6);
        // ---- (6) new v for my columns (every workgroup of the pair computes the same bits), and -- while they are in registers --
        //      the dustbin-row dual of the NEXT iteration, u_M' = log2 a_M - (z + LSE2 of the new v): per-wave (max, sum) first,
        //      one LDS round for each ----
        float vnew[2] = {OG_NEG_INF, OG_NEG_INF};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = tid + 512 * c;
            if (j < N) {
                const float vo = vL[j];
                vnew[c] = vo + lb2 - __builtin_amdgcn_logf(colsum[c] + __builtin_amdgcn_exp2f(zr2 + vo + uM2));
                vL[j] = vnew[c];
            }
        }
        float vNn = OG_NEG_INF;
        if (tid == 0) {
            vNn = vN2 + lb_bin2 - __builtin_amdgcn_logf(colsum[2] + __builtin_amdgcn_exp2f(zr2 + vN2 + uM2));
            vL[RS_NCOL] = vNn;
            red[32] = uM2;
        }
        {
            float mx = rs_wave_max(fmaxf(fmaxf(vnew[0], vnew[1]), vNn));
            if (lane == 0) red[wave] = mx;
            __syncthreads();
            mx = red[0];
#pragma unroll
            for (int w = 1; w < RS_NW; ++w) mx = fmaxf(mx, red[w]);
            float sv = __builtin_amdgcn_exp2f(vnew[0] - mx) + __builtin_amdgcn_exp2f(vnew[1] - mx) + __builtin_amdgcn_exp2f(vNn - mx);   // 2^-inf = 0
            sv = rs_wave_sum(sv);
            if (lane == 0) red[8 + wave] = sv;
            __syncthreads();
            float svt = red[8];
#pragma unroll
            for (int w = 1; w < RS_NW; ++w) svt += red[8 + w];
            uM2 = la_bin2 - (zr2 + mx + __builtin_amdgcn_logf(svt));
        }
        RS_TP(7);
        if (__syncthreads_or(failed)) break;               // a peer never arrived: leave together (status = 1)
        RS_TP(8);
    }

    // ---- results in natural units: u of my rows, and (workgroup 0 of the pair) v and u_M ----
#pragma unroll
    for (int s = 0; s < RS_RW; ++s)
        if (lane == 0 && row0 + s < row_end) ub[row0 + s] = ur[s] * RS_LN2;
    if (g == 0) {
        for (int j = tid; j < N; j += 512) a.v_out[(int64_t)b * a.ldv + j] = vL[j] * RS_LN2;
        if (tid == 0) a.v_out[(int64_t)b * a.ldv + N] = vL[RS_NCOL] * RS_LN2;
        if (tid == 0) ub[M] = red[32] * RS_LN2;
    }
}

}  // namespace
#if OG_SK_TRACE
extern "C" int og_debug_sk_trace(void* host_dst, size_t bytes) {
    if (bytes > sizeof(og_sk_trace_buf)) bytes = sizeof(og_sk_trace_buf);
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(og_sk_trace_buf), bytes);
}
#endif
namespace {
int rs_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cus;
    }();
    return n;
}

}  // namespace

// workgroups per pair: the smallest G with ceil(m / G) <= 128 rows per workgroup
static inline int rs_groups(int m) { return (m + RS_ROWS - 1) / RS_ROWS; }

bool og_sinkhorn_resident_shape_ok(int B, int m, int n) {
    return B > 0 && m > 0 && n > 0 && n <= RS_NCOL && (int64_t)B * rs_groups(m) <= 4096;
}

size_t og_sinkhorn_resident_ws_bytes(int B, int m, int n) {
    if (!og_sinkhorn_resident_shape_ok(B, m, n)) return 0;
    return (size_t)2 * B * RS_XPAIR + 256 + (size_t)B * RS_GMAX * sizeof(unsigned);      // status word, exchange areas (two parities), XCC table
}

// mode: 1 = when the whole batch is co-resident AND large enough to pay off, 2 = whenever it is co-resident (tests)
bool og_sinkhorn_resident_wanted(int B, int m, int n, int mode) {
    if (mode <= 0 || !og_sinkhorn_resident_shape_ok(B, m, n)) return false;
    const int cus = rs_num_cus();
    if (cus <= 0 || (int64_t)B * rs_groups(m) > cus) return false;          // every workgroup needs its own CU, all at once
    if (mode >= 2) return true;
    return (int64_t)B * m * n >= (int64_t)4 << 20;                           // >= 16 MB of scores: below that the streaming kernels are launch-bound anyway
}

int og_launch_sinkhorn_resident(const float* S, int64_t lds, const float* zdev, float zhost, int B, int m, int n, int iters,
                                float inv_reg, float la, float la_bin, float lb, float lb_bin, float* u, int ldu, const float* v_in,
                                float* v_out, int ldv, void* xws, hipStream_t st, bool trusted_padding) {
    if (!S || !u || !v_in || !v_out || !xws || iters < 1 || !og_sinkhorn_resident_shape_ok(B, m, n)) return OG_E_INVALID;
    const int G = rs_groups(m);
    const size_t bytes = og_sinkhorn_resident_ws_bytes(B, m, n);
    hipError_t e = hipMemsetAsync(xws, 0, bytes, st);                        // epochs start at 1: every tag must read 0 first
    if (e != hipSuccess) return (int)e;
    SkResArgs a{};
    a.S = S; a.lds = lds; a.strideS = (int64_t)m * lds;
    a.u = u; a.ldu = ldu; a.v_in = v_in; a.v_out = v_out; a.ldv = ldv;
    a.xg = (rs_u64*)((char*)xws + 256); a.status = (unsigned*)xws;
    a.xcc = (unsigned*)((char*)xws + 256 + (size_t)2 * B * RS_XPAIR);
    { const char* e = getenv("OG_SINKHORN_AGENT_SCOPE"); a.force_agent_scope = e && atoi(e) != 0; }       // read per call: the tests switch it
    a.zdev = zdev; a.zhost = zhost; a.inv_reg = inv_reg; a.la = la; a.la_bin = la_bin; a.lb = lb; a.lb_bin = lb_bin;
    a.m = m; a.n = n; a.mb = (m + G - 1) / G; a.iters = iters;
    a.sanitize_pad = (n & 3) && !trusted_padding;
    hipLaunchKernelGGL(sinkhorn_resident_kernel, dim3(B, G), dim3(512), 0, st, a);
    return og_launch_status();
}

// Log-domain Sinkhorn with the PLAN MATRIX RESIDENT ON CHIP for all iterations after the first (gfx950).
//
// Replaces the iteration loop of log_otp_solver (reference optimal_transport.py:24-26).  256 CUs x (512 KB of VGPRs + 160 KB of
// LDS) = 172 MB; a round of pairs whose score matrices fill at most 128 MB (256 workgroups x 128 rows x 1024 columns of fp32) is
// loaded ONCE and iterated in ONE launch; larger batches run as a sequence of such rounds (32 pairs of 1024 x 1024, 8 pairs of
// 2048 x 2048 or 2 pairs of 4096 x 4096 per round; ragged pairs packed one column block per XCD slot range).  Round 4 rewrite of the
// round-2/3 kernel, two changes of substance:
//
// (1) LINEAR-DOMAIN resident state, never rewritten.  After the max-subtracted first iteration (sinkhorn.hip) every plan entry
//         P_ij = 2^(s_ij + u_i + v_j),   s = S/reg * log2 e, duals in base 2,
//     is <= max(a_i, b_j) < 1 and stays so under every half-update.  The resident matrix is E = P as it was when the workgroup last
//     EVALUATED it from the scores; the plan of the moment is P_ij = E_ij F_i C_j with the row factors F_i = 2^(u_i - u_i at the
//     evaluation) (one register, a row per lane) and the column factors C_j = 2^(v_j - v_j at the evaluation) (an LDS vector):
//         pass 1:  rowsum_i = F_i sum_j E_ij C_j + 2^(z + v_N + u_i) ;  u_i' = u_i + log2 a_i - log2 rowsum_i ;  F_i *= 2^(u_i' - u_i)
//         pass 2:  colsum_j = C_j sum_i E_ij F_i + 2^(z + v_j + u_M') ;  v_j' = v_j + log2 b_j - log2 colsum_j ;  C_j *= 2^(v_j' - v_j)
//     -- the same recursion as optimal_transport.py:24-26 in exact arithmetic, TWO packed-fp32 fmas per entry and iteration (the
//     round-3 kernel: 2 adds + 1 exponential + 1 add + 1 fma) and no store; the factors are products of the exponentials of the
//     ACTUAL (rounded) dual increments, so the plan tracks the duals (tests/emulate_sinkhorn_linear.py: ~1e-5 on the log-scores
//     after 100 iterations, the level of the reference's own fp32 solver).  What the linear domain cannot represent are entries
//     below 2^-126 at the time of the evaluation: they are zero until the next one, whereas the log-domain recursion could bring
//     them back.  A bound on how far ANY factor product has moved, drift = sum_t (max_i |du_i| + max_j |dv_j|), is kept per
//     workgroup; beyond RS_DRIFT_BITS = 40 bits the workgroup re-reads its rows of S and re-evaluates E = 2^(s + u + v) with
//     F = C = 1 (a no-op in exact arithmetic, so workgroups refresh independently): a lost entry is below 2^-86 then, against
//     marginals of 2^-14.  Ordinary problems never refresh; |S/reg| of several hundred does 5-7 times in 30 iterations.
//     Nothing is streamed from memory inside the loop: 12 of a wave's 16 rows live in registers (192), 4 in LDS.
//
// (2) ANY SIZE UP TO 4096 x 4096: A PAIR IS A 2-D GRID OF WORKGROUP TILES.  A wave always owns a 16-row x 1024-column tile (16 floats
//     per lane and row); the 8 waves of a workgroup form WR x WC wave tiles (W = WC = 1, 2, 4: workgroup tiles of 128 x 1024, 64 x 2048,
//     32 x 4096 entries); a pair is X column blocks x Gx row blocks of such tiles, X Gx <= 128 workgroups (rs_geom: the widest tile whose
//     Gx <= 32 row blocks fit one XCD; 4096 x 4096 = 4 x 32 tiles of 128 x 1024).
//     * Row sums cross the WC waves of a row through LDS (one barrier: pass 1 and pass 2 are decoupled because E is resident) and,
//       when X > 1, the X column blocks of a row block in ONE hop: every tile publishes a record of its 128 partial row sums as 8-byte
//       {epoch, value} granules (cdna_hip_programming.md Guideline 16, form R2: the data is the flag) and reads the X records of its row
//       block at AGENT scope -- 1 KB per tile and iteration is all that crosses XCDs.  (The first version of this round kept the pair's
//       column sums crossing XCDs instead: 32 KB of granules written and 64 KB read per workgroup and iteration at n = 4096 saturated the
//       L2s: 31.6k cycles per iteration against 18.3k now, profiles/r04_e_* vs r04_i_*.)
//     * Column sums cross the Gx row blocks of a column block -- all on one XCD -- in two hops: every workgroup publishes its NC column
//       partials, the OWNER of a slice of columns sums the Gx partials of its columns in a fixed order and publishes the totals,
//       everybody reads the NC totals: 2 NC granules read per workgroup and iteration whatever Gx is (the round-3 kernel read Gx x NC).
//       Every workgroup of the column block then computes the new v_j of the block's columns redundantly and bit-identically.
//     * The dustbin column's dual lives in column block 0 and travels to the other blocks in the row record, as do the blocks' shares of
//       the log-sum-exp of v that the dustbin ROW's dual needs.
//     Granules are double-buffered by epoch parity; every spin is bounded (status word: 1 = a wait timed out -> the safety-net
//     kernel of sinkhorn.hip recomputes the batch); all workgroups of a round must be co-resident: one 512-thread workgroup per CU
//     (~150 KB of LDS), at most #CUs workgroups per launch, checked by the launcher.
//     Workgroups are dealt to the XCDs round-robin by linear id; the id -> (pair, block, row block) map puts a column block on ONE
//     XCD, and its column exchange then goes through that XCD's L2 (verified at run time; otherwise agent scope).
#include <stdlib.h>
#include <math.h>
#include <type_traits>

#include "og_common.h"

namespace {

constexpr int RS_RW = 16;              // row slots per wave
constexpr int RS_RR = 12;              //   slots 0..11: rows held in registers
constexpr int RS_LR = 4;               //   slots 12..15: rows held in LDS
static_assert(RS_RR + RS_LR == RS_RW, "sixteen row slots per wave");
constexpr int RS_NW = 8;               // waves per workgroup
constexpr int RS_SEG = 1024;           // columns of a wave's tile (16 per lane)
constexpr int RS_PAD = 64;             // granule rows are NC + 64 long: the dustbin column sits at index NC
constexpr int RS_RREC = 136;           // granules of a tile's row record: 128 partial row sums, (sum, shift) of 2^v, the dustbin column's dual
constexpr int RS_MAXWG = 256;          // workgroups per launch at most (= CUs of the part; the launcher checks the real count)
constexpr int RS_MAXPAIRS = 128;       // pairs per launch at most (G >= 2)
constexpr unsigned RS_SPIN_LIMIT = 1u << 21;
constexpr float RS_DRIFT_BITS = 40.f;
constexpr float RS_LOG2E = 1.4426950408889634f;
constexpr float RS_LN2 = 0.6931471805599453f;

typedef unsigned long long rs_u64;
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) rs_u64 rs_gu64;
typedef __attribute__((address_space(1))) unsigned rs_gu32;
typedef __attribute__((address_space(1))) char rs_gchar;
typedef __attribute__((address_space(1))) f32x4 rs_gf32x4;

// A batch of 8-byte granules from the L2 -- never from this CU's vector L1 (an aligned 8-byte granule cannot tear).  The wait is
// part of the block: the compiler does not count the loads of an asm statement, and a result register may be copied the moment the
// statement ends (loads and wait in separate statements returned the registers' OLD contents now and then -- {epoch, 0} where they
// had been initialised with the tag: right epoch, wrong value).
//   agent scope (sc1): coherent over the whole device -- on this multi-XCD part every such load goes to the fabric behind the
//   per-XCD L2s (which are not coherent with each other);
//   XCD-local (LOCAL): the workgroups of a pair share one XCD = one L2 (verified at run time, kernel prologue).  Workgroup-scope
//   streaming loads (sc0 nt) do not keep their line in the vector L1, so a re-poll reads the L2 again.  (sc0 alone may hit a stale
//   L1 line for ever; buffer_inv sc1 is the guaranteed-progress fallback of the poll loops, every 64th poll.)
#define RS_LD(i, o) "global_load_dwordx2 %" #i ", %" #o ", %[b] " 
template <bool LOCAL, int N>
__device__ __forceinline__ void rs_load_granules(rs_u64 (&d)[N], const rs_gchar* base, const unsigned (&off)[N]) {
    static_assert(N == 3 || N == 5 || N == 9, "2 W + 1 granules");
#define RS_M3(M) asm volatile(RS_LD(0, 3) M "\n\t" RS_LD(1, 4) M "\n\t" RS_LD(2, 5) M "\n\ts_waitcnt vmcnt(0)"                                       \
                              : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), [b] "s"(base) : "memory")
#define RS_M5(M) asm volatile(RS_LD(0, 5) M "\n\t" RS_LD(1, 6) M "\n\t" RS_LD(2, 7) M "\n\t" RS_LD(3, 8) M "\n\t" RS_LD(4, 9) M "\n\ts_waitcnt vmcnt(0)"      \
                              : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4])                                                \
                              : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), [b] "s"(base) : "memory")
#define RS_M9(M) asm volatile(RS_LD(0, 9) M "\n\t" RS_LD(1, 10) M "\n\t" RS_LD(2, 11) M "\n\t" RS_LD(3, 12) M "\n\t" RS_LD(4, 13) M "\n\t"                 \
                              RS_LD(5, 14) M "\n\t" RS_LD(6, 15) M "\n\t" RS_LD(7, 16) M "\n\t" RS_LD(8, 17) M "\n\ts_waitcnt vmcnt(0)"                  \
                              : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8])  \
                              : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]),   \
                                [b] "s"(base) : "memory")
    if constexpr (N == 3) { if constexpr (LOCAL) RS_M3("sc0 nt"); else RS_M3("sc1"); }
    else if constexpr (N == 5) { if constexpr (LOCAL) RS_M5("sc0 nt"); else RS_M5("sc1"); }
    else { if constexpr (LOCAL) RS_M9("sc0 nt"); else RS_M9("sc1"); }
#undef RS_M3
#undef RS_M5
#undef RS_M9
}
#undef RS_LD

// A thread's NCOL adjacent columns of one granule row: 16-byte stores of two {value, epoch} granules (a torn 16-byte store is two
// whole granules), and the matching sweep: NCOL / 2 16-byte loads + the dustbin column's granule, polled as one batch.
typedef unsigned rs_u32x4 __attribute__((ext_vector_type(4)));
template <bool LOCAL>
__device__ __forceinline__ void rs_store_pair(const rs_gchar* base, unsigned off, float v0, float v1, unsigned epoch) {
    const rs_u32x4 q = {__builtin_bit_cast(unsigned, v0), epoch, __builtin_bit_cast(unsigned, v1), epoch};
    // (the trailing s_nop: a VMEM store of more than 8 bytes must not be followed at once by a write of its data registers; the
    // compiler's hazard recognizer does not look inside inline asm, and it does reuse q for the next pair)
    if constexpr (LOCAL) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(off), "v"(q), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(off), "v"(q), "s"(base) : "memory");
}
template <bool LOCAL, int NQ>
__device__ __forceinline__ void rs_load_totals(rs_u32x4 (&q)[NQ], rs_u64& d, const rs_gchar* base, unsigned off, unsigned offd) {
    static_assert(NQ == 1 || NQ == 2 || NQ == 4, "2, 4 or 8 columns per thread");
#define RS_T1(M) asm volatile("global_load_dwordx4 %0, %2, %4 " M "\n\tglobal_load_dwordx2 %1, %3, %4 " M "\n\ts_waitcnt vmcnt(0)"                      \
                              : "=&v"(q[0]), "=&v"(d) : "v"(off), "v"(offd), "s"(base) : "memory")
#define RS_T2(M) asm volatile("global_load_dwordx4 %0, %3, %5 " M "\n\tglobal_load_dwordx4 %1, %3, %5 offset:16 " M "\n\t"                           \
                              "global_load_dwordx2 %2, %4, %5 " M "\n\ts_waitcnt vmcnt(0)"                                                         \
                              : "=&v"(q[0]), "=&v"(q[1]), "=&v"(d) : "v"(off), "v"(offd), "s"(base) : "memory")
#define RS_T4(M) asm volatile("global_load_dwordx4 %0, %5, %7 " M "\n\tglobal_load_dwordx4 %1, %5, %7 offset:16 " M "\n\t"                           \
                              "global_load_dwordx4 %2, %5, %7 offset:32 " M "\n\tglobal_load_dwordx4 %3, %5, %7 offset:48 " M "\n\t"                \
                              "global_load_dwordx2 %4, %6, %7 " M "\n\ts_waitcnt vmcnt(0)"                                                         \
                              : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(d) : "v"(off), "v"(offd), "s"(base) : "memory")
    if constexpr (NQ == 1) { if constexpr (LOCAL) RS_T1("sc0 nt"); else RS_T1("sc1"); }
    else if constexpr (NQ == 2) { if constexpr (LOCAL) RS_T2("sc0 nt"); else RS_T2("sc1"); }
    else { if constexpr (LOCAL) RS_T4("sc0 nt"); else RS_T4("sc1"); }
#undef RS_T1
#undef RS_T2
#undef RS_T4
}

// Experiment builds only (-DOG_SK_TRACE=1): shader-cycle stamps of the phases of iterations 8..15 of every wave of the first and
// the last workgroup of the launch, read back by og_debug_sk_trace (scripts/trace_sinkhorn.py)
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

// ---- which (pair, g) a workgroup is ----
// Uniform batches: arithmetic.  The G workgroups of a pair form X GROUPS of Gx (X = 1, 2, 4, 8; Gx <= 32), one group per XCD when the
// part has 8 XCDs x 32 CUs: XCD x = id % 8, slot q = id / 8; layer = q / Gx, pair = layer * (8 / X) + x / X, group = x % X,
// g = group * Gx + q % Gx.  layers == 0: pairs take G consecutive ids instead (few CUs; agent scope everywhere).
struct RsUniform {
    int G, Gx, X, layers;
};
// Ragged batches: per-pair sizes and the id -> (pair, g) table by value in the kernarg segment (no device-side table, no copy)
struct RsRagged {                      // 32-bit entries only: the 16-bit tables of the first version came back wrong for ODD indices
    int wg[RS_MAXWG];                  // (pair << 8) | g, -1 = idle        (dynamic indexing of sub-dword kernel-argument arrays, hipcc 7.2)
    int gb[OG_MAX_RAGGED];             // pair of the round -> pair of the batch
    int gbase[OG_MAX_RAGGED], G[OG_MAX_RAGGED];      // (one group per pair: G <= 32)
    int m[OG_MAX_RAGGED], n[OG_MAX_RAGGED];
};

struct SkResArgs {
    const float* S; int64_t lds, strideS;      // raw scores [B][m][lds]
    float* u; int ldu;                         // [B][ldu]: duals of the rows after the first (max-subtracted) iteration, natural units; updated in place
    const float* v_in; float* v_out; int ldv;  // [B][ldv]
    char* xa;                                  // [2][slots][NC + 64] granules: column partials, one row per workgroup   } zeroed before
    char* xb;                                  // [2][groups][NC + 64] granules: column totals, one row per group of a pair } the launch
    char* xc;                                  // [2][groups][32][RS_RREC] granules: the row records of the tiles (pairs of X > 1 column blocks)
    unsigned* status;                          // 0 / 1 = a wait timed out (sticky over the rounds of a call)
    unsigned* xcc;                             // [slots] XCC id + 1 of every workgroup, zeroed before the launch
    int force_agent_scope;                     // experiments / tests: never take the XCD-local path
    int sanitize_pad;                          // n % 4 != 0 and the padding columns [n, lds) of S are the CALLER's (og_sinkhorn): they may
                                               // hold NaN / Inf, which -inf duals do not neutralise -- zero them as the rows are loaded
    const float* zdev; float zhost;
    float inv_reg, la, la_bin, lb, lb_bin;     // natural units (uniform batches; ragged ones derive them from the pair's sizes)
    int m, n;                                  // uniform sizes
    int b0, npairs, slots, groups;             // this round: first pair of the batch, pairs, workgroups with a pair (= rows of xa), groups (rows of xb, xc)
    int iters;                                 // dual-stabilised iterations to run (>= 1)
    int local_ok;                              // the id map puts every pair on one XCD (to be verified)
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
// Eight rows at once: p[k] = this lane's partial sum of row k.  Returns, in EVERY lane, the total of row (lane & 7): three
// reduce-scatter steps (the lane pair / quad / octet exchanges what the other half keeps: 8 -> 4 -> 2 -> 1 registers), then plain
// sums over the remaining lane bits.  25 vector instructions against 8 x 13 for eight 64-lane trees; fixed order.
__device__ __forceinline__ float rs_reduce8(const float (&p)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float t[4], w[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (b0 ? p[2 * k + 1] : p[2 * k]) + rs_dpp<0xB1, 0xF>(b0 ? p[2 * k] : p[2 * k + 1]);        // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < 2; ++k) w[k] = (b1 ? t[2 * k + 1] : t[2 * k]) + rs_dpp<0x4E, 0xF>(b1 ? t[2 * k] : t[2 * k + 1]);        // quad_perm [2,3,0,1]
    const float send = b2 ? w[0] : w[1];
    float z = (b2 ? w[1] : w[0]) + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F));   // lane ^ 4
    z += rs_dpp<0x128, 0xF>(z);                                                  // row_ror:8 = lane ^ 8
    {
        const unsigned zi = __builtin_bit_cast(unsigned, z);
        const auto r16 = __builtin_amdgcn_permlane16_swap(zi, zi, false, false);  // rows of 16 lanes: (z0, z0, z2, z2) and (z1, z1, z3, z3)
        const unsigned a0 = r16[0], a1 = r16[1];
        z = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
        const unsigned zj = __builtin_bit_cast(unsigned, z);
        const auto r32 = __builtin_amdgcn_permlane32_swap(zj, zj, false, false);  // halves: (lo, lo) and (hi, hi)
        const unsigned c0 = r32[0], c1 = r32[1];
        z = __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
    }
    return z;
}
__device__ __forceinline__ float rs_uniform(float v) {      // a value every lane holds -> an SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Everything a thread derives from its id -- lane, column offsets, LDS addresses, its place in an owner's sweep -- is RE-DERIVED from
// an opaque copy of the id where it is used: hoisted out of the iteration loop each of these would hold a register for the whole
// loop (next to the 192 of E), i.e. be spilled and re-loaded from scratch memory in every iteration.
#define RS_THREAD_LOCALS                                                                                                        \
    int tq_ = threadIdx.x;                                                                                                     \
    asm volatile("" : "+v"(tq_));                                                                                              \
    const int tid = tq_, lane = tid & 63;                                                                                      \
    const int colb = wc * RS_SEG + 4 * lane;                  /* my lane's 16 columns of the pair: colb + 256 k + e */          \
    const float* Xl = X + colb;                               /* ... of X: + 256 k (immediate offsets) */                       \
    float* Sw = Srows + wave * RS_LR * RS_SEG + 4 * lane;     /* this wave's LDS rows: + 1024 s + 256 k (immediates within 16 KB) */ \
    const int ocol = (gl << cwl) + (tid & (CW - 1)), oslot0 = tid >> cwl;                                                      \
    (void)colb; (void)Xl; (void)Sw; (void)ocol; (void)oslot0

// The 12 register-resident rows of a wave are an ordinary array of 192 floats (fully unrolled, static indices only).
// Audit after every edit: no scratch traffic inside the iteration loop (hipcc -Rpass-analysis=kernel-resource-usage).
// RW = row slots per wave: 16 (12 rows in registers + 4 in LDS: the chip holds 128 MB of plan entries), or 4 / 8 -- the FEW-PAIRS geometries
// (round 5): a launch of one to eight pairs of <= 1024 x 1024 keypoints occupies 8 .. 64 of the 256 CUs with 16-row waves and spends 4.9k of its
// 12.7k cycles per iteration in the two passes over its 128 x 1024 tile (profiles/r04_e_sinkhorn_lazy_trace.log); with 4 rows per wave a pair is
// 32 workgroup tiles of 32 x 1024 (all rows in registers, a quarter of the arithmetic per workgroup, the exchange unchanged: Gx = 32 row blocks
// on one XCD, as the 2048 x 2048 pairs have them).
template <int W, class MAP, int RW = 16>
__global__ __launch_bounds__(512) void sinkhorn_resident_kernel(SkResArgs a, MAP map) {
    static_assert(RW == 16 || RW == 8 || RW == 4, "row slots per wave");
    constexpr int RS_RW = RW, RS_RR = RW == 16 ? 12 : RW, RS_LR = RS_RW - RS_RR;      // (shadow the namespace-scope defaults inside this kernel)
    constexpr int NHALF = (RS_RW + 7) / 8; // batches of eight rows in the row reductions
    constexpr int NC = RS_SEG * W;         // columns on chip
    constexpr int NCX = NC + RS_PAD;       // granules per row of the exchange areas
    constexpr int WC = W, WR = RS_NW / W;  // wave tiles of the workgroup: WR rows x WC columns
    constexpr int RB = RS_RW * WR;         // rows per workgroup
    constexpr int PS = 4 / W;              // column-partial buffers of NC floats in the 16 KB of Pbuf
    constexpr int CPT = NC / 512;          // columns per thread in the phases that own columns: tid CPT + c, c < CPT (adjacent: 16-byte granule pairs)
    constexpr int SB = 2 * W;              // granules per thread and batch of an owner's sweep (at most two batches: < 4 W slots per thread)
    static_assert(W == 1 || W == 2 || W == 4, "1024, 2048 or 4096 columns");
    (void)RB;

    // The small, hot arrays sit at LOW LDS addresses (ds_read/ds_write immediates are 16 bits).
    __shared__ __attribute__((aligned(16))) float smem[256 + 512 + 256 + (W == 1 ? NC : 0) + PS * NC + RS_NW * RS_LR * RS_SEG];
    float* red = smem;                                     // [256] block reductions: [0,8) max, [8,16) sums, [16,24) dustbin-column partials,
                                                           //   [24,32) + [88,96) row drift (by parity), [32] u_M, [33] v_N, [40,48) column drift,
                                                           //   [48,56) time-out flags, [128, 256) row partials [wave][16]
    float* osum = smem + 256;                              // [512] owner: per-thread partial sums of a column's slots
    float* dred = osum + 512;                              // [256] owner 0: the dustbin-column partials of the G workgroups
    float* Pbuf = dred + 256 + (W == 1 ? NC : 0);          // [PS][NC] per-wave column partials, two rounds
    float* X = W == 1 ? dred + 256 : Pbuf;                 // [NC] the column factors C_j, re-written every iteration (W > 1: a barrier separates its
                                                           //   last read from the first write of Pbuf, so they share the space)
    float* Srows = Pbuf + PS * NC;                         // [wave][RS_LR][1024]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wr = wave / WC;

    // ---- which pair, which of its G workgroups ----
    int r, g, G, gbase, bglob, M, N;
    int XG = 1, Gx, xg = 0;                                 // groups of the pair, workgroups per group, my group
    bool local_hint;
    if constexpr (std::is_same<MAP, RsUniform>::value) {
        const int id = blockIdx.x;
        G = map.G; Gx = map.Gx; XG = map.X;
        if (map.layers > 0) {
            const int x = id & 7, q = id >> 3;
            r = (q / Gx) * (8 / XG) + x / XG; xg = x % XG; g = xg * Gx + q % Gx;
        } else { r = id / G; g = id % G; xg = g / Gx; }
        if (r >= a.npairs) return;                         // nobody waits for a workgroup without a pair
        gbase = r * G; bglob = a.b0 + r; M = a.m; N = a.n;
        local_hint = a.local_ok != 0;
    } else {
        const int e = map.wg[blockIdx.x];
        if (e < 0) return;
        r = e >> 8; g = e & 255;
        G = map.G[r]; Gx = G; gbase = map.gbase[r]; bglob = map.gb[r]; M = map.m[r]; N = map.n[r];
        local_hint = true;
    }
    const int gl = g - xg * Gx;                            // my index inside the group (= my row block)
    const int grow = r * XG + xg;                           // my group's row of xb
    // column blocks: group xg of a pair owns the columns [xg NC, xg NC + NC) of its matrices -- a pair wider than one workgroup tile is
    // a 2-D grid of tiles (XG column blocks x Gx row blocks); Nl = my columns that exist
    const int c0 = xg * NC;
    const int Nl = min(max(N - c0, 0), NC);
    float la = a.la, la_bin = a.la_bin, lb = a.lb, lb_bin = a.lb_bin;
    if constexpr (!std::is_same<MAP, RsUniform>::value) {  // as the ragged streaming kernels (sinkhorn.hip)
        const float norm = -__logf((float)(M + N));
        la = norm; lb = norm; la_bin = norm + __logf((float)N); lb_bin = norm + __logf((float)M);
    }
    const int mb = (M + Gx - 1) / Gx;                      // rows per workgroup (<= RB): the rows are split over the Gx row blocks of a column block
    // (wave-uniform floats computed by the vector ALU stay in VECTOR registers unless moved: every one of these would cost a VGPR
    // for the whole loop, next to the 192 of E)
    const float c2 = rs_uniform(a.inv_reg * RS_LOG2E);
    const float zr2 = rs_uniform((a.zdev ? a.zdev[0] : a.zhost) * c2);
    const float la2 = rs_uniform(la * RS_LOG2E), la_bin2 = rs_uniform(la_bin * RS_LOG2E), lb2 = rs_uniform(lb * RS_LOG2E), lb_bin2 = rs_uniform(lb_bin * RS_LOG2E);
    const float* Sb = a.S + (int64_t)bglob * a.strideS;
    float* ub = a.u + (int64_t)bglob * a.ldu;
    const int row0 = gl * mb + wr * RS_RW;                 // global row of this wave's slot 0
    const int row_end = min(M, (gl + 1) * mb);             // rows >= row_end belong to the next workgroup (or do not exist)
    rs_gu32* status = (rs_gu32*)a.status;
    // rows of this wave that exist: slot s is live iff s < nvalid (ONE scalar)
    const int nvalid = __builtin_amdgcn_readfirstlane(min(max(row_end - row0, 0), RS_RW));

    // ---- owner geometry: workgroup gl of a group sums the group's partials of columns [gl CW, (gl + 1) CW), CW = the power of two >= NC / Gx, <= 512 ----
    int cwl = 0;
    while ((Gx << cwl) < NC) ++cwl;                        // log2 CW
    const int CW = 1 << cwl, TPC = 512 >> cwl;             // threads per column
    const bool owner = (gl << cwl) < NC;
    const int spt = (Gx + TPC - 1) / TPC;                  // slots per thread: s = tid / CW + i TPC  (< 4 W)

    // ---- duals of my rows (base 2, wave-uniform) and of my columns (tid + 512 c) ----
    // per-row scalars live ACROSS THE LANES of one register: lane s < 16 holds the value of row slot s (sixteen wave-uniform copies of
    // each would cost 48 SGPRs and sixteen-fold scalar arithmetic)
    float urv = (lane < RS_RW && row0 + lane < row_end) ? ub[row0 + lane] * RS_LOG2E : 0.f;
    auto lane_value = [](float v, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s)); };
    float vv[CPT];                                         // v of my columns (base 2)
    float cj[CPT];                                         // ... and their factors C_j since the evaluation
    float Frv = 1.f;                                       // lane s: F of row slot s
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int j = tid * CPT + c;
        vv[c] = j < Nl ? a.v_in[(int64_t)bglob * a.ldv + c0 + j] * RS_LOG2E : OG_NEG_INF;
    }
    if (tid == 0) red[33] = a.v_in[(int64_t)bglob * a.ldv + N] * RS_LOG2E;      // the dual of the dustbin COLUMN
    __syncthreads();
    float vN2 = rs_uniform(red[33]);

    // ---- do the Gx workgroups of my group sit on ONE XCD (one L2)?  Dispatcher behaviour, not a contract: every workgroup
    //      publishes its XCC id (agent scope) and reads its peers'; all of them see the same table and take the same decision.
    //      A pair that is spread over XCDs exchanges its granules at agent scope (slower, always correct). ----
    bool failed = false;
    bool xcd_local = false;
    if (local_hint && !a.force_agent_scope) {
        const unsigned mine = (__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0x00F0000Fu) + 1u;      // HW_REG_XCC_ID: XCC_ID [3:0], DIE_ID [23:20]
        rs_gu32* tab = (rs_gu32*)a.xcc + gbase + xg * Gx;   // my group's entries
        if (tid == 0) __hip_atomic_store(tab + gl, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned verdict = 1u;                             // 1 same, 0 different, 2 timed out
        if (tid < Gx) {
            unsigned x = 0u, spins = 0u;
            while ((x = __hip_atomic_load(tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                if (++spins > RS_SPIN_LIMIT) { verdict = 2u; __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (verdict == 1u && x != mine) verdict = 0u;
        }
        failed = __syncthreads_or(verdict == 2u) != 0;
        xcd_local = __syncthreads_and(verdict == 1u) != 0;
    }

    // raw scores of one row segment (rows past the end: row 0's bytes, never used).  (scalar row base + 32-bit lane offset) addressing
    auto load_row_raw = [&](int row, f32x4 (&x)[4], int colb) {
        const uint64_t v = (uint64_t)(uintptr_t)(Sb + (int64_t)(row < row_end ? row : 0) * a.lds);      // wave-uniform: pinned to an SGPR pair
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        const rs_gchar* rp = (const rs_gchar*)(uintptr_t)(((uint64_t)hi32 << 32) | lo);                    // global address space: global_load, not flat_load
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned o = (unsigned)(colb + 256 * k < Nl ? c0 + colb + 256 * k : 0) * 4u;      // clamped to a valid address
            x[k] = *(const rs_gf32x4*)(rp + o);
        }
        if (a.sanitize_pad) {                              // wave-uniform; only the one chunk that straddles column N has anything to clear
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (colb + 256 * k + e >= Nl) x[k][e] = 0.f;
        }
    };

#if OG_SK_TRACE
    const int tsel = blockIdx.x == 0 ? 0 : (r == a.npairs - 1 && g == G - 1) ? 1 : -1;
#endif
    f32x4 er[RS_RR][4];                                    // the register-resident rows of E
    float uM2 = 0.f;                                       // dual of the dustbin ROW (every workgroup, identically)
    float vsh = 0.f;                                       // max_j v_j of the previous iteration: the shift of the log-sum-exp of v
    float lse_sum = 0.f, lse_shift = 0.f;                  // XG > 1: sum_j 2^(v_j - lse_shift) over MY column block, published at the next row hop
    float drift = 0.f;                                     // bound (bits) on the growth of any of my entries since they were evaluated
    int it = 0;
    while (!failed) {
        RS_THREAD_LOCALS;
        // ================= (re)evaluate E = 2^(s + v + u) from the scores: at the start, and after a refresh request =================
        // columns n..NC-1 do not exist: their v is -inf, so their entries vanish whatever (finite) bytes the loads fetched
#pragma unroll
        for (int c = 0; c < CPT; ++c) X[tid * CPT + c] = vv[c];
        __syncthreads();
        {
            f32x4 xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4*>(Xl + 256 * k);
#pragma unroll
            for (int s = 0; s < RS_RW; ++s) {
                // no branches around the register rows (conditional in-place updates of the 192-register array make the compiler copy it
                // through scratch at every join): rows past the end read row 0's bytes with u = -inf, i.e. hold zeros
                if (s < RS_RR || s < nvalid) {
                    f32x4 x[4];
                    load_row_raw(row0 + s, x, colb);
                    const float u = s < nvalid ? lane_value(urv, s) : OG_NEG_INF;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][e] = __builtin_amdgcn_exp2f((x[k][e] * c2 + xv[k][e]) + u);
                    if (s < RS_RR) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) er[s < RS_RR ? s : 0][k] = x[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(Sw + (s - RS_RR) * RS_SEG + 256 * k) = x[k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // one row in flight at a time: the registers are full of E
            }
        }
        // the dustbin-row dual from the current v: u_M' = log2 a_M - (z + LSE2_{j<=N} v_j)
        {
            float mx = (tid == 0 && xg == 0) ? vN2 : OG_NEG_INF;      // (the dustbin column's dual enters the sum of column block 0)
#pragma unroll
            for (int c = 0; c < CPT; ++c) mx = fmaxf(mx, vv[c]);
            mx = rs_wave_max(mx);
            __syncthreads();                               // every wave has read X (the v of its columns)
            if (lane == 0) red[wave] = mx;
#pragma unroll
            for (int c = 0; c < CPT; ++c) { cj[c] = 1.f; X[tid * CPT + c] = 1.f; }      // C = 1 (and F = 1) right after an evaluation
            __syncthreads();
            mx = red[0];
#pragma unroll
            for (int w = 1; w < RS_NW; ++w) mx = fmaxf(mx, red[w]);
            mx = fmaxf(mx, -1.0e30f);                      // (a column block without columns: every v is -inf; keep the shift finite)
            float sv = (tid == 0 && xg == 0) ? __builtin_amdgcn_exp2f(vN2 - mx) : 0.f;
#pragma unroll
            for (int c = 0; c < CPT; ++c) sv += __builtin_amdgcn_exp2f(vv[c] - mx);     // 2^-inf = 0
            sv = rs_wave_sum(sv);
            if (lane == 0) red[8 + wave] = sv;
            __syncthreads();
            float svt = red[8];
#pragma unroll
            for (int w = 1; w < RS_NW; ++w) svt += red[8 + w];
            uM2 = rs_uniform(la_bin2 - (zr2 + mx + __builtin_amdgcn_logf(svt)));       // (XG > 1: replaced at the first row hop by the sum over the column blocks)
            vsh = rs_uniform(mx);
            lse_sum = rs_uniform(svt); lse_shift = vsh;
            __syncthreads();                               // red[] is reused by the loop
        }
        drift = 0.f;
        Frv = 1.f;

        bool refresh = false;
#pragma unroll 1
        for (; it < a.iters && !refresh; ++it) {
            RS_THREAD_LOCALS;
            const unsigned epoch = (unsigned)it + 1u;
            const rs_gchar* xa_par = (const rs_gchar*)a.xa + (int64_t)(it & 1) * a.slots * NCX * 8;          // this parity's areas
            const rs_gchar* xb_par = (const rs_gchar*)a.xb + ((int64_t)(it & 1) * a.groups + grow) * NCX * 8;   // ... and my group's totals
            // ... and the row records of the pair's tiles: [column block][row block][RS_RREC granules]
            const rs_gchar* xr_par = (const rs_gchar*)a.xc + ((int64_t)(it & 1) * a.groups + r * XG) * (int64_t)(32 * RS_RREC * 8);

            RS_TP(0);
            // ---- (1) pass 1: sum_j E_ij C_j of my 16 rows ----
            float rsv;                                     // lane l: sum_j E C of slot l & 15 (times F below)
            {
                f32x4 xr[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) xr[k] = *reinterpret_cast<const f32x4*>(Xl + 256 * k);
                float zA = 0.f, zB = 0.f;
#pragma unroll
                for (int half = 0; half < NHALF; ++half) {
                    float ps[8];                               // this lane's partial sums of eight rows
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int s = 8 * half + q;
                        rs_f32x2 sum2 = {0.f, 0.f};
                        if (s < RS_RR) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int e = 0; e < 4; e += 2) {
                                    sum2 += rs_f32x2{er[s < RS_RR ? s : 0][k][e], er[s < RS_RR ? s : 0][k][e + 1]} * rs_f32x2{xr[k][e], xr[k][e + 1]};
                                }
                        } else if (s < RS_RW && s < nvalid) {
                            const int sl = s - RS_RR;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const f32x4 x = *reinterpret_cast<const f32x4*>(Sw + sl * RS_SEG + 256 * k);
#pragma unroll
                                for (int e = 0; e < 4; e += 2) sum2 += rs_f32x2{x[e], x[e + 1]} * rs_f32x2{xr[k][e], xr[k][e + 1]};
                            }
                        }
                        ps[q] = sum2[0] + sum2[1];
                    }
                    const float z = rs_reduce8(ps, lane);
                    if (half == 0) zA = z; else zB = z;
                }
                rsv = (NHALF == 2 && (lane & 8)) ? zB : zA;
            }
            RS_TP(1);
            if constexpr (WC > 1) {                        // the rows cross WC waves: partial sums through LDS, fixed order
                if (lane < RS_RW) red[128 + wave * RS_RW + lane] = rsv;
                __syncthreads();
                if (lane < RS_RW) {
                    const float* rq = red + 128 + wr * WC * RS_RW + lane;
                    rsv = rq[0];
#pragma unroll
                    for (int c = 1; c < WC; ++c) rsv += rq[c * RS_RW];
                }
            }
            float rowp = Frv * rsv;                        // lane s: the PLAN's row sum over my columns (F is this workgroup's own)
            if (XG > 1) {
                // ---- the row hop: a row crosses the XG column blocks of its row block.  Every tile publishes a record of RS_RREC granules
                //      -- its 128 partial row sums, the (sum, shift) of 2^v over its columns (for the dustbin-row dual) and, column block
                //      0, the dual of the dustbin column -- and reads the XG records of its row block; sums in block order, so the XG
                //      tiles arrive at the same bits.  AGENT scope: the column blocks of a pair sit on different XCDs (the bulk of the
                //      traffic -- the column exchange of a block -- stays inside one). ----
                const rs_u64 tag = (rs_u64)epoch << 32;
                rs_gu64* mine = (rs_gu64*)(xr_par + ((int64_t)xg * 32 + gl) * (RS_RREC * 8));
                if (wc == 0 && lane < RS_RW) __hip_atomic_store(mine + wr * RS_RW + lane, tag | __builtin_bit_cast(unsigned, rowp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wave == 0 && lane >= 16 && lane < 19) {
                    const float ex = lane == 16 ? lse_sum : lane == 17 ? lse_shift : vN2;
                    __hip_atomic_store(mine + 128 + (lane - 16), tag | __builtin_bit_cast(unsigned, ex), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // lane s < 16: granule wr 16 + s of every block; lane 16 + 3 b + k: extra k of block b; the others: an own granule again
                float got[4];
                auto hop = [&](auto N_) {
                    constexpr int NX = decltype(N_)::value;                       // XG + 1 loads (the last one a repeat)
                    rs_u64 gx[NX];
                    unsigned xo[NX];
                    const int e3 = lane - 16, eb = e3 / 3, ek = e3 - 3 * eb;
                    const bool isx = lane >= 16 && eb < XG;
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const int b = i < NX - 1 ? i : 0;
                        const int gi = lane < RS_RW ? wr * RS_RW + lane : isx ? 128 + ek : wr * RS_RW;
                        xo[i] = (unsigned)(((isx ? eb : b) * 32 + gl) * RS_RREC + gi) * 8u;
                    }
                    unsigned spins = 0;
                    for (;;) {
                        rs_load_granules<false>(gx, xr_par, xo);
                        unsigned bad = 0u;
#pragma unroll
                        for (int i = 0; i < NX; ++i) bad |= (unsigned)(gx[i] >> 32) ^ epoch;
                        if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0) break;
                        if (++spins > RS_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            failed = true;
                            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int i = 0; i < NX - 1; ++i) { const unsigned w32 = (unsigned)gx[i]; got[i] = __builtin_bit_cast(float, w32); }
#pragma unroll
                    for (int i = NX - 1; i < 4; ++i) got[i] = 0.f;
                };
                if (XG == 2) hop(std::integral_constant<int, 3>{}); else hop(std::integral_constant<int, 5>{});
                rowp = (got[0] + got[1]) + (got[2] + got[3]);                      // lanes < 16: the row sums over ALL columns
                // the extras (lane 16 + 3 b + k holds extra k of block b in got[0]): dustbin-row dual from the blocks' (sum, shift) pairs
                float smax = lane_value(got[0], 17);
                for (int b = 1; b < XG; ++b) smax = fmaxf(smax, lane_value(got[0], 17 + 3 * b));
                float tot2 = 0.f;
                for (int b = 0; b < XG; ++b) tot2 += lane_value(got[0], 16 + 3 * b) * __builtin_amdgcn_exp2f(lane_value(got[0], 17 + 3 * b) - smax);
                uM2 = rs_uniform(la_bin2 - (zr2 + smax + __builtin_amdgcn_logf(tot2)));
                vN2 = lane_value(got[0], 18);                                       // block 0's dual of the dustbin column
            }
            const float dcol2 = rs_uniform(zr2 + vN2);
            // ---- new u of my rows, the row factors f, the dustbin-column partial (lanes 0..15, one row each) ----
            float usum, dumax;
            {
                const bool live = lane < nvalid;
                const float pd = __builtin_amdgcn_exp2f(dcol2 + urv);              // the row's dustbin-column entry with the old u
                const float un = urv + la2 - __builtin_amdgcn_logf(rowp + pd);
                const float du = un - urv;
                const float f = __builtin_amdgcn_exp2f(du);
                usum = rs_wave_sum(live ? pd * f : 0.f);                            // sum_i 2^(z + v_N + u_i')
                dumax = rs_wave_max(live ? __builtin_fabsf(du) : 0.f);
                Frv = live ? Frv * f : 0.f;                                         // (rows that do not exist: F = 0, E = 0)
                urv = live ? un : urv;
            }
            RS_TP(2);
            // ---- (2) pass 2: sum_i E_ij F_i over my 16 rows ----
            f32x4 cs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cs[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < RS_RR; ++s) {
                const float f_ = lane_value(Frv, s);
                const rs_f32x2 ff = {f_, f_};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const rs_f32x2 c = rs_f32x2{cs[k][e], cs[k][e + 1]} + rs_f32x2{er[s][k][e], er[s][k][e + 1]} * ff;
                        cs[k][e] = c[0]; cs[k][e + 1] = c[1];
                    }
            }
#pragma unroll
            for (int sl = 0; sl < RS_LR; ++sl) {
                if (RS_RR + sl < nvalid) {
                    const float f_ = lane_value(Frv, RS_RR + sl);
                    const rs_f32x2 ff = {f_, f_};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(Sw + sl * RS_SEG + 256 * k);
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const rs_f32x2 c = rs_f32x2{cs[k][e], cs[k][e + 1]} + rs_f32x2{x[e], x[e + 1]} * ff;
                            cs[k][e] = c[0]; cs[k][e + 1] = c[1];
                        }
                    }
                }
            }
            RS_TP(3);
            // ---- (3) workgroup column partials: the WR tiles of a column segment through LDS in two rounds, fixed order ----
            {
                float* pb = Pbuf + (wr % PS) * NC + colb;
                if (wr < PS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(pb + 256 * k) = cs[k];
                }
                if (lane == 0) { red[16 + wave] = (wc == 0 && xg == 0) ? usum : 0.f; red[24 + 64 * (it & 1) + wave] = dumax; }     // (drift maxima: read after the last barrier of the iteration, hence two copies)
                __syncthreads();
                if (wr >= PS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(pb + 256 * k) = *reinterpret_cast<const f32x4*>(pb + 256 * k) + cs[k];
                }
                __syncthreads();
            }
            float tot[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                tot[c] = Pbuf[tid * CPT + c];
#pragma unroll
                for (int p = 1; p < PS; ++p) tot[c] += Pbuf[p * NC + tid * CPT + c];
                tot[c] *= cj[c];                           // the partial sums of the PLAN leave the workgroup: C_j is this workgroup's own (the
                                                           // workgroups of a pair evaluate E at different times, so their factors differ)
            }
            RS_TP(4);
            // ---- (4) publish my partials: row gbase + g of the parity's area, 8-byte {epoch, value} granules, relaxed stores (the vector
            //      L1 is write-through: they land in the L2).  Columns >= n publish their (zero) partials too.
            //      (5) the owner of a column slice sums the G partials of its columns and publishes the totals; everybody sweeps the totals.
            //      Batches are polled as a whole until every granule carries this epoch. ----
            float colsum[CPT];
            float colsumN = 0.f;
            auto exchange = [&](auto LOCAL_) __attribute__((always_inline)) {
                constexpr bool LOCAL = decltype(LOCAL_)::value;
                constexpr int SCOPE = LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
                const rs_u64 tag = (rs_u64)epoch << 32;
                {
                    const rs_gchar* mrow = xa_par + (int64_t)(gbase + g) * NCX * 8;
                    rs_gu64* mine = (rs_gu64*)mrow;
#pragma unroll
                    for (int c = 0; c < CPT; c += 2) rs_store_pair<LOCAL>(mrow, (unsigned)(tid * CPT + c) * 8u, tot[c], tot[c + 1], epoch);
                    if (tid == 0) {
                        float us = red[16];
#pragma unroll
                        for (int w = 1; w < RS_NW; ++w) us += red[16 + w];
                        __hip_atomic_store(mine + NC, tag | __builtin_bit_cast(unsigned, us), __ATOMIC_RELAXED, SCOPE);
                    }
                }
                RS_TP(5);
                const rs_gchar* abase = xa_par + (int64_t)(gbase + xg * Gx) * NCX * 8;   // my group's rows (wave-uniform: an SGPR pair)
                if (owner) {
                    const bool dmine = gl == 0 && tid < Gx;                           // owner 0 also sums the dustbin column
                    const unsigned offd = gl == 0 ? (unsigned)((dmine ? tid : 0) * NCX + NC) * 8u : (unsigned)(min(oslot0, Gx - 1) * NCX + ocol) * 8u;
                    float acc = 0.f;
                    for (int i0 = 0; i0 < spt && !failed; i0 += SB) {                 // batches of 2 W granules (one batch unless G is not a power of two)
                        rs_u64 gr[SB + 1];
                        unsigned go[SB + 1];
#pragma unroll
                        for (int i = 0; i < SB; ++i) go[i] = (unsigned)(min(oslot0 + (i0 + i) * TPC, Gx - 1) * NCX + ocol) * 8u;   // past Gx: any valid granule, not summed
                        go[SB] = offd;                                                // (workgroups other than 0: a granule of their own sweep once more)
                        unsigned spins = 0;
                        for (;;) {
                            rs_load_granules<LOCAL>(gr, abase, go);
                            unsigned bad = 0u;
#pragma unroll
                            for (int i = 0; i <= SB; ++i) bad |= (unsigned)(gr[i] >> 32) ^ epoch;
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
                        for (int i = 0; i < SB; ++i)
                            if (oslot0 + (i0 + i) * TPC < Gx) { const unsigned w32 = (unsigned)gr[i]; acc += __builtin_bit_cast(float, w32); }
                        if (gl == 0 && i0 == 0 && tid < 256) { const unsigned w32 = (unsigned)gr[SB]; dred[tid] = dmine ? __builtin_bit_cast(float, w32) : 0.f; }
                    }
                    osum[tid] = acc;
                }
                __syncthreads();
                if (owner) {
                    // my columns' sums over the group (threads < CW), the dustbin column's (wave 7 of the group's workgroup 0)
                    const bool colw = tid < CW, dustw = gl == 0 && wave == RS_NW - 1;
                    float t = 0.f;
                    if (colw) {
                        t = osum[tid];
                        for (int q = 1; q < TPC; ++q) t += osum[(q << cwl) + tid];
                    }
                    float td = 0.f;
                    if (dustw) {                                                      // Gx <= 256 partials: four per lane, then the fixed DPP tree
                        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dred + 4 * lane);
                        td = rs_wave_sum((d4[0] + d4[1]) + (d4[2] + d4[3]));
                    }
                    if (colw) __hip_atomic_store((rs_gu64*)xb_par + ocol, tag | __builtin_bit_cast(unsigned, t), __ATOMIC_RELAXED, SCOPE);
                    if (dustw && lane == 0) __hip_atomic_store((rs_gu64*)xb_par + NC, tag | __builtin_bit_cast(unsigned, td), __ATOMIC_RELAXED, SCOPE);
                }
                RS_TP(6);
                {
                    rs_u32x4 gq[CPT / 2];
                    rs_u64 gn;
                    unsigned spins = 0;
                    for (;;) {
                        rs_load_totals<LOCAL>(gq, gn, xb_par, (unsigned)(tid * CPT) * 8u, (unsigned)NC * 8u);
                        unsigned bad = (unsigned)(gn >> 32) ^ epoch;
#pragma unroll
                        for (int c = 0; c < CPT / 2; ++c) bad |= (gq[c][1] ^ epoch) | (gq[c][3] ^ epoch);
                        if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0) break;
                        if (++spins > RS_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            failed = true;
                            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        if (LOCAL && (spins & 63u) == 0u) asm volatile("buffer_inv sc1" ::: "memory");
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int c = 0; c < CPT / 2; ++c) {
                        // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index, hipcc 7.2)
                        const unsigned w0 = gq[c][0], w1 = gq[c][2];
                        colsum[2 * c] = __builtin_bit_cast(float, w0);
                        colsum[2 * c + 1] = __builtin_bit_cast(float, w1);
                    }
                    { const unsigned w32 = (unsigned)gn; colsumN = __builtin_bit_cast(float, w32); }
                }
            };
            if (xcd_local) exchange(std::true_type{}); else exchange(std::false_type{});
            RS_TP(7);
            // ---- (6) new v for my columns (every workgroup of the pair computes the same bits), the column factors g for the next
            //      pass 1, and -- while the new v are in registers -- the dustbin-row dual of the NEXT iteration,
            //      u_M' = log2 a_M - (z + LSE2 of the new v): per-wave (max, sum) first, one LDS round for each ----
            float dvmax = 0.f;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int j = tid * CPT + c;
                if (j < Nl) {
                    const float vo = vv[c];
                    const float vn = vo + lb2 - __builtin_amdgcn_logf(colsum[c] + __builtin_amdgcn_exp2f(zr2 + vo + uM2));
                    const float dv = vn - vo;
                    cj[c] *= __builtin_amdgcn_exp2f(dv);
                    dvmax = fmaxf(dvmax, __builtin_fabsf(dv));
                    vv[c] = vn;
                }
                X[j] = cj[c];
            }
            float vNn = OG_NEG_INF;
            if (tid == 0) {
                if (xg == 0) vNn = vN2 + lb_bin2 - __builtin_amdgcn_logf(colsumN + __builtin_amdgcn_exp2f(zr2 + vN2 + uM2));
                red[33] = xg == 0 ? vNn : vN2;               // (the other column blocks receive the new dual at the next row hop)
                red[32] = uM2;
            }
            {
                // ONE reduction round: the sum of 2^(v - shift) with the PREVIOUS iteration's maximum as the shift (a dual moves by a few
                // bits per iteration -- at most log2(m + n) + 1 -- so nothing can overflow and the sum cannot vanish), this iteration's
                // maximum for the next one, the two drift maxima and the time-out flags of the eight waves
                float mx = vNn, sv = __builtin_amdgcn_exp2f(vNn - vsh);                 // 2^-inf = 0
#pragma unroll
                for (int c = 0; c < CPT; ++c) { mx = fmaxf(mx, vv[c]); sv += __builtin_amdgcn_exp2f(vv[c] - vsh); }
                mx = rs_wave_max(mx);
                sv = rs_wave_sum(sv);
                dvmax = rs_wave_max(dvmax);
                const bool wfail = __builtin_amdgcn_ballot_w64(failed) != 0;
                if (lane == 0) { red[wave] = mx; red[8 + wave] = sv; red[40 + wave] = dvmax; red[48 + wave] = wfail ? 1.f : 0.f; }
                __syncthreads();
                const float* rk = red + 24 + 64 * (it & 1);
                mx = red[0];
                float svt = red[8], dm = red[40], um = rk[0], fl = red[48];
#pragma unroll
                for (int w = 1; w < RS_NW; ++w) { mx = fmaxf(mx, red[w]); svt += red[8 + w]; dm = fmaxf(dm, red[40 + w]); um = fmaxf(um, rk[w]); fl += red[48 + w]; }
                mx = fmaxf(mx, -1.0e30f);
                drift = rs_uniform(drift + (dm + um));
                if (XG == 1) uM2 = rs_uniform(la_bin2 - (zr2 + vsh + __builtin_amdgcn_logf(svt)));
                lse_sum = rs_uniform(svt); lse_shift = vsh;     // XG > 1: my block's part of the sum, combined at the next row hop
                vsh = rs_uniform(mx);
                vN2 = rs_uniform(red[33]);
                failed = fl != 0.f;                        // a peer never arrived: leave together (status = 1)
            }
            RS_TP(8);
            refresh = drift > RS_DRIFT_BITS;               // workgroup-uniform (every input of drift is)
            RS_TP(9);
            if (failed) break;
        }
        if (failed || it >= a.iters) break;
        // refresh: the loop above left with ++it done; E is re-evaluated from S with the current duals at the top
    }

    // ---- results in natural units: u of my rows, and (workgroup 0 of the pair) v and u_M ----
    {
    RS_THREAD_LOCALS;
    if (wc == 0 && xg == 0 && lane < nvalid) ub[row0 + lane] = urv * RS_LN2;
    if (gl == 0) {                                          // row block 0 of every column block: its slice of v
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int j = tid * CPT + c;
            if (j < Nl) a.v_out[(int64_t)bglob * a.ldv + c0 + j] = vv[c] * RS_LN2;
        }
        if (tid == 0 && xg == 0) { a.v_out[(int64_t)bglob * a.ldv + N] = vN2 * RS_LN2; ub[M] = red[32] * RS_LN2; }
    }
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

// geometry of one pair: W = wave tiles per row of a workgroup tile (tile = 128 / W rows x 1024 W columns; 0 = not resident-capable),
// X column blocks of Gx row blocks each (G = X Gx workgroups).  One column block per XCD: Gx <= 32.
struct RsGeom { int W, G, Gx, X; };
RsGeom rs_geom(int m, int n, int rw = RS_RW) {
    RsGeom q{0, 0, 0, 1};
    if (m <= 0 || n <= 0 || n > 4 * RS_SEG) return q;
    auto rows = [&](int W) { const int RB = rw * RS_NW / W; int g = (m + RB - 1) / RB; return g < 2 * W ? 2 * W : g; };   // an owner sums <= 512 columns
    // the widest tile whose row blocks still fit one XCD; narrower tiles = more column blocks = a row hop per iteration
    if (n <= RS_SEG) { q.W = 1; q.X = 1; }
    else if (n <= 2 * RS_SEG) { if (rows(2) <= 32) { q.W = 2; q.X = 1; } else { q.W = 1; q.X = 2; } }
    else { if (rows(4) <= 32) { q.W = 4; q.X = 1; } else if (rows(2) <= 32) { q.W = 2; q.X = 2; } else { q.W = 1; q.X = 4; } }
    q.Gx = rows(q.W);
    if (q.Gx > 32) { q.W = 0; return q; }                  // more than 4096 rows: streaming kernels
    q.G = q.X * q.Gx;
    return q;
}
// uniform batch: pairs per launch with `cus` CUs (0 = never)
int rs_pairs_per_round(const RsGeom& q, int cus) {
    if (q.W == 0 || cus < 8) return 0;
    if (cus >= 256) return (8 / q.X) * (32 / q.Gx);        // the one-group-per-XCD map
    return q.X == 1 ? cus / q.G : 0;
}

size_t rs_area_bytes(int W, int slots, int groups) {    // status + xcc table, then the exchange areas (column partials, column totals, row records)
    const size_t NCX = (size_t)RS_SEG * W + RS_PAD;
    return 256 + (size_t)RS_MAXWG * sizeof(unsigned) + (size_t)2 * slots * NCX * 8 + (size_t)2 * groups * NCX * 8 + (size_t)2 * groups * 32 * RS_RREC * 8;
}

template <int W, class MAP>
void rs_launch(const SkResArgs& a, const MAP& map, int grid, hipStream_t st) {
    hipLaunchKernelGGL((sinkhorn_resident_kernel<W, MAP>), dim3(grid), dim3(512), 0, st, a, map);
}

// Few pairs (round 5): 4 rows per wave instead of 16 when the 16-row geometry would leave three quarters of the CUs idle and the pair fits one
// XCD as 32 x 1024 tiles (n <= 1024, m <= 1024).  OG_SINKHORN_FEW=0 / 1 forces (1: whenever the geometry exists).  -> rows per wave (16 or 4)
int rs_rows_per_wave(int B, int m, int n, int cus) {
    if (cus < 256 || n > RS_SEG) return RS_RW;
    const RsGeom q16 = rs_geom(m, n);
    if (q16.W == 0) return RS_RW;
    const char* e = getenv("OG_SINKHORN_FEW");                       // read per call: the tests switch it; 0 = never, 4 / 8 = that geometry when it exists, 1 = the finest
    const int want = e ? atoi(e) : -1;
    if (want == 0) return RS_RW;
    for (int rw = 4; rw <= 8; rw *= 2) {                             // the finest geometry that still fits ONE launch (and, by default, fills <= all CUs from <= half of them)
        if (want > 1 && want != rw) continue;
        const RsGeom q = rs_geom(m, n, rw);
        if (q.W != 1 || q.X != 1 || (int64_t)B * q.G > 256 || B > rs_pairs_per_round(q, cus)) continue;      // ONE launch: with the per-XCD map a round holds 8 (32 / Gx) pairs -- 8 once Gx > 16 (9-12 pairs of 513-900 rows took two launches of the 4-row geometry)
        if (want < 0 && (int64_t)B * rs_geom(m, n, 2 * rw).G > 128) continue;      // the next coarser geometry already uses more than half the chip: stay there
        return rw;
    }
    return RS_RW;
}

}  // namespace

bool og_sinkhorn_resident_shape_ok(int B, int m, int n) {
    return B > 0 && rs_geom(m, n).W != 0;
}

size_t og_sinkhorn_resident_ws_bytes(int B, int m, int n) {
    if (!og_sinkhorn_resident_shape_ok(B, m, n)) return 0;
    // A ragged batch launches every width class with ITS OWN W, and a pair narrower in rows can take a wider tile than the maxima
    // (m_max, n_max) would (2100 x 900 next to 1000 x 2100: the maxima give W = 1, the second pair W = 4): the slot is sized for the widest
    // class any pair of n_b <= n can fall into, not for rs_geom(m, n).W.
    const int Wmax = n <= RS_SEG ? 1 : n <= 2 * RS_SEG ? 2 : 4;
    return rs_area_bytes(Wmax, RS_MAXWG, RS_MAXPAIRS);
}

// The tile geometry of one pair on a part with 8 XCDs x 32 CUs (pure host arithmetic, no device needed: CPU tests): out = {W, X, Gx,
// pairs per launch}; returns 0, or OG_E_SHAPE when the shape has no resident geometry.
extern "C" int og_sinkhorn_resident_geometry(int32_t m, int32_t n, int32_t* out4) {
    if (!out4) return OG_E_INVALID;
    const RsGeom q = rs_geom(m, n);
    if (q.W == 0) return OG_E_SHAPE;
    out4[0] = q.W; out4[1] = q.X; out4[2] = q.Gx; out4[3] = rs_pairs_per_round(q, 256);
    return 0;
}

// Host arithmetic only (8 XCDs x 32 CUs assumed; CPU tests): the row slots per wave a uniform launch of B pairs of m x n keypoints takes -- 16, or the
// few-pairs geometries 4 / 8 (OG_SINKHORN_FEW overrides as at run time); 0 when the shape has no resident geometry.
extern "C" int og_sinkhorn_resident_rows_per_wave(int32_t batch, int32_t m, int32_t n) {
    if (batch <= 0 || rs_geom(m, n).W == 0) return 0;
    return rs_rows_per_wave(batch, m, n, 256);
}

// launches the resident kernel would need for this uniform batch on this device (0 = not possible)
int og_sinkhorn_resident_rounds(int B, int m, int n) {
    if (!og_sinkhorn_resident_shape_ok(B, m, n)) return 0;
    const int ppr = rs_pairs_per_round(rs_geom(m, n, rs_rows_per_wave(B, m, n, rs_num_cus())), rs_num_cus());
    return ppr > 0 ? (B + ppr - 1) / ppr : 0;
}

// mode: 1 = when the launches are co-resident AND the problem is large enough to pay off, 2 = whenever possible (tests)
bool og_sinkhorn_resident_wanted(int B, int m, int n, int mode) {
    if (mode <= 0) return false;
    const int rounds = og_sinkhorn_resident_rounds(B, m, n);
    if (rounds <= 0) return false;
    if (mode >= 2) return true;
    return (int64_t)B * m * n >= (int64_t)1 << 18;         // below that the streaming kernels are launch-bound either way and need no co-residency
}

int og_launch_sinkhorn_resident(const float* S, int64_t lds, const float* zdev, float zhost, int B, int m, int n, int iters,
                                float inv_reg, float la, float la_bin, float lb, float lb_bin, float* u, int ldu, const float* v_in,
                                float* v_out, int ldv, void* xws, unsigned* status, hipStream_t st, bool trusted_padding) {
    if (!S || !u || !v_in || !v_out || !xws || !status || iters < 1 || !og_sinkhorn_resident_shape_ok(B, m, n)) return OG_E_INVALID;
    const int rw = rs_rows_per_wave(B, m, n, rs_num_cus());
    const RsGeom q = rs_geom(m, n, rw);
    const int ppr = rs_pairs_per_round(q, rs_num_cus());
    if (ppr <= 0) return OG_E_SHAPE;
    const int rounds = (B + ppr - 1) / ppr, per = (B + rounds - 1) / rounds;      // balanced rounds
    hipError_t e = hipSuccess;                                                     // (the status word is the caller's, zeroed with its duals: sticky over the rounds)
    SkResArgs a{};
    a.S = S; a.lds = lds; a.strideS = (int64_t)m * lds;
    a.u = u; a.ldu = ldu; a.v_in = v_in; a.v_out = v_out; a.ldv = ldv;
    a.status = status;
    a.xcc = (unsigned*)((char*)xws + 256);
    a.xa = (char*)xws + 256 + RS_MAXWG * sizeof(unsigned);
    { const char* ev = getenv("OG_SINKHORN_AGENT_SCOPE"); a.force_agent_scope = ev && atoi(ev) != 0; }       // read per call: the tests switch it
    a.zdev = zdev; a.zhost = zhost; a.inv_reg = inv_reg; a.la = la; a.la_bin = la_bin; a.lb = lb; a.lb_bin = lb_bin;
    a.m = m; a.n = n; a.iters = iters;
    a.sanitize_pad = (n & 3) && !trusted_padding;
    const size_t NCX = (size_t)RS_SEG * q.W + RS_PAD;
    const bool xcd_map = rs_num_cus() >= 256;
    for (int b0 = 0; b0 < B; b0 += per) {
        const int np = B - b0 < per ? B - b0 : per;
        a.b0 = b0; a.npairs = np; a.slots = np * q.G; a.groups = np * q.X;
        a.xb = a.xa + (size_t)2 * a.slots * NCX * 8;
        a.xc = a.xb + (size_t)2 * a.groups * NCX * 8;
        // epochs start at 1: every tag (and every XCC entry) must read 0 first
        e = hipMemsetAsync(a.xcc, 0, RS_MAXWG * sizeof(unsigned) + (size_t)2 * a.slots * NCX * 8 + (size_t)2 * a.groups * NCX * 8 + (size_t)2 * a.groups * 32 * RS_RREC * 8, st);
        if (e != hipSuccess) return (int)e;
        RsUniform map{q.G, q.Gx, q.X, 0};
        int grid = np * q.G;
        a.local_ok = 0;
        if (xcd_map) {
            const int ppl = 8 / q.X;                       // pairs per layer
            map.layers = (np + ppl - 1) / ppl; grid = 8 * map.layers * q.Gx; a.local_ok = 1;
        }
        if (rw == 4) hipLaunchKernelGGL((sinkhorn_resident_kernel<1, RsUniform, 4>), dim3(grid), dim3(512), 0, st, a, map);
        else if (rw == 8) hipLaunchKernelGGL((sinkhorn_resident_kernel<1, RsUniform, 8>), dim3(grid), dim3(512), 0, st, a, map);
        else if (q.W == 1) rs_launch<1>(a, map, grid, st);
        else if (q.W == 2) rs_launch<2>(a, map, grid, st);
        else rs_launch<4>(a, map, grid, st);
    }
    return og_launch_status();
}

// ---- ragged batches: every pair gets the geometry of its own size; pairs of one width class W are packed into launches in which
//      each pair sits on ONE XCD (first fit, largest first: 8 XCDs x 32 slots), the classes run one after the other ----
namespace {
struct RsPlanPair { int b, W, G; };
// false: some pair has no resident geometry (or needs more than one XCD): the caller streams the whole batch
bool rs_ragged_plan(const RaggedDesc& rd, RsPlanPair* pp, int cus) {
    if (rd.B <= 0 || rd.B > OG_MAX_RAGGED || cus < 256) return false;
    for (int b = 0; b < rd.B; ++b) {
        const RsGeom q = rs_geom(rd.off0[b + 1] - rd.off0[b], rd.off1[b + 1] - rd.off1[b]);
        if (q.W == 0 || q.X != 1) return false;
        pp[b] = RsPlanPair{b, q.W, q.G};
    }
    return true;
}
// bytes of the exchange slot (from its first byte, the status word) one ragged launch of width class W touches
size_t rs_ragged_launch_bytes(int W, int slots, int groups) {
    const size_t NCX = (size_t)RS_SEG * W + RS_PAD;
    return 256 + (size_t)RS_MAXWG * sizeof(unsigned) + (size_t)2 * slots * NCX * 8 + (size_t)2 * groups * NCX * 8;   // (ragged pairs: one column block, no row records)
}
// The launches of a ragged batch, pure host arithmetic: f(W, map, np, slots, qmax) once per launch, widest class first.
template <class F>
void rs_ragged_for_each_launch(const RaggedDesc& rd, RsPlanPair* pp, F&& f) {
    // largest pairs first (stable: ties keep the batch order)
    for (int i = 1; i < rd.B; ++i) { const RsPlanPair t = pp[i]; int j = i; while (j > 0 && pp[j - 1].G < t.G) { pp[j] = pp[j - 1]; --j; } pp[j] = t; }
    bool done[OG_MAX_RAGGED] = {};
    for (int W = 4; W >= 1; W /= 2) {                                               // widest class first: its launches take narrower pairs along
        for (;;) {                                                                  // one launch per pass
            RsRagged map;
            for (int i = 0; i < RS_MAXWG; ++i) map.wg[i] = -1;
            int used[8] = {0, 0, 0, 0, 0, 0, 0, 0}, np = 0, slots = 0, qmax = 0;
            auto place = [&](int i, int G) {
                int x = -1;
                for (int k = 0; k < 8; ++k) if (used[k] + G <= 32 && (x < 0 || used[k] < used[x])) x = k;      // the emptiest XCD that still fits
                if (x < 0) return;
                const int b = pp[i].b;
                map.gb[np] = b; map.G[np] = G; map.gbase[np] = slots;
                map.m[np] = rd.off0[b + 1] - rd.off0[b]; map.n[np] = rd.off1[b + 1] - rd.off1[b];
                for (int g = 0; g < G; ++g) map.wg[x + 8 * (used[x] + g)] = (np << 8) | g;
                used[x] += G; if (used[x] > qmax) qmax = used[x];
                slots += G; ++np; done[i] = true;
            };
            for (int i = 0; i < rd.B; ++i)
                if (!done[i] && pp[i].W == W) place(i, pp[i].G);
            if (np > 0) {
                // free XCD slots of this launch take pairs of the NARROWER classes along, cut into this launch's tiles (a 128 x 1024 pair as
                // 64 x 2048 tiles leaves half of every wave row empty, but a launch costs ~0.5 ms whatever it holds: the 16 pairs of the C5
                // draw take 2 launches instead of 3)
                const int RB = RS_RW * RS_NW / W;
                for (int i = 0; i < rd.B; ++i) {
                    if (done[i] || pp[i].W >= W) continue;
                    const int b = pp[i].b, mrows = rd.off0[b + 1] - rd.off0[b];
                    int G = (mrows + RB - 1) / RB;
                    if (G < 2 * W) G = 2 * W;
                    if (G <= 32) place(i, G);
                }
            }
            if (np == 0) break;
            f(W, map, np, slots, qmax);
        }
    }
}
}  // namespace

bool og_sinkhorn_resident_ragged_wanted(const RaggedDesc& rd, int mode) {
    if (mode <= 0) return false;
    RsPlanPair pp[OG_MAX_RAGGED];
    if (!rs_ragged_plan(rd, pp, rs_num_cus())) return false;
    if (mode >= 2) return true;
    int64_t elems = 0;
    for (int b = 0; b < rd.B; ++b) elems += (int64_t)(rd.off0[b + 1] - rd.off0[b]) * (rd.off1[b + 1] - rd.off1[b]);
    return elems >= (int64_t)1 << 18;
}

// Host arithmetic only (a part with 8 XCDs x 32 CUs assumed; CPU tests): the launches a ragged batch of per-pair sizes would take and the
// bytes of the resident exchange slot the widest of them touches; out2 = {launches, bytes}.  OG_E_SHAPE: some pair has no one-XCD geometry
// (the batch streams).  The slot og_sinkhorn_workspace_bytes(batch, max m, max n) reserves must hold `bytes`.
extern "C" int og_sinkhorn_resident_ragged_footprint(int32_t batch, const int32_t* lens0, const int32_t* lens1, int64_t* out2) {
    if (!lens0 || !lens1 || !out2 || batch <= 0 || batch > OG_MAX_RAGGED) return OG_E_INVALID;
    RaggedDesc rd{};
    rd.B = batch;
    for (int b = 0; b < batch; ++b) {
        if (lens0[b] <= 0 || lens1[b] <= 0) return OG_E_SHAPE;
        rd.off0[b + 1] = rd.off0[b] + lens0[b]; rd.off1[b + 1] = rd.off1[b] + lens1[b];
    }
    RsPlanPair pp[OG_MAX_RAGGED];
    if (!rs_ragged_plan(rd, pp, 256)) return OG_E_SHAPE;
    int64_t launches = 0, bytes = 0;
    rs_ragged_for_each_launch(rd, pp, [&](int W, const RsRagged&, int np, int slots, int) {
        ++launches;
        const int64_t by = (int64_t)rs_ragged_launch_bytes(W, slots, np);
        if (by > bytes) bytes = by;
    });
    out2[0] = launches; out2[1] = bytes;
    return 0;
}

int og_launch_sinkhorn_resident_ragged(const float* S, int64_t lds, const float* zdev, float zhost, const RaggedDesc& rd, int m_max, int n_max, int iters,
                                       float inv_reg, float* u, int ldu, const float* v_in, float* v_out, int ldv, void* xws, unsigned* status,
                                       hipStream_t st, bool trusted_padding, int* count_only) {
    if (!count_only && (!S || !u || !v_in || !v_out || !xws || !status || iters < 1)) return OG_E_INVALID;
    RsPlanPair pp[OG_MAX_RAGGED];
    if (!rs_ragged_plan(rd, pp, rs_num_cus())) return OG_E_SHAPE;
    if (count_only) *count_only = 0;
    const size_t ws_bytes = og_sinkhorn_resident_ws_bytes(rd.B, m_max, n_max);      // what the caller's slot holds
    SkResArgs a{};
    a.S = S; a.lds = lds; a.strideS = (int64_t)m_max * lds;
    a.u = u; a.ldu = ldu; a.v_in = v_in; a.v_out = v_out; a.ldv = ldv;
    a.status = status;
    a.xcc = (unsigned*)((char*)xws + 256);
    a.xa = (char*)xws + 256 + RS_MAXWG * sizeof(unsigned);
    { const char* ev = getenv("OG_SINKHORN_AGENT_SCOPE"); a.force_agent_scope = ev && atoi(ev) != 0; }
    a.zdev = zdev; a.zhost = zhost; a.inv_reg = inv_reg;
    a.m = m_max; a.n = n_max; a.iters = iters; a.local_ok = 1;
    (void)trusted_padding;
    a.sanitize_pad = 1;     // columns [n_b, lds) of a ragged pair's rows were never written by anybody: they may hold NaN / Inf
    int rc = 0;
    rs_ragged_for_each_launch(rd, pp, [&](int W, const RsRagged& map, int np, int slots, int qmax) {
        if (count_only) { ++*count_only; return; }
        if (rc) return;
        if (rs_ragged_launch_bytes(W, slots, np) > ws_bytes) { rc = OG_E_SHAPE; return; }      // never past the slot (cannot happen: the slot is sized for W = 4 whenever n_max allows it)
        const size_t NCX = (size_t)RS_SEG * W + RS_PAD;
        a.b0 = 0; a.npairs = np; a.slots = slots; a.groups = np;
        a.xb = a.xa + (size_t)2 * a.slots * NCX * 8;
        a.xc = a.xb + (size_t)2 * a.groups * NCX * 8;
        const hipError_t e = hipMemsetAsync(a.xcc, 0, rs_ragged_launch_bytes(W, slots, np) - 256, st);
        if (e != hipSuccess) { rc = (int)e; return; }
        const int grid = 8 * qmax;
        if (W == 1) rs_launch<1>(a, map, grid, st);
        else if (W == 2) rs_launch<2>(a, map, grid, st);
        else rs_launch<4>(a, map, grid, st);
    });
    if (rc) return rc;
    return count_only ? 0 : og_launch_status();
}

// Shared device/host helpers for the gfx950 SuperGlue kernels.  Internal (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/openglue_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define OG_WAVE 64
#define OG_NEG_INF (-__builtin_huge_valf())

// Row of a 32x32 MFMA accumulator element: lane holds column (lane & 31); register r of the 16
// holds row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)   (cdna_hip_programming.md §3, C/D layout,
// dtype-independent on gfx950).
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hipGetLastError() is sticky per thread and also reports errors left behind by OTHER users of the
// runtime in this process (e.g. a benign probe inside torch).  Every ABI entry clears it first, so a
// non-zero status after our launches is ours.
static inline void og_clear_status() { (void)hipGetLastError(); }

static inline int og_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline int64_t og_round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Split-f16 operands: x = hi + lo with hi = f16(x), lo = f16(x - hi), both IEEE binary16 (|x| < 65504).  lo is kept
// at its TRUE scale (about 2^-11 |x|), so the three MFMA passes  Ah·Bh + Ah·Bl + Al·Bh  accumulate into ONE fp32
// accumulator.  For |x| < 2^-3 lo is an f16 subnormal; v_mfma_f32_32x32x16_f16 honours subnormal inputs
// (scripts/probes/mfma_denorm.hip, profiles/r01_probe_mfma_denorm.log), so the representation error is
// <= max(2^-22 |x|, 2^-25).  Weights are additionally pre-scaled by OG_W_SCALE at pack time (exact) so that their
// lo parts are normal numbers; the GEMM epilogue multiplies the accumulator by 1/OG_W_SCALE.  og_pack_weights lowers the pre-scale
// per weight matrix (a power of two, og_weight_prescale) when 256 |w| would leave binary16 -- e.g. a BatchNorm fold over a dead
// channel -- and stores 1 / S next to the matrix; the kernels read it from there (GemmHArgs::scale_dev).
#define OG_W_SCALE 256.0
// largest power of two S <= 256 with S * maxabs <= 32768 (half the binary16 range: room for rounding), at least 2^-14 (S itself is
// used as a binary16 constant by mlp_fused.hip: weights up to 5e8; beyond that og_pack_weights fails with OG_E_RANGE)
static inline double og_weight_prescale(double maxabs) {
    double S = OG_W_SCALE;
    while (S * maxabs > 32768.0 && S > 6.103515625e-05) S *= 0.5;
    return S;
}
// CALLERS: if x is the result of a multiply, pin it first (`asm("" : "+v"(x))`, or compute it under `#pragma clang fp
// contract(off)` and check the ISA for v_fma_mix).  The compiler may otherwise fuse that multiply into ONE of the two
// conversions (v_fma_mixlo_f16: single rounding) while the other uses the separately rounded fp32 product
// (v_cvt_pk_f16_f32); the two his can differ by an ulp and hi + lo is then off by 2^-11 |x| (seen on the attention
// output; tests/test_gpu_parity.py::test_attention_vs_oracle catches it).
__device__ __forceinline__ void og_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Four fp32 values -> two packed (hi, lo) f16 pairs, 3 instructions per pair: hi = RNE pack; lo = f16(x - hi) by the
// mixed-precision FMA (f32 x * 1.0 - f16 hi, one rounding) written straight into the low / high half of the result;
// bit-identical to og_split (scripts/probes/split_asm.hip).  ONE asm block with a fixed internal order, because the
// compiler's hazard recognizer does not look inside inline asm: gfx950 needs a wait state between a transcendental op
// (e.g. the v_exp_f32 that produced x) and a VALU reading its result, and between an op_sel partial register write
// (mixlo) and the next access of that register (mixhi) -- hence the leading s_nop and the A/B interleave.
// ha = (hi(x0), hi(x1)), la = (lo(x0), lo(x1)), hb / lb likewise for x2, x3.
__device__ __forceinline__ void og_split4(float x0, float x1, float x2, float x3, unsigned& ha, unsigned& la, unsigned& hb, unsigned& lb) {
    asm("s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %2, %6, %7\n\t"
        "v_fma_mixlo_f16 %1, %4, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, 1.0, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(ha), "=&v"(la), "=&v"(hb), "=&v"(lb)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}

// The same four values for the attention kernel's block-scaled P V cross products (attention.hip, MX form): the f16 hi pairs as above, plus
// h8 = four e4m3 bytes of x0..x3 (the B operand of P8h . V8lo) and l8 = four e4m3 bytes of (x - hi) / rscale (B operand of P8l . V8hi; rscale = 2^-11:
// the residual is brought to the scale of x, the MFMA's block scale takes the 2^-11 back).  x - hi is exact in fp32 (one v_fma_mix_f32 per value).
// Byte e of h8 / l8 belongs to x_e: the k order of the 8-bit MFMA operand.  5 instructions per pair against og_split4's 3.
__device__ __forceinline__ void og_split4_mx(float x0, float x1, float x2, float x3, float rscale, unsigned& ha, unsigned& hb, unsigned& h8, unsigned& l8) {
    float t0, t1, t2, t3;
    asm("s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
        "v_cvt_pk_fp8_f32 %2, %8, %9\n\t"
        "v_fma_mix_f32 %4, %8, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %9, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %6, %10, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_fp8_f32 %2, %10, %11 op_sel:[0,0,1]\n\t"
        "v_fma_mix_f32 %7, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_scalef32_pk_fp8_f32 %3, %4, %5, %12\n\t"
        "s_nop 0\n\t"
        "v_cvt_scalef32_pk_fp8_f32 %3, %6, %7, %12 op_sel:[0,0,0,1]"
        : "=&v"(ha), "=&v"(hb), "=&v"(h8), "=&v"(l8), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(rscale));
}

// "hl32" operand format of the split-f16 GEMM (gemm_f16x3.hip): the hi and lo halves of a row live in ONE
// row of 2K halves, interleaved in groups of 32 channels -- [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63 | ...]
// -- so that the 32-channel k-slab a GEMM stage consumes is one full 128-byte cache line (64 B hi + 64 B lo).
// Column c of the hi part sits at og_hl_col(c), its lo partner 32 halves further.  Vector accesses of 4 or 8
// halves at aligned columns never straddle a group.
__host__ __device__ __forceinline__ int64_t og_hl_col(int c) { return ((int64_t)(c >> 5) << 6) + (c & 31); }

// Ragged batches: every pair b has its own (m_b, n_b).  Tokens are PACKED (no padding): image-0 sets at rows
// off0[b] .. off0[b+1], image-1 sets at rows T0 + off1[b] ..; the descriptor travels to the kernels by value
// (kernarg), so no device-side table and no host->device copy is needed.  B == 0 means "uniform batch".
struct RaggedDesc {                    // OG_MAX_RAGGED: include/openglue_amd.h
    int B;
    int off0[OG_MAX_RAGGED + 1];       // prefix sums of m_b
    int off1[OG_MAX_RAGGED + 1];       // prefix sums of n_b
    int64_t soff[OG_MAX_RAGGED + 1];   // float offsets of the [m_b+1][n_b+1] blocks in the packed scores output
};

// Uniform batches pass this empty stand-in instead of the 1.3 KB descriptor (kernels are templates on the descriptor type).
// [Measured with the per-block entry stamps of scripts/trace_gemm.py: with the full descriptor in the kernarg segment the
// 256 workgroups of a GEMM launch entered over 12-14 us; with a small kernarg segment they all enter within 0.3 us.]
struct RaggedNone {
    static constexpr int B = 0;
    static constexpr int off0[2] = {0, 0};
    static constexpr int off1[2] = {0, 0};
    static constexpr int64_t soff[2] = {0, 0};
};

// ---- internal launchers shared between api.hip and the per-stage entry points ----
struct GemmArgs {
    const float* A; int64_t lda, strideA;
    const float* B; int64_t ldb, strideB;
    float* C; int64_t ldc, strideC;
    int M, N, K, batch;
    const float* bias;        // [N] or null
    int relu;                 // activation: 0 none, 1 ReLU, 2 sin(30 v)
    const float* res; int64_t ldr, strideR;   // residual / mix source, rows like C
    const float* alpha;       // [N] or null: v = alpha*v + (1-alpha)*res
    float scale;
    float* Ct; int64_t ldct, strideCt; int ct_rows;   // optional transposed copy: Ct[row / ct_rows][col][row % ct_rows]
    _Float16* Ch; _Float16* Cl; int64_t ldch;         // optional split-f16 copy (hi, lo), batch == 1 only
    int c_hl;                                         //   1: hl32 row format (Cl == Ch + 32, ldch = row stride), 0: two planes
    const RaggedDesc* rag;                            // host pointer or null: batched problem z = pair z of a ragged batch
                                                      // (A rows off0[z].., B rows T0 + off1[z].., M = m_z, N = n_z)
    int ta, tb;                                       // operand stored K-MAJOR: A[k][m] (row stride lda) / B[k][n]; forms: 00, 01, 11
    int a_colsum;                                     // ta: column N of C (ldc > N) receives sum_k A[k][m] * scale (the bias gradient of a conv)
    int ktot;                                         // ta && tb, split-K: problem z contracts k-rows [z K, min((z+1) K, ktot)); 0 = off
};
int og_launch_gemm(const GemmArgs& a, hipStream_t stream);

// split-f16 GEMM (gemm_f16x3.hip): tokens A [M][K] and weights B [N][K], both in the hl32 row format
struct GemmHArgs {
    const _Float16* A; int64_t lda;           // row stride in halves (>= 2K)
    const _Float16* B; int64_t ldb;
    int M, N, K;
    float scale;                              // v = acc * scale + bias (undoes a power-of-two pre-scale of B)
    float inv_scale;                          // set by the launcher: 1 / scale (the bias enters the accumulators as bias / scale)
    const float* scale_dev;                   // optional DEVICE scalar that overrides `scale` (read by the kernel): the per-matrix
                                              // power-of-two pre-scale og_pack_weights chose for this weight matrix
    const float* bias; int relu;
    const float* res; int64_t ldr;            // fp32 residual [M][N] (may alias C32), or
    const _Float16* res_hl; int64_t ldrh;     //   split-f16 residual in hl32 rows (may alias Ch when c_hl): v += hi + lo
    float* C32; int64_t ldc;                  // optional fp32 output
    _Float16* Ch; _Float16* Cl; int64_t ldch; // optional split-f16 output: two planes, or
    int c_hl;                                 //   1: hl32 rows (Cl == Ch + 32, ldch = row stride in halves)
    const float* alpha;                       // [N] or null: with res, v = alpha*v + (1-alpha)*res instead of v + res
    float* Ct; int64_t ldct; int ct_rows;     // optional channel-first fp32 copy: Ct[row / ct_rows][col][row % ct_rows] (ldct = ct_rows)
    int ct_rag;                               //   ragged batch (with `rag`, batch <= 1): 1 = rows are image-0 tokens, 2 = image-1 tokens;
                                              //   pair b's [N][rows_b] block starts at Ct + N * off[b] (packed, no padding)
    int batch;                                // 0/1: one problem; > 1: problems z = blockIdx.y with the strides below
    int64_t strideA, strideB, strideC32;      // elements (halves / floats) between consecutive problems
    const RaggedDesc* rag;                    // host pointer or null: batched problem z = pair z of a ragged batch
                                              // (A rows off0[z].., B rows T0 + off1[z].., M = m_z, N = n_z; C32 at z*strideC32)
    int split_row, split_n;                   // split_row > 0: rows < split_row produce only columns < split_n (two row ranges with different
                                              // column counts in ONE launch: the cross layer's q of image 0 + q | k | v of image 1).  Only the
                                              // 256-tile kernel honours it: og_gemm_f16x3_row_split_ok() tells whether a launch qualifies.
};
int og_launch_gemm_f16x3(const GemmHArgs& a, hipStream_t stream);
bool og_gemm_f16x3_row_split_ok(const GemmHArgs& a);
int og_launch_split_f16(const float* x, int64_t n, void* hi, void* lo, hipStream_t stream);
int og_launch_split_f16_hl(const float* x, int64_t rows, int cols, int64_t ldx, void* out, int64_t ldo, hipStream_t stream);
int og_launch_merge_f16_hl(const void* in, int64_t rows, int cols, int64_t ldi, float* out, int64_t ldo, hipStream_t stream);

// mlp_fused.hip: the message MLP of a GNN layer (fc.0 -> ReLU -> fc.3 + residual) in one launch, the hidden activation in registers
struct MlpFusedArgs {
    _Float16* XO; int64_t ld;     // [M] hl32 rows of [x | O] (4D halves used, row stride ld halves); x is updated in place
    int M;
    const char* wstream;          // og_pack_mlp_stream: fragment-major (hi, lo) halves of 256 W0', 256 W3'
    const float* b0;              // [2D] folded fc.0 bias
    const float* b3;              // [D] folded fc.3 bias
    float scale;                  // 1 / OG_W_SCALE: accumulator multiplier of both matrices, unless ...
    const float* scales_dev;      // ... DEVICE [2] = {1 / S0, 1 / S3}: the per-matrix pre-scales og_pack_weights chose (null: `scale`)
};
bool og_mlp_fused_supported(int D);
bool og_mlp_fused_enabled(int D);                    // supported and not switched off (OG_MLP_FUSED=0)
size_t og_mlp_stream_bytes(int D);
bool og_pack_mlp_stream(int D, const double* W0, const double* W3, void* out, double S0 = OG_W_SCALE, double S3 = OG_W_SCALE);   // false: a weight does not fit binary16
int og_launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t stream);      // picks mlp_small_kernel (32-token workgroups) when og_mlp_small_wanted(M)
bool og_mlp_small_wanted(int M);
// the q / k / v projections for few token rows (mlp_fused.hip: proj_small_kernel), a fragment-major copy of the packed matrix
size_t og_proj_stream_bytes(int N, int K);
bool og_pack_proj_stream(int N, int K, const double* W, void* out, double S);
// ... and for batches (mlp_fused.hip: proj_stream_kernel, 128-token workgroups, x fragments in registers, the weights through an LDS ring)
size_t og_proj_stream_big_bytes(int N, int K);
bool og_pack_proj_stream_big(int N, int K, const double* W, void* out, double S);
bool og_proj_stream_wanted(int M, int K, bool full);      // og_forward's choice: K = 128 batches (OG_PROJ_STREAM forces)
int og_launch_proj_stream(const _Float16* X, int64_t ld, int M, int K, const char* wstream, const float* bias, const float* scale_dev,
                          _Float16* Ch, _Float16* Cl, int64_t ldc, int split_row, int a0, int a1, int b0, int b1, hipStream_t stream);   // ranges in units of 128 channels
int og_launch_proj_small(const _Float16* X, int64_t ld, int M, int K, const char* wstream, const float* bias, const float* scale_dev,
                         _Float16* Ch, _Float16* Cl, int64_t ldc, int split_row, int a0, int a1, int b0, int b1, hipStream_t stream);

constexpr int OG_ATTN_COUNTERS = 256;                                   // (problem, head, query tile) triples of a key-split launch
constexpr int64_t OG_ATTN_PARTIAL_FLOATS = (int64_t)512 * 4 * 34 * 64;     // 512 workgroups (two per CU) x 4 waves x (32 O registers + m + l) x 64 lanes
struct AttnArgs {
    const _Float16* qh; const _Float16* ql; int64_t ldq;     // leading dimensions in halves
    const _Float16* kh; const _Float16* kl; int64_t ldk;
    const _Float16* vh; const _Float16* vl; int64_t ldv;
    _Float16* oh; _Float16* ol; int64_t ldo;
    int o_hl;                 // 1: output rows in the hl32 format (ol == oh + 32), 0: two planes
    // problem z in [0, nz): rows of q/out start at q_row0(z), rows of k/v at kv_row0(z)
    int nz, num_heads, dh;
    int split;                // problems z < split use geometry A, the others geometry B
    int64_t q_base[2], q_step[2], kv_base[2], kv_step[2];
    int nq[2], nk[2];
    int qtiles;               // set by the launcher: query tiles per (problem, head)
    const RaggedDesc* rag;    // host pointer or null; with rag_mode: 1 = self (z < B image 0, else image 1),
    int rag_mode;             // 2 = cross, queries of image 0 attend image 1, 3 = cross, image 1 attends image 0
    int feat;                 // og_launch_favor_attention only: random features per head (columns of q and k; v / out have dh columns)
    float* lse;               // optional (uniform single-geometry calls): [nz][num_heads][nq] row log-sum-exp of the scaled scores, natural units
    float* partial;           // optional scratch of OG_ATTN_PARTIAL_FLOATS floats + int* counters of OG_ATTN_COUNTERS zeroed ints: with both set, launches of
    int* counters;            // very few workgroups split the KEY range of a query tile over 2 or 4 workgroups (attention.hip: GS)
    int gs_scatter;           // set by the launcher (test knob OG_ATTN_GS_SCATTER): key parts of a query tile on consecutive workgroups
    int mx;                   // 1: `vl` holds the 8-bit rows [e4m3(V 2^-sv) | e4m3((V - Vh) 2^(11 - sv))] instead of the f16 lo parts: the P V cross products run on the
    int mx_scale_a;           //    block-scaled 8-bit MFMA (attention.hip, MX); mx_scale_a = the E8M0 byte 127 + sv - 11 replicated (set by the launcher from mx_sv)
    int mx_sv;                //    power-of-two exponent the producer divided V by before the e4m3 conversion
};
int og_launch_attention(const AttnArgs& a, hipStream_t stream);
int og_launch_linear_attention(const AttnArgs& a, hipStream_t stream);   // attention = 'linear' (elu+1 feature map)
int og_launch_favor_attention(const AttnArgs& a, hipStream_t stream);    // attention = 'favor_relu': q, k hold relu(P x d^-1/4) (a.feat columns)

// m, n are the (maximum) sizes; with `rag` pair b uses m_b, n_b, S stays at stride m*lds per pair, scores are packed
// row_best (optional): the kernel that writes the scores also leaves max / argmax_j<n of every row i<m there (the first half of the
// mutual-NN extraction, matches.hip), so that the extraction need not read the rows again
struct RowBest { int* idx; float* val; int stride; };      // [batch][stride]
int og_launch_sinkhorn(const float* S, int64_t lds, const float* dustbin_dev /*or null*/, float dustbin_host, int batch, int m, int n, int iters,
                       float reg, float* scores, void* workspace, hipStream_t stream, const RaggedDesc* rag = nullptr,
                       const RowBest* row_best = nullptr, bool trusted_padding = true);   // false: columns [n, lds) of S may hold anything
// sinkhorn_resident.hip: the dual-stabilised iterations with the plan matrices resident in registers + LDS (one launch per round of
// co-resident pairs: n <= 4096, any batch)
bool og_sinkhorn_resident_shape_ok(int B, int m, int n);
size_t og_sinkhorn_resident_ws_bytes(int B, int m, int n);            // exchange granules + status word (0: shape never resident)
bool og_sinkhorn_resident_wanted(int B, int m, int n, int mode);      // mode 1: resident-capable and large enough, 2: resident-capable
int og_sinkhorn_resident_rounds(int B, int m, int n);                 // launches of the resident kernel for this uniform batch on this device (0: none)
// ragged batches: every pair gets the geometry of its own (m_b, n_b); pairs are packed into launches of at most one pair-group per XCD slot range
bool og_sinkhorn_resident_ragged_wanted(const RaggedDesc& rd, int mode);
int og_launch_sinkhorn_resident_ragged(const float* S, int64_t lds, const float* zdev, float zhost, const RaggedDesc& rd, int m_max, int n_max, int iters,
                                       float inv_reg, float* u, int ldu, const float* v_in, float* v_out, int ldv, void* xws, unsigned* status,
                                       hipStream_t st, bool trusted_padding = true, int* count_only = nullptr);    // count_only: no work, *count_only = launches it would take
static inline bool og_sinkhorn_resident_ragged_wanted(const RaggedNone&, int) { return false; }
int og_launch_sinkhorn_resident(const float* S, int64_t lds, const float* zdev, float zhost, int B, int m, int n, int iters,
                                float inv_reg, float la, float la_bin, float lb, float lb_bin, float* u, int ldu, const float* v_in,
                                float* v_out, int ldv, void* xws, unsigned* status /* zeroed by the caller */, hipStream_t st, bool trusted_padding = true);
int og_launch_matches(const float* scores, int batch, int m, int n, float thr, int64_t* matches0,
                      float* ms0, int64_t* matches1, float* ms1, void* workspace, hipStream_t stream,
                      const RaggedDesc* rag = nullptr, bool rows_done = false);      // rows_done: og_matches_row_best() was filled already
RowBest og_matches_row_best(void* matches_workspace, int batch, int m, int n);
// per-pair image sizes of a ragged batch (keypoint normalisation, superglue.py:74-78): pair b owns tokens off[b] .. off[b+1]
struct EncoderRagged {
    int B;                              // 0: uniform batch, one image size for every token
    int off[OG_MAX_RAGGED + 1];
    float wm1[OG_MAX_RAGGED], hm1[OG_MAX_RAGGED];   // W - 1, H - 1
};
int og_launch_encoder_input(const float* kpts, const float* side, int64_t tokens, int s, float wx, float wy,
                            float* out /*[tokens][32]*/, hipStream_t stream, const EncoderRagged* er = nullptr);

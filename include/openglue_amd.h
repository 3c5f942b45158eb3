/*
 * openglue_amd.h -- C ABI of the MI355X (gfx950) SuperGlue hot path.
 *
 * Drop-in boundary for ucuapps/OpenGlue's keypoint-graph matcher.  The reference has no FFI or
 * operator registry: its plugin point is the Python class
 *     models/superglue/superglue.py:11   class SuperGlue(nn.Module)
 *     models/superglue/superglue.py:29   def forward(self, data) -> dict
 * constructed at models/matching_module.py:44 and inference.py:74, and the match extraction that
 * consumes its `scores` at models/matching_module.py:174-187 / inference.py:176-190.
 * openglue_amd/superglue.py keeps that Python signature and calls the functions below through
 * ctypes; any other host (C, C++, a cgo/JNI stub) can bind the same symbols.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (the stream is passed as void*,
 *     i.e. a hipStream_t; NULL = the default stream).
 *   - the library never allocates or frees device memory and keeps no global state: inputs,
 *     packed weights, workspace and outputs are caller-owned device buffers (fp32 unless noted).
 *   - every function only ENQUEUES work on `stream` (no internal synchronisation), so it composes
 *     with the caller's stream/allocator semantics and can be captured into a hipGraph.
 *   - return value: 0 = success; negative = OG_E_* invalid argument; positive = hipError_t of a
 *     failed launch.  Nothing throws across the ABI.
 *   - layouts are token-major: a set of n keypoints with C channels is an [n][C] row-major matrix.
 */
#ifndef OPENGLUE_AMD_H
#define OPENGLUE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_ABI_VERSION 10

#define OG_E_INVALID   (-1)  /* NULL pointer / non-positive size                         */
#define OG_E_SHAPE     (-2)  /* unsupported shape (see og_check_shape)                   */
#define OG_E_ALIGN     (-3)  /* a device pointer is not 16-byte aligned                  */
#define OG_E_FLAG      (-4)  /* unknown flag bits                                        */
#define OG_E_RANGE     (-5)  /* og_pack_weights: a folded weight is not finite (or beyond 2^40).  Large finite weights -- e.g. a BatchNorm
                                fold over a dead channel with running_var ~ 0 -- are packed with a smaller per-matrix pre-scale   */

/* config flags (reference keys: superglue.py:18-19 `residual`, `no_descriptors`;
 * attention_gnn.py:51-52 `use_offset`) */
#define OG_FLAG_RESIDUAL        1
#define OG_FLAG_USE_OFFSET      2
#define OG_FLAG_NO_DESCRIPTORS  4
#define OG_FLAG_LINEAR_ATTENTION 16  /* attention_gnn.attention = 'linear': elu+1 linear attention (attention.py:22-40) instead of softmax */
#define OG_FLAG_FAVOR_RELU      32  /* attention_gnn.attention = 'favor_relu': generalised FAVOR+ attention with ReLU random features
                                       (attention.py:43-95, __init__.py:19-25): phi(x) = relu(P x d^-1/4) + 1e-8 on q and k, P the
                                       [2D][D] buffer og_layer_params.favor_projection, then linear attention over the 2D features.
                                       As in the reference this needs num_heads == 1 (its matmul of P with the per-head tensors only
                                       type-checks when the head size equals D); D <= 256.  Not combinable with LINEAR_ATTENTION. */
#define OG_FLAG_SIREN_ENCODER   8   /* positional_encoding.encoder_name = FeedForwardNetSiren: [Conv, sin(30x)]*h + Conv,
                                       no BatchNorm (models/utils.py:23-45); og_params.enc_bn is ignored */

#define OG_MAX_HIDDEN 8

/* Everything SuperGlue.__init__ reads from its config (superglue.py:16-27, 64, 108-109) plus the
 * batch geometry of one call.  Within one call every pair has m keypoints in image 0 and n in
 * image 1 (m != n allowed), as the reference requires. */
typedef struct og_shape {
    int32_t batch;                  /* B image pairs                                             */
    int32_t m, n;                   /* keypoints per image 0 / image 1                           */
    int32_t desc_dim;               /* D = descriptor_dim = attention_gnn.embed_dim (mult. of 64) */
    int32_t num_heads;              /* H; head h owns channels h*D/H .. (h+1)*D/H-1 (attention_gnn.py:24-26); D/H in {16,32,64,128} (128, round 6: register-staged attention kernel; training backward GEMM by GEMM)
                                       (OG_FLAG_FAVOR_RELU: H == 1, any D <= 256) */
    int32_t num_stages;             /* L self+cross stages (attention_gnn.py:84-89)              */
    int32_t side_info;              /* s = positional_encoding.side_info_size (2+s <= 32)        */
    int32_t num_hidden;             /* len(positional_encoding.hidden_layers_sizes) <= OG_MAX_HIDDEN */
    int32_t hidden[OG_MAX_HIDDEN];  /* e.g. 32, 64, 128 (config/config.yaml:45)                  */
    int32_t sinkhorn_iters;         /* otp.num_iters                                             */
    float   sinkhorn_reg;           /* otp.reg                                                   */
    int32_t flags;                  /* OG_FLAG_*                                                 */
    float   match_threshold;        /* inference.match_threshold (config/config.yaml:40)         */
} og_shape;

/* ---- host-side parameter views (fp32, HOST memory), names as in the reference state-dict ---- */
typedef struct og_conv {            /* nn.Conv1d(kernel_size=1): weight [out][in] row-major, bias [out] */
    const float* weight;
    const float* bias;
} og_conv;

typedef struct og_bn {              /* nn.BatchNorm1d, eval mode (running statistics), eps = 1e-5 */
    const float* weight;
    const float* bias;
    const float* running_mean;
    const float* running_var;
} og_bn;

typedef struct og_layer_params {    /* attention_gnn.layers.{l}.module.*  (attention_gnn.py:35-41, 9-20) */
    og_conv in_proj_q, in_proj_k, in_proj_v, out_proj;   /* mha.*  [D][D]          */
    og_conv fc0;                                         /* fc.0   [2D][2D]        */
    og_bn   fc_bn;                                       /* fc.2   BatchNorm1d(2D) */
    og_conv fc3;                                         /* fc.3   [D][2D]         */
    const float* favor_projection;                       /* mha.attention_func.projection_matrix [2D][D] (attention.py:53); read with
                                                            OG_FLAG_FAVOR_RELU only, may be NULL otherwise (ABI v4)                  */
} og_layer_params;

typedef struct og_params {
    og_conv enc_conv[OG_MAX_HIDDEN + 1];  /* positional_encoding.encoder.{0,3,6,..}            */
    og_bn   enc_bn[OG_MAX_HIDDEN];        /* positional_encoding.encoder.{2,5,8,..}            */
    const og_layer_params* layers;        /* 2*L entries: even = self, odd = cross             */
    og_conv linear_proj;                  /* linear_proj [D][D]                  (superglue.py:22) */
    const float* mix_coefs;               /* [D] (state-dict [D,1]); may be NULL without OG_FLAG_RESIDUAL (superglue.py:21) */
    float dustbin_score;                  /* dustbin_score                       (superglue.py:23) */
} og_params;

/* ---- device-side call arguments ---- */
typedef struct og_inputs {
    const float* keypoints0;          /* [B][m][2] pixel xy                   data['keypoints0']         */
    const float* keypoints1;          /* [B][n][2]                                                       */
    const float* descriptors0;        /* [B][m][D]                            data['local_descriptors0'] */
    const float* descriptors1;        /* [B][n][D]                                                       */
    const float* side_info0;          /* [B][m][s]                            data['side_info0']         */
    const float* side_info1;          /* [B][n][s]                                                       */
    float image0_wh[2];               /* (W, H) of image 0   (superglue.py:35-38, 74-78)                 */
    float image1_wh[2];
} og_inputs;

typedef struct og_outputs {
    float*   scores;                  /* [B][m+1][n+1] log-assignment incl. dustbins   'scores'          */
    float*   context_descriptors0;    /* [B][D][m] channel-first, as the reference returns them; may be NULL */
    float*   context_descriptors1;    /* [B][D][n]; may be NULL                                          */
    int64_t* matches0;                /* [B][m]  index into image 1 or -1 (matching_module.py:181); may be NULL */
    float*   matching_scores0;        /* [B][m]  exp(max) if mutual else 0;  NULL iff matches0 is NULL   */
    int64_t* matches1;                /* [B][n]  inference.py:188; may be NULL                           */
    float*   matching_scores1;        /* [B][n]; NULL iff matches1 is NULL                               */
} og_outputs;

int og_abi_version(void);

/* 0 if the shape is supported by this build, else OG_E_SHAPE / OG_E_FLAG. */
int og_check_shape(const og_shape* shape);

/* Size in bytes of the packed-weight blob / of the per-call workspace for `shape` (0 on error).
 * The packed size does not depend on batch, m, n. */
size_t og_packed_weights_bytes(const og_shape* shape);
size_t og_workspace_bytes(const og_shape* shape);

/* Host-side, one-off (re-run when parameters change): fold eval-mode BatchNorm into the following
 * conv (the reference order is Conv -> ReLU -> BN, models/utils.py:52-56), fold out_proj into fc.0,
 * concatenate and pre-scale the q/k/v projections, zero-pad the encoder MLP to MFMA tile sizes.
 * `packed_host` (og_packed_weights_bytes bytes, host memory) is then copied to the device by the
 * caller.  Arithmetic in double precision. */
int og_pack_weights(const og_shape* shape, const og_params* params, void* packed_host);

/* Introspection of the packed blob (offsets in floats).  All matrices are [out][in] row-major; the
 * GNN matrices are stored as split-f16 hl32 rows of 256 * w (see og_split_f16_hl / og_gemm_nt_f16x3 below:
 * [out][2*in] halves = out*in floats), everything else as fp32.
 *   enc_w[i] [enc_out[i]][enc_k[i]], enc_b[i] [enc_out[i]]   keypoint-encoder conv i, zero-padded (k: 32 | out: mult. of 64),
 *                                                            BatchNorm i-1 folded in
 *   layer l at layer0 + l*layer_stride:  wqkv [3D][D] (q rows pre-scaled by (D/H)^-1/2 * log2 e), bqkv [3D],
 *                                        w0 [2D][2D] = [W0a | Wm*Wo], b0 [2D], w3 [D][2D] (BN folded), b3 [D]
 *   wp [D][D] (hl32 rows of 256 * w like the GNN matrices), bp [D], alpha [D] = sigmoid(mix_coefs), dustbin [1] */
typedef struct og_packed_layout_t {
    int32_t n_enc;
    int32_t enc_k[OG_MAX_HIDDEN + 1], enc_out[OG_MAX_HIDDEN + 1];
    int64_t enc_w[OG_MAX_HIDDEN + 1], enc_b[OG_MAX_HIDDEN + 1];
    int64_t layer0, layer_stride, o_wqkv, o_bqkv, o_w0, o_b0, o_w3, o_b3;   /* o_w*: hl32 rows of 2K halves */
    int64_t wp, bp, alpha, dustbin, total;
    int64_t o_scale;  /* ABI v5: per layer, 4 floats {1 / S_qkv, 1 / S_0, 1 / S_3, 0}: every split-f16 matrix is stored as (hi, lo) halves of
                         S * w with its own power-of-two S = 256 unless 256 * max|w| would leave binary16 (a BatchNorm fold over a dead
                         channel, a huge gamma / sigma); the kernels multiply the accumulator by the stored 1 / S                    */
    int64_t scales;   /* ABI v5: 2 floats {1 / S_wp, 1 / S of the last encoder conv}                                                 */
    int64_t o_wmlp;   /* ABI v5: per layer, the SAME folded w0 / w3 once more as the fragment-major stream og_mlp_block consumes
                         (og_mlp_block_stream_bytes(D) bytes; -1 when D has no fused message-MLP kernel)                         */
    int64_t o_wqkvs;  /* ABI v7: per layer, the q | k | v matrix once more as the fragment-major stream og_proj_block consumes (the
                         small-batch projection kernel; N K 4 bytes; -1: D not 256 / 128, or favor_relu)   */
    int64_t o_wqkvb;  /* ABI v8: ... and a third time as the stream of the BATCH projection kernel (proj_stream_kernel: launches of more than
                         8192 token rows; N K 4 bytes; -1 as above).  og_proj_block_stream_bytes(N, K) = both streams, the small one first  */
} og_packed_layout_t;
int og_packed_layout(const og_shape* shape, og_packed_layout_t* layout);

/* The whole hot path: keypoint encoder -> L x (self, cross) attention -> final projection ->
 * score matrix -> log-domain Sinkhorn with dustbins -> scores (+ optional mutual-NN matches).
 * Replaces SuperGlue.forward (superglue.py:29-72) and, when outputs->matches0 != NULL, the match
 * extraction of matching_module.py:174-187 / inference.py:176-190. */
int og_forward(const og_shape* shape, const og_inputs* in, const void* packed_dev,
               void* workspace_dev, const og_outputs* out, void* stream);

/* og_forward plus a copy of the RESIDUAL STREAM at one stage boundary, for per-stage parity tests (the reference's sub-modules:
 * positional_encoding.py:16-19 + superglue.py:52-55 for tap 0; attention_gnn.py:57-77, one DescriptorsSelfAttention /
 * DescriptorsCrossAttention = ResidualAttentionMessagePropagation on both images, for tap k >= 1):
 *   tap = 0:        x = local_descriptors + keypoint_encoder(...) as it enters the GNN;
 *   tap = k in 1..2L: x after GNN layer k - 1 (even layers self, odd layers cross);
 * tap_x: [B*m + B*n][D] fp32, token-major, image-0 sets first (what the kernels hold as (hi, lo) f16 pairs, merged).
 * Everything else as og_forward (the call runs the whole path). */
int og_forward_tap(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                   const og_outputs* out, void* stream, int32_t tap, float* tap_x);

/* Per-stage entries SURVEY.md 8(b) names beside og_sinkhorn / og_attention / og_mlp_block / og_extract_matches:
 * og_keypoint_encoder -- superglue.py:44-55 with positional_encoding.py:16-19 and the normalisation of :74-78: x = local_descriptors +
 *   encoder(normalised keypoints, side info) for all B*m + B*n tokens (image-0 sets first), fp32 [T][D] -- exactly what og_forward_tap(tap = 0)
 *   copies out, without running the rest of the path (outputs: none but x_out; workspace and packed weights as og_forward).
 * og_scores -- superglue.py:64, 80-86: S[b] = g0[b] g1[b]^T * D^-1/2 for token-major fp32 descriptors g0 [B][m][D], g1 [B][n][D] into
 *   S [B][m][lds] (lds >= n, multiple of 4), exact fp32 MFMA (og_gemm_nt).  og_forward forms the same product from the (hi, lo) rows its
 *   final projection leaves, on the split-f16 kernel. */
int og_keypoint_encoder(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev, float* x_out, void* stream);
int og_scores(const float* g0, const float* g1, int32_t batch, int32_t m, int32_t n, int32_t D, float* S, int64_t lds, void* stream);

/* Ragged batch (BASELINE config 5): pair b has lens0[b] keypoints in image 0 and lens1[b] in image 1
 * (host arrays, batch <= OG_MAX_RAGGED; shape->m / shape->n are the maxima).  Every tensor is PACKED without
 * padding in pair order: keypoints0 [sum m_b][2], descriptors0 [sum m_b][D], ..., scores = the
 * [m_b+1][n_b+1] blocks one after the other, matches0 / matching_scores0 [sum m_b], matches1 [sum n_b],
 * context_descriptors0 = the channel-first [D][m_b] blocks one after the other (may be NULL), likewise 1.
 * image0_wh / image1_wh: host arrays [batch][2] = (W, H) of every pair's images (keypoint normalisation,
 * superglue.py:35-41, 74-78); NULL = in->image{0,1}_wh for all pairs.
 * The result of pair b equals og_forward on that pair alone (the reference has no masks: SURVEY.md 3.5). */
#define OG_MAX_RAGGED 64
int og_forward_ragged(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh,
                      const float* image1_wh, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                      const og_outputs* out, void* stream);

/* Profiling variant of og_forward (bench.py): same work, but every launch group is bracketed by HIP
 * events recorded on `stream`; the call SYNCHRONISES the stream and returns the summed elapsed
 * milliseconds and the number of bracketed launch groups per kernel class.  GEMM and attention are
 * bracketed per kernel launch, Sinkhorn / matches per stage (many small launches: og_sinkhorn_schedule tells which). */
#define OG_STAGE_ENCODER_INPUT 0
#define OG_STAGE_GEMM          1
#define OG_STAGE_ATTENTION     2
#define OG_STAGE_SINKHORN      3
#define OG_STAGE_MATCHES       4
#define OG_STAGE_GEMM_F16X3    5   /* split-f16 GEMMs (q/k/v projections, final projection, score matrix, last encoder conv; the message
                                      MLP as two launches when it is not fused); OG_STAGE_GEMM = the exact-fp32 ones */
#define OG_STAGE_MLP_FUSED     6   /* ABI v5: the fused message-MLP kernel (csrc/mlp_fused.hip), one launch per bracket */
#define OG_NUM_STAGES          7
int og_forward_profiled(const og_shape* shape, const og_inputs* in, const void* packed_dev,
                        void* workspace_dev, const og_outputs* out, void* stream,
                        float* stage_ms /*[OG_NUM_STAGES]*/, int32_t* stage_launches /*[OG_NUM_STAGES]*/);
int og_forward_ragged_profiled(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh,
                               const float* image1_wh, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                               const og_outputs* out, void* stream, float* stage_ms, int32_t* stage_launches);

/* ---- per-stage entry points (unit parity tests; also usable on their own) ---- */

/* C[z] = epilogue(A[z] * B[z]^T): A [M][K] (lda), B [N][K] (ldb), exact fp32 MFMA.
 * v = acc + bias[col]; relu; if res: alpha ? alpha[col]*v + (1-alpha[col])*res : v + res; v *= scale. */
int og_gemm_nt(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB,
               float* C, int64_t ldc, int64_t strideC, int32_t M, int32_t N, int32_t K, int32_t batch,
               const float* bias, int32_t relu, const float* res, int64_t ldr, const float* alpha,
               float scale, void* stream);

/* The same kernel with K-MAJOR operands (exact fp32 MFMA, no epilogue but `scale`): B[z] is stored [K][N] (row stride ldb) and, with
 * a_kmajor, A[z] is stored [K][M] -- the layouts the backward products of a 1x1 conv and of attention come in (dX = dZ W with
 * W [out][in]; dW = dZ^T X with both operands [tokens][channels]), so that no transposed copy of a token-sized tensor is made.
 * k_total > 0 (a_kmajor only) = split-K: problem z contracts k-rows [z K, min((z+1) K, k_total)) -- strideA = K*lda, strideB = K*ldb
 * cut one long contraction into `batch` partial products C[z] the caller sums.  a_colsum (a_kmajor only, ldc > N): column N of C[z]
 * receives sum_k A[k][m] -- the bias gradient of the conv comes out of the weight-gradient launch.  lda, ldb multiples of 4;
 * K % 4 == 0 only when A is K-contiguous (a_kmajor == 0). */
int og_gemm_kmajor(const float* A, int64_t lda, int64_t strideA, int32_t a_kmajor, const float* B, int64_t ldb, int64_t strideB,
                   float* C, int64_t ldc, int64_t strideC, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t k_total,
                   int32_t a_colsum, float scale, void* stream);

/* Split-f16 representation used inside the GNN: x = hi + lo with hi = f16(x), lo = f16(x - hi), both IEEE
 * binary16 (|x| < 65504; lo may be subnormal, the matrix cores honour it), stored either as two planes or in the "hl32" row format the GEMM consumes: one row of
 * 2K halves per token, hi and lo interleaved in groups of 32 channels
 * [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63 | ...] so that a 32-channel k-slab is one 128-byte line.
 * og_split_f16 converts n (multiple of 4) fp32 values into two planes; og_split_f16_hl converts
 * x [rows][cols] (row stride ldx, cols % 32 == 0) into hl32 rows (row stride ldo >= 2*cols halves). */
int og_split_f16(const float* x, int64_t n, void* hi, void* lo, void* stream);
int og_split_f16_hl(const float* x, int64_t rows, int32_t cols, int64_t ldx, void* out, int64_t ldo, void* stream);

/* C = epilogue(A * B^T) with A [M][K], B [N][K] in the hl32 row format (lda, ldb: row strides in halves,
 * multiples of 8, >= 2K; K % 32 == 0): 3 f16 MFMAs per product, fp32 accumulate, fp32-class accuracy.
 * v = acc * scale + bias[col] (scale undoes a power-of-two pre-scale of B: og_pack_weights stores 256 * W so
 * that the lo parts of small weights stay normal numbers); relu; + res[row][col] (fp32, ldr); written as fp32 (C32, may be NULL) and/or in
 * split-f16 form: c_hl == 0 -> two planes Ch/Cl with leading dimension ldch; c_hl != 0 -> hl32 rows at Ch
 * (row stride ldch halves, Cl ignored, N % 32 == 0).  N % 4 == 0. */
int og_gemm_nt_f16x3(const void* A, int64_t lda, const void* B, int64_t ldb, int32_t M, int32_t N, int32_t K,
                     float scale, const float* bias, int32_t relu, const float* res, int64_t ldr, float* C32, int64_t ldc,
                     void* Ch, void* Cl, int64_t ldch, int32_t c_hl, void* stream);
/* The same with the residual given as hl32 rows (res_hl, row stride ldrh halves; may alias Ch when c_hl != 0: every element is read
 * before it is written by the same lane): v += hi + lo.  This is the form of the GNN's fc.3 (attention_gnn.py:55: desc_q + fc(message)):
 * og_forward keeps the residual stream as (hi, lo) rows.  C32 may be NULL when a split-f16 output is given. */
int og_gemm_nt_f16x3_reshl(const void* A, int64_t lda, const void* B, int64_t ldb, int32_t M, int32_t N, int32_t K,
                           float scale, const float* bias, int32_t relu, const void* res_hl, int64_t ldrh, float* C32, int64_t ldc,
                           void* Ch, void* Cl, int64_t ldch, int32_t c_hl, void* stream);

/* The message MLP of one GNN layer (attention_gnn.py:53-55 with models/utils.py:48-58, BatchNorm and out_proj folded as
 * og_pack_weights does) as ONE launch, in place on M token rows of hl32 rows [x | O] (4D halves used per row, row stride ld halves):
 *     x <- x + W3 relu(W0 [x ; O] + b0) + b3,     W0 [2D][2D], W3 [D][2D] row-major fp32, b0 [2D], b3 [D] (device).
 * The hidden activation lives in registers (csrc/mlp_fused.hip); the weights are consumed as a fragment-major stream of (hi, lo)
 * halves of 256 w that og_mlp_block_pack writes on the host (og_mlp_block_stream_bytes(D) bytes, 0 = D not supported: D == 256, and since
 * ABI v8 D == 128, the reference's SIFT / HardNet descriptor width).
 * Same arithmetic as og_gemm_nt_f16x3 (relu) followed by og_gemm_nt_f16x3_reshl. */
/* ABI v7.  A 1x1 conv of the residual stream for FEW token rows (the q / k / v projections of attention_gnn.py:43-47 when one or a few
 * image pairs are matched: og_forward uses it for launches of <= 8192 rows): y[M][N] = x W^T + bias on the x halves of M hl32 rows
 * (the first 2K halves of a row, row stride ld halves), W [N][K] row-major fp32, K = 256 or (ABI v8) 128, N a multiple of 32; the result leaves as
 * (hi, lo) f16 planes yh / yl [M][ldy].  32-token workgroups whose 8 waves split the output blocks (csrc/mlp_fused.hip:
 * proj_small_kernel); the weights are consumed as a fragment-major stream of (hi, lo) halves of 256 w that og_proj_block_pack writes on
 * the host (og_proj_block_stream_bytes(N, K) bytes, 0 = shape not supported).  Rows below split_row get the output columns
 * [32 a0, 32 a1), the others [32 b0, 32 b1) (split_row a multiple of 32, or 0 / >= M for one range).  inv_scale_dev: DEVICE float,
 * 1 / 256 for streams packed here.  Same arithmetic as og_gemm_nt_f16x3.  ABI v8: og_proj_block takes K; launches of more than 8192 rows whose column
 * ranges are whole groups of 128 channels (and split_row a multiple of 128, yh / yl 128-byte aligned) take the batch kernel (proj_stream_kernel:
 * 128-token workgroups, the x fragments in registers, weights through an LDS ring) -- og_proj_block_pack writes its stream behind the small-batch
 * one.  OG_PROJ_STREAM=0 / 1 forces either kernel.  ABI v10: N (the row count of W the stream was packed for) is an argument -- it locates the batch
 * stream behind the small-batch one; before, it was read off ldy, which silently mis-addressed the stream for an output plane padded beyond N.
 * ldy >= N, a multiple of 64 halves for the batch kernel. */
size_t og_proj_block_stream_bytes(int32_t N, int32_t K);
int og_proj_block_pack(int32_t N, int32_t K, const float* W, void* stream_host);
int og_proj_block(const void* x_rows, int64_t ld, int32_t M, int32_t K, int32_t N, const void* stream_dev, const float* bias, const float* inv_scale_dev,
                  void* yh, void* yl, int64_t ldy, int32_t split_row, int32_t a0, int32_t a1, int32_t b0, int32_t b1, void* stream);

size_t og_mlp_block_stream_bytes(int32_t D);
int og_mlp_block_pack(int32_t D, const float* W0, const float* W3, void* stream_host);
int og_mlp_block(int32_t D, void* xo_rows, int64_t ld, int32_t M, const void* stream_dev, const float* b0, const float* b3,
                 void* stream);

/* softmax attention (attention.py:8-19) for `batch` independent problems and H heads, operands and
 * result as split-f16 planes: q [batch][nq][ldq] (columns h*dh.. of row i = head h, PRE-SCALED by
 * dh^-0.5 * log2(e): the kernel evaluates softmax as 2^(q.k - max)), k, v [batch][nk][ld*],
 * out [batch][nq][ldo]; leading dimensions in elements.  dh in {16,32,64,128}.
 * ABI v6: lse (may be NULL) [batch][num_heads][nq] receives the row log-sum-exp of the scaled scores in natural units,
 * ln sum_j exp(dh^-0.5 q_i . k_j) -- what a backward pass that recomputes the attention matrix needs (og_attention_backward). */
int og_attention(const void* qh, const void* ql, int64_t ldq, const void* kh, const void* kl, int64_t ldk,
                 const void* vh, const void* vl, int64_t ldv, void* oh, void* ol, int64_t ldo, int32_t batch,
                 int32_t nq, int32_t nk, int32_t num_heads, int32_t dh, float* lse, void* stream);

/* log-domain Sinkhorn with implicit dustbins (superglue.py:88-111 + optimal_transport.py:20-28):
 * S [B][m][lds] (lds % 4 == 0) is the raw score matrix, `dustbin` the learnt bin score; writes
 * scores [B][m+1][n+1] = S_aug/reg + u + v - norm.  workspace: og_sinkhorn_workspace_bytes. */
size_t og_sinkhorn_workspace_bytes(int32_t batch, int32_t m, int32_t n);
int og_sinkhorn(const float* S, int64_t lds, float dustbin, int32_t batch, int32_t m, int32_t n,
                int32_t iters, float reg, float* scores, void* workspace_dev, void* stream);

/* Diagnostics: state of the last og_sinkhorn / og_forward Sinkhorn stage that ran on this workspace.  Copies two words on a private stream
 * after the work ALREADY ENQUEUED ON THE NULL STREAM has finished (an event, no device-wide synchronisation: other streams keep running and
 * are not waited for -- a caller that launched on its own stream synchronises that stream first, as openglue_amd/superglue.py does).
 * State of the last og_sinkhorn / og_forward Sinkhorn stage that ran on
 * this Sinkhorn workspace (every call resets it).
 *   0 = completed normally (streaming kernels, or the on-chip-resident kernel without incident);
 *   2 = a cross-workgroup wait of the on-chip-resident iteration kernel timed out (the workgroups of a launch must be
 *       co-resident; the launcher checks that against the CU count, but another stream or process holding CUs can still break it)
 *       and the safety-net kernel enqueued behind it solved the problem again, one workgroup per pair: the scores are VALID, the
 *       call was slow (milliseconds);
 *   3 = a NON-FINITE value was written to the scores: something upstream overflowed or was NaN -- an activation beyond the binary16
 *       range of the split-f16 operands (|x| >= 65504), non-finite inputs.  The scores are invalid;
 *   1 = timed out and not recomputed (cannot happen with this build's launch sequence; scores invalid); -1 = bad arguments.
 * For n <= 4096 and at least 2^18 matrix entries per call (OG_SINKHORN_RESIDENT=0 disables, =2 drops the size threshold) iterations
 * 2..iters run with the plan matrices held in registers + LDS, one launch per round of co-resident pairs (csrc/sinkhorn_resident.hip).
 * OG_SINKHORN_FORCE_TIMEOUT=1 (tests) makes those launches behave as if they had timed out. */
int og_sinkhorn_status(const void* sinkhorn_workspace_dev, int32_t batch, int32_t m, int32_t n);
/* Which schedule og_sinkhorn / og_forward take for a UNIFORM batch of this shape in this process (environment switches and the
 * device's CU count included): k >= 1 = first iteration streaming, then k launches of the on-chip-resident kernel for iterations
 * 2..iters (one per round of co-resident pairs: 32 pairs of 1024 x 1024, 8 of 2048 x 2048, 2 of 4096 x 4096 on 256 CUs) + the
 * safety-net and scores kernels; 0 = streaming (2 launches per iteration + 1).  bench.py uses it to name what its Sinkhorn bracket
 * timed.  og_sinkhorn_schedule_ragged: the same for a ragged batch (host arrays of per-pair sizes, batch <= OG_MAX_RAGGED): pairs are
 * grouped by width class and packed one pair per XCD slot range; 0 when any pair has no resident geometry. */
int og_sinkhorn_schedule(int32_t batch, int32_t m, int32_t n, int32_t iters);
int og_sinkhorn_schedule_ragged(int32_t batch, const int32_t* lens0, const int32_t* lens1, int32_t iters);
/* How the on-chip-resident schedule tiles ONE pair of m x n keypoints on a part with 8 XCDs x 32 CUs (host arithmetic only, no device):
 * out4 = {W, X, Gx, pairs per launch} -- workgroup tiles of (128 / W) rows x (1024 W) columns, X column blocks (one per XCD) x Gx row blocks;
 * returns 0, or OG_E_SHAPE when the shape has no resident geometry (more than 4096 rows or columns: streaming kernels). */
int og_sinkhorn_resident_geometry(int32_t m, int32_t n, int32_t* out4);
/* Host arithmetic only (8 XCDs x 32 CUs assumed): the row slots per WAVE a uniform launch of `batch` pairs takes in the resident kernel: 16 (workgroup
 * tiles of 128 / W rows), or -- launches of few pairs of <= 1024 x 1024 keypoints, the reference's one-pair-per-call regime (inference.py:214-235) --
 * 4 (a pair = 32 tiles of 32 x 1024: up to 8 pairs) or 8 (16 tiles of 64 x 1024: 9 to 16 pairs); 0 = no resident geometry.  OG_SINKHORN_FEW=0 / 1 / 4 / 8
 * overrides. */
int og_sinkhorn_resident_rows_per_wave(int32_t batch, int32_t m, int32_t n);
/* Host arithmetic only (8 XCDs x 32 CUs assumed): the resident launches a RAGGED batch of per-pair sizes takes and the bytes of the
 * resident exchange slot the widest of them touches, out2 = {launches, bytes}; OG_E_SHAPE when some pair has no one-XCD geometry (the
 * batch streams).  The slot og_sinkhorn_workspace_bytes(batch, max m, max n) reserves is sized for the widest tile class any pair with
 * n_b <= max n can take, so `bytes` always fits it (tests/test_boundary_cpu.py checks that over random plans). */
int og_sinkhorn_resident_ragged_footprint(int32_t batch, const int32_t* lens0, const int32_t* lens1, int64_t* out2);
/* The same check for the workspace of an og_forward / og_forward_ragged call with this shape (for ragged calls: the shape
 * that was passed, i.e. the maxima).  Waits like og_sinkhorn_status (NULL stream only).  Return values as og_sinkhorn_status. */
int og_forward_status(const og_shape* shape, const void* workspace_dev);

/* ---- training slice of the optimal-transport layer (reference: autograd through superglue.py:88-111 +
 * optimal_transport.py:20-28, consumed by the NLL of utils/losses.py:7-53) ----
 * og_sinkhorn_train_forward: like og_sinkhorn (max-subtracted iterations only) but keeps the duals of every iteration in
 * the training workspace; og_sinkhorn_backward: given grad_scores = dL/dscores [B][m+1][n+1] it back-propagates through the
 * `iters` unrolled iterations and writes dS [B][m][ldds] (gradient w.r.t. the raw score matrix) and *d_dustbin (device
 * scalar, may be NULL; gradient w.r.t. dustbin_score).  n <= 4159, iters >= 1.  fp32 atomics: gradients reproducible to
 * rounding, not bit for bit.  The same workspace must be passed to both calls.  ABI v6: dustbin_dev (device scalar, or NULL =
 * use the host value `dustbin`) -- the learnable dustbin_score is read ON the device, the training step never waits for the GPU. */
size_t og_sinkhorn_train_workspace_bytes(int32_t batch, int32_t m, int32_t n, int32_t iters);
int og_sinkhorn_train_forward(const float* S, int64_t lds, float dustbin, const float* dustbin_dev, int32_t batch, int32_t m, int32_t n,
                              int32_t iters, float reg, float* scores, void* train_workspace_dev, void* stream);
int og_sinkhorn_backward(const float* S, int64_t lds, float dustbin, const float* dustbin_dev, int32_t batch, int32_t m, int32_t n, int32_t iters,
                         float reg, const float* grad_scores, void* train_workspace_dev, float* dS, int64_t ldds,
                         float* d_dustbin, void* stream);

/* ---- training slice: train-mode BatchNorm of the MLPs (reference models/utils.py:48-58, FeedForwardNet = Conv1d -> ReLU ->
 * nn.BatchNorm1d, in training mode: torch.nn.functional.batch_norm(training=True)) on TOKEN-MAJOR activations
 * x [rows][channels] (row stride ldx floats; the reference's [B, C, N] tensor with rows = B*N): batch statistics per channel
 * over all rows, y = (x - mean) / sqrt(biased var + eps) * weight + bias, running_mean / running_var updated in place with
 * `momentum` (running_var takes the unbiased variance), all like torch.  weight / bias / running_* / save_* may be NULL;
 * save_mean / save_invstd [channels] are what a backward pass needs.  channels % 4 == 0, ldx % 4 == 0, ldy % 4 == 0,
 * 16-byte aligned pointers; y may alias x.  workspace: og_batchnorm_train_workspace_bytes (0 = unsupported shape). */
size_t og_batchnorm_train_workspace_bytes(int64_t rows, int32_t channels);
int og_batchnorm_train_forward(const float* x, int64_t ldx, int64_t rows, int32_t channels, const float* weight,
                               const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                               float* y, int64_t ldy, float* save_mean, float* save_invstd, void* workspace_dev,
                               void* stream);

/* Backward of the same block (autograd of relu + torch.nn.functional.batch_norm(training=True), i.e. of models/utils.py:53-55
 * during MatchingTrainingModule.training_step): a = the block's BatchNorm INPUT (the ReLU output), dy = dL/dy, save_mean /
 * save_invstd from the forward.  Writes dz = dL/d(pre-ReLU conv output) when relu_mask != 0 (else dL/da), dweight, dbias
 * (may be NULL).  Same shape rules and workspace size as the forward; dz may alias dy. */
int og_batchnorm_train_backward(const float* a, int64_t lda, const float* dy, int64_t lddy, int64_t rows, int32_t channels,
                                const float* weight, const float* save_mean, const float* save_invstd, int32_t relu_mask,
                                float* dz, int64_t lddz, float* dweight, float* dbias, void* workspace_dev, void* stream);
/* Helpers of the 1x1-conv backward on token-major activations (dW = dZ^T X as og_gemm_nt(dZ^T, X^T); db = column sums of dZ); kept
 * for callers of the NT form -- openglue_amd.train uses og_gemm_kmajor (operands as they lie, bias gradient in the same launch) since v6:
 * dst[c][r] = src[r][c]; out[c] = sum_r x[r][c] (workspace: og_batchnorm_train_workspace_bytes(rows, channels) is enough). */
int og_transpose_f32(const float* src, int64_t ld_src, int64_t rows, int32_t cols, float* dst, int64_t ld_dst, void* stream);
int og_colsum_f32(const float* x, int64_t ldx, int64_t rows, int32_t channels, float* out, void* workspace_dev, void* stream);

int og_transpose_f32_batched(const float* src, int64_t ld_src, int64_t stride_src, int64_t rows, int32_t cols, float* dst,
                             int64_t ld_dst, int64_t stride_dst, int32_t batch, void* stream);
/* Training-mode softmax attention keeps the attention matrix, like the reference (models/superglue/attention.py:8-19; autograd
 * needs it): og_softmax_rows turns S [rows][ld] (= scale * Q K^T from og_gemm_nt) into P in place, og_softmax_rows_backward turns
 * dP (= dO V^T) into dS = scale * P o (dP - rowsum(dP o P)) in place.  Columns [cols, ld) are zeroed. */
int og_softmax_rows(float* S, int64_t ld, int64_t rows, int32_t cols, void* stream);
int og_softmax_rows_backward(const float* P, float* dP, int64_t ld, int64_t rows, int32_t cols, float scale, void* stream);

/* Backward of multi-head softmax attention without the attention matrix (flash style, exact fp32 MFMA; csrc/attention_train.hip):
 * q, dout [batch][nq][D], k, v [batch][nk][D] token-major fp32, D = num_heads * dh, head h = columns [h dh, (h+1) dh), dh in {16, 32, 64}.
 * og_attention_train_lse: lse[batch][num_heads][nq] = log sum_j exp(scale q_i . k_j).
 * og_attention_backward: delta[batch][nq][num_heads] = sum_c dout o out over the head's columns (caller); writes dk, dv [batch][nk][D] and
 * og_attention_backward_parts(nk) partial dq tensors dq_part[part][batch][nq][D] (one per 64-key block; dq = their sum: deterministic,
 * no atomics). */
int og_attention_train_lse(const float* q, const float* k, int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh,
                           float scale, float* lse, void* stream);
int og_attention_backward_parts(int32_t nk);
int og_attention_backward(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                          int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh, float scale, float* dq_part, float* dk,
                          float* dv, void* stream);
/* ABI v9 -- the same (the autograd of models/superglue/attention.py:8-19 inside attention_gnn.py:22-32) with ROW STRIDES (floats, multiples of 4, >= D) on q, k, v and on the dk, dv outputs, so that the training step hands over
 * column ranges of its [tokens][3D] projection matrix and receives dk, dv inside the [tokens][3D] gradient matrix (no slices copied out, no
 * concatenation); dout and dq_part rows stay D wide.  og_attention_delta: delta[row][h] = sum_c dout[row][h dh + c] out[row][h dh + c] for
 * contiguous [rows][num_heads * dh] tensors (what og_attention_backward wants as `delta`), one launch. */
int og_attention_backward_ld(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                             const float* lse, const float* delta, int32_t batch, int32_t nq, int32_t nk, int32_t num_heads, int32_t dh,
                             float scale, float* dq_part, float* dk, int64_t lddk, float* dv, int64_t lddv, void* stream);
int og_attention_delta(const float* dout, const float* out, int64_t rows, int32_t num_heads, int32_t dh, float* delta, void* stream);
/* ABI v9 -- glue of the training step (MatchingTrainingModule.training_step, models/matching_module.py:93-105, through SuperGlue.forward) as single
 * launches (openglue_amd/train.py):
 * og_split_f16_rows: x [rows][cols] fp32 (row stride ldx) -> (hi, lo) binary16 planes (row stride ldo), columns [0, scale_cols) multiplied by
 *   s1, then s2, first (the q columns of a q | k | v matrix: dh^-1/2, then the log2(e) of og_attention's base-2 softmax);
 * og_merge_f16: out[i] = float(hi[i]) + float(lo[i]) (og_attention's output planes back to fp32);
 * og_splitk_reduce: the `parts` partial products part[p][rows][ld] of a split-K weight gradient (og_gemm_kmajor with batch = parts) summed in
 *   part order into dW [rows][cols]; db (may be NULL) [rows] = the sums of column `cols` (the a_colsum column: ld >= cols + 4 then). */
int og_split_f16_rows(const float* x, int64_t ldx, int64_t rows, int32_t cols, int32_t scale_cols, float s1, float s2, void* hi, void* lo,
                      int64_t ldo, void* stream);
int og_merge_f16(const void* hi, const void* lo, int64_t n, float* out, void* stream);
int og_splitk_reduce(const float* part, int32_t parts, int32_t rows, int64_t ld, int32_t cols, float* dW, float* db, void* stream);

/* mutual-NN match extraction from a scores tensor [B][m+1][n+1] (matching_module.py:174-187;
 * matches1/matching_scores1 as inference.py:183-188, may be NULL).  First maximal index wins.
 * workspace: og_matches_workspace_bytes. */
size_t og_matches_workspace_bytes(int32_t batch, int32_t m, int32_t n);
int og_extract_matches(const float* scores, int32_t batch, int32_t m, int32_t n, float match_threshold,
                       int64_t* matches0, float* matching_scores0, int64_t* matches1,
                       float* matching_scores1, void* workspace_dev, void* stream);

/* ---- the steps either side of the matcher in OpenGlue's inference loop ---- */

/* models/features/utils.py:54-65 prepare_features_output + models/laf_converter.py: lafs [tokens][2][3],
 * responses [tokens] -> keypoints [tokens][2] (LAF centre) and side_info [tokens][s].
 * method: 0 'none' (s=1), 1 'scale' (2), 2 'rotation' (3), 3 'scale_rotation' (4), 4 'affine' (6);
 * log_response: response -> log(response + 0.1).  LAF scale as kornia.feature.laf.get_laf_scale. */
int og_prepare_features(const float* lafs, const float* responses, int64_t tokens, int32_t method, int32_t log_response,
                        float* keypoints, float* side_info, void* stream);

/* inference.py:192-209: order-preserving compaction of the valid matches of a batch.
 * matches0 / matching_scores0 [B][m]; lafs0 [B][m][2][3], lafs1 [B][n][2][3] (both may be NULL).
 * Outputs sized for the worst case (B*m rows): matching_idxs [K][2] (keypoint index in image 0, in image 1),
 * batch_indexes [K], confidence [K], mlafs0/mlafs1 [K][2][3], keypoints0/1 [K][2]; K is written to count_dev.
 * workspace: og_compact_workspace_bytes. */
size_t og_compact_workspace_bytes(int32_t batch, int32_t m);
int og_compact_matches(const int64_t* matches0, const float* matching_scores0, const float* lafs0, const float* lafs1,
                       int32_t batch, int32_t m, int32_t n, int64_t* matching_idxs, int64_t* batch_indexes,
                       float* confidence, float* mlafs0, float* mlafs1, float* keypoints0, float* keypoints1,
                       int32_t* count_dev, void* workspace_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENGLUE_AMD_H */

// C-ABI entry points: shape checks, host-side weight packing, workspace layout and the launch
// sequence of the whole hot path (og_forward).  See include/openglue_amd.h for the contract.
#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <dlfcn.h>
#include <stdio.h>

#include "og_common.h"

namespace {

constexpr double BN_EPS = 1e-5;   // nn.BatchNorm1d default (reference models/utils.py:55)

struct PackedLayout {
    int n_enc;                                  // conv layers of the keypoint encoder
    int enc_k[OG_MAX_HIDDEN + 1];               // padded input width of layer i
    int enc_out[OG_MAX_HIDDEN + 1];             // padded output width of layer i
    int64_t enc_w[OG_MAX_HIDDEN + 1], enc_b[OG_MAX_HIDDEN + 1];
    int64_t enc_whl;                            // last encoder conv once more, as hl32 rows of 256*w (split-f16 kernel), -1: none
    int64_t layer0, layer_stride;               // per GNN layer block
    // offsets inside a layer block (float units); weights in the hl32 row format: [N][2K] halves = N*K floats
    int64_t o_wqkv, o_bqkv, o_w0, o_b0, o_w3, o_b3;
    int64_t o_wmlp;                             // fragment-major stream of mlp_fused.hip (W0' then W3' per hidden half), -1: none
    int64_t o_wqkvs;                            // fragment-major copy of the q | k | v matrix for proj_small_kernel (few token rows), -1: none
    int64_t o_wqkvb;                            // ... and for proj_stream_kernel (batches), -1: none
    int64_t o_scale;                            // per layer: accumulator multipliers {1 / S_qkv, 1 / S_0, 1 / S_3, 0} of its three split-f16 matrices
    int64_t scales;                             // {1 / S_wp, 1 / S_enc_whl}: final projection, last encoder conv
    int64_t wp, bp, alpha, dustbin;
    int64_t total;                              // floats
    int enc_maxw;                               // widest padded hidden activation
};

inline int64_t al64(int64_t x) { return og_round_up(x, 64); }   // 256-byte aligned sections

// Width of the q (and k) block of the q | k | v planes and of the packed projection matrix: D channels, or with
// attention = 'favor_relu' the 2D random features phi(q), phi(k) (the feature map's projection is folded into in_proj_q / in_proj_k).
inline int qk_width(const og_shape& s) { return (s.flags & OG_FLAG_FAVOR_RELU) ? 2 * s.desc_dim : s.desc_dim; }
inline int qkv_width(const og_shape& s) { return 2 * qk_width(s) + s.desc_dim; }

PackedLayout packed_layout(const og_shape& s) {
    PackedLayout L{};
    const int64_t D = s.desc_dim;
    L.n_enc = s.num_hidden + 1;
    int64_t off = 0;
    int k = 32;
    L.enc_maxw = 0;
    for (int i = 0; i < L.n_enc; ++i) {
        const int out = i < s.num_hidden ? (int)og_round_up(s.hidden[i], 64) : (int)D;
        L.enc_k[i] = k; L.enc_out[i] = out;
        L.enc_w[i] = off; off = al64(off + (int64_t)out * k);
        L.enc_b[i] = off; off = al64(off + out);
        if (i < s.num_hidden && out > L.enc_maxw) L.enc_maxw = out;
        k = out;
    }
    // the last conv (no activation behind it) runs on the split-f16 kernel when its input is a hidden activation that fits the G
    // buffer as hl32 rows (api.hip: forward_impl)
    L.enc_whl = -1;
    if (L.n_enc >= 2 && L.enc_k[L.n_enc - 1] <= D) { L.enc_whl = off; off = al64(off + D * (int64_t)L.enc_k[L.n_enc - 1]); }
    int64_t lo = 0;
    L.o_wqkv = lo; lo = al64(lo + qkv_width(s) * D);
    L.o_bqkv = lo; lo = al64(lo + qkv_width(s));
    L.o_w0 = lo; lo = al64(lo + 4 * D * D);
    L.o_b0 = lo; lo = al64(lo + 2 * D);
    L.o_w3 = lo; lo = al64(lo + 2 * D * D);
    L.o_b3 = lo; lo = al64(lo + D);
    L.o_scale = lo; lo = al64(lo + 4);
    L.o_wmlp = -1;
    if (og_mlp_fused_supported((int)D)) { L.o_wmlp = lo; lo = al64(lo + (int64_t)(og_mlp_stream_bytes((int)D) / 4)); }
    L.o_wqkvs = -1;
    if (!(s.flags & OG_FLAG_FAVOR_RELU) && og_proj_stream_bytes(qkv_width(s), (int)D)) {
        L.o_wqkvs = lo; lo = al64(lo + (int64_t)(og_proj_stream_bytes(qkv_width(s), (int)D) / 4));
    }
    L.o_wqkvb = -1;
    // only where og_forward can take the batch projection kernel: the 128-d family (at D = 256 the copy was 0.75 MB per layer of dead weight and
    // packing time; the K = 256 form of the kernel stays reachable through the stage entry og_proj_block, which packs its own stream)
    if (!(s.flags & OG_FLAG_FAVOR_RELU) && D == 128 && og_proj_stream_big_bytes(qkv_width(s), (int)D)) {
        L.o_wqkvb = lo; lo = al64(lo + (int64_t)(og_proj_stream_big_bytes(qkv_width(s), (int)D) / 4));
    }
    L.layer_stride = lo;
    L.layer0 = off; off += lo * 2 * s.num_stages;
    L.wp = off; off = al64(off + D * D);
    L.bp = off; off = al64(off + D);
    L.alpha = off; off = al64(off + D);
    L.dustbin = off; off = al64(off + 1);
    L.scales = off; off = al64(off + 2);
    L.total = off;
    return L;
}

struct WorkspaceLayout {
    // float offsets; f16 planes take half a float per element, hl32 rows one float per element
    int64_t x32, xo, qkvh, qkvl, h, g, ei, ea, eb, sbuf, sink, match, attn, total;      // attn: key-split scratch of attention.hip (counters, then partial results)
    int64_t lds;
};

WorkspaceLayout workspace_layout(const og_shape& s) {
    WorkspaceLayout W{};
    const int64_t D = s.desc_dim, T = (int64_t)s.batch * ((int64_t)s.m + s.n);
    W.lds = og_round_up(s.n, 4);
    int64_t off = 0;
    const PackedLayout PL = packed_layout(s);
    const int64_t ew = PL.enc_maxw > 0 ? PL.enc_maxw : 64;
    W.x32 = off; off = al64(off + T * D);
    W.xo = off; off = al64(off + T * 2 * D);       // [T] hl32 rows of [x | O]: 4D halves each
    W.qkvh = off; off = al64(off + T * qkv_width(s) / 2); // [T][3D] halves (favor_relu: [T][5D])
    W.qkvl = off; off = al64(off + T * qkv_width(s) / 2);
    W.h = off;                                     // [T] hl32 rows of the 2D hidden activations -- only when fc.0 and fc.3 run as two
    if (!og_mlp_fused_enabled((int)D)) off = al64(off + T * 2 * D);   // launches (mlp_fused.hip keeps the hidden activation in registers)
    W.g = off; off = al64(off + T * D);
    W.ei = off; off = al64(off + T * 32);
    W.ea = off; off = al64(off + T * ew);
    W.eb = off; off = al64(off + T * ew);
    W.sbuf = off; off = al64(off + (int64_t)s.batch * s.m * W.lds);
    W.sink = off; off = al64(off + (int64_t)(og_sinkhorn_workspace_bytes(s.batch, s.m, s.n) + 3) / 4);
    W.match = off; off = al64(off + (int64_t)(og_matches_workspace_bytes(s.batch, s.m, s.n) + 3) / 4);
    W.attn = off; off = al64(off + OG_ATTN_COUNTERS + OG_ATTN_PARTIAL_FLOATS);       // 17.8 MB; used by launches of one or two pairs only
    W.total = off;
    return W;
}

int check_shape(const og_shape* s) {
    if (!s) return OG_E_INVALID;
    if (s->batch <= 0 || s->m <= 0 || s->n <= 0) return OG_E_SHAPE;
    if (s->desc_dim <= 0 || s->desc_dim % 64) return OG_E_SHAPE;
    if (s->num_heads <= 0 || s->desc_dim % s->num_heads) return OG_E_SHAPE;
    const int dh = s->desc_dim / s->num_heads;
    if (s->flags & OG_FLAG_FAVOR_RELU) {      // the reference's FAVOR attention only runs with one head (openglue_amd.h)
        if (s->num_heads != 1 || s->desc_dim > 256 || (s->flags & OG_FLAG_LINEAR_ATTENTION)) return OG_E_SHAPE;
    } else if (dh != 16 && dh != 32 && dh != 64 && dh != 128) return OG_E_SHAPE;      // 128 (round 6): the register-staged attention kernel
    if (s->num_stages < 0) return OG_E_SHAPE;
    if (s->side_info < 0 || 2 + s->side_info > 32) return OG_E_SHAPE;
    if (s->num_hidden < 0 || s->num_hidden > OG_MAX_HIDDEN) return OG_E_SHAPE;
    for (int i = 0; i < s->num_hidden; ++i)
        if (s->hidden[i] <= 0 || og_round_up(s->hidden[i], 64) > 2 * s->desc_dim) return OG_E_SHAPE;
    if (s->n > 8192) return OG_E_SHAPE;                 // Sinkhorn sweep geometry (sinkhorn.hip)
    if (s->sinkhorn_iters < 0 || !(s->sinkhorn_reg > 0.f)) return OG_E_SHAPE;
    if (s->flags & ~(OG_FLAG_RESIDUAL | OG_FLAG_USE_OFFSET | OG_FLAG_NO_DESCRIPTORS | OG_FLAG_SIREN_ENCODER | OG_FLAG_LINEAR_ATTENTION | OG_FLAG_FAVOR_RELU)) return OG_E_FLAG;
    return 0;
}

// w -> (hi, lo) of w * S, element (row, col) of an hl32 weight matrix with K columns (og_common.h).  S is the matrix's power-of-two
// pre-scale (og_weight_prescale of its largest |w|: 256 unless that would leave binary16, e.g. after a BatchNorm fold over a dead
// channel with running_var ~ 0).  Returns false only when the weight is not finite (or beyond 2^40): og_pack_weights then fails
// with OG_E_RANGE instead of packing an inf.
inline bool put_split(_Float16* W, int64_t row, int col, int K, double w, double S) {
    w *= S;
    if (!(fabs(w) <= 65504.0)) return false;
    const _Float16 hi = (_Float16)w;
    _Float16* d = W + row * 2 * K + og_hl_col(col);
    d[0] = hi;
    d[32] = (_Float16)(w - (double)hi);
    return true;
}

// a whole [rows][K] matrix (row-major doubles) as hl32 rows of S * w with S = og_weight_prescale(max |w|); *inv_scale = 1 / S
inline bool put_matrix(_Float16* W, const double* M, int64_t rows, int K, float* inv_scale, double* S_out = nullptr) {
    double mx = 0.0;
    bool ok = true;
    for (int64_t i = 0; i < rows * K; ++i) { const double a = fabs(M[i]); if (!(a <= 1e300)) ok = false; else if (a > mx) mx = a; }
    const double S = og_weight_prescale(mx);
    for (int64_t r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) ok &= put_split(W, r, k, K, M[r * K + k], S);
    *inv_scale = (float)(1.0 / S);
    if (S_out) *S_out = S;
    return ok;
}

// BatchNorm (eval) as y*g + c
void bn_affine(const og_bn& bn, int C, std::vector<double>& g, std::vector<double>& c) {
    g.resize(C); c.resize(C);
    for (int i = 0; i < C; ++i) {
        g[i] = (double)bn.weight[i] / sqrt((double)bn.running_var[i] + BN_EPS);
        c[i] = (double)bn.bias[i] - (double)bn.running_mean[i] * g[i];
    }
}

}  // namespace

extern "C" int og_abi_version(void) { return OG_ABI_VERSION; }

extern "C" int og_check_shape(const og_shape* shape) { return check_shape(shape); }

extern "C" size_t og_packed_weights_bytes(const og_shape* shape) {
    if (!shape) return 0;
    og_shape s = *shape; s.batch = s.m = s.n = 1;
    if (check_shape(&s)) return 0;
    return (size_t)packed_layout(*shape).total * sizeof(float);
}

extern "C" int og_packed_layout(const og_shape* shape, og_packed_layout_t* o) {
    if (!shape || !o) return OG_E_INVALID;
    og_shape s = *shape; s.batch = s.m = s.n = 1;
    if (int e = check_shape(&s)) return e;
    const PackedLayout L = packed_layout(*shape);
    memset(o, 0, sizeof(*o));
    o->n_enc = L.n_enc;
    for (int i = 0; i < L.n_enc; ++i) { o->enc_k[i] = L.enc_k[i]; o->enc_out[i] = L.enc_out[i]; o->enc_w[i] = L.enc_w[i]; o->enc_b[i] = L.enc_b[i]; }
    o->layer0 = L.layer0; o->layer_stride = L.layer_stride;
    o->o_wqkv = L.o_wqkv; o->o_bqkv = L.o_bqkv;
    o->o_w0 = L.o_w0; o->o_b0 = L.o_b0;
    o->o_w3 = L.o_w3; o->o_b3 = L.o_b3; o->o_wmlp = L.o_wmlp; o->o_wqkvs = L.o_wqkvs; o->o_wqkvb = L.o_wqkvb; o->o_scale = L.o_scale; o->scales = L.scales;
    o->wp = L.wp; o->bp = L.bp; o->alpha = L.alpha; o->dustbin = L.dustbin; o->total = L.total;
    return 0;
}

extern "C" int og_sinkhorn_status(const void* workspace_dev, int32_t batch, int32_t m, int32_t n);      // sinkhorn.hip

// include/openglue_amd.h
extern "C" int og_forward_status(const og_shape* shape, const void* workspace_dev) {
    if (!workspace_dev || check_shape(shape)) return -1;
    const WorkspaceLayout W = workspace_layout(*shape);
    return og_sinkhorn_status((const float*)workspace_dev + W.sink, shape->batch, shape->m, shape->n);
}

extern "C" size_t og_workspace_bytes(const og_shape* shape) {
    if (check_shape(shape)) return 0;
    return (size_t)workspace_layout(*shape).total * sizeof(float);
}

extern "C" int og_pack_weights(const og_shape* shape, const og_params* P, void* packed_host) {
    if (!shape || !P || !packed_host) return OG_E_INVALID;
    og_shape chk = *shape; chk.batch = chk.m = chk.n = 1;
    if (int e = check_shape(&chk)) return e;
    const og_shape& s = *shape;
    const PackedLayout L = packed_layout(s);
    const int D = s.desc_dim, D2 = 2 * D;
    float* out = (float*)packed_host;
    memset(out, 0, (size_t)L.total * sizeof(float));

    // ---- keypoint encoder: [Conv, ReLU, BN]*h + Conv; BN_i folds into Conv_{i+1} ----
    {
        std::vector<double> g, c;
        int in_real = 2 + s.side_info;
        for (int i = 0; i < L.n_enc; ++i) {
            const int out_real = i < s.num_hidden ? s.hidden[i] : D;
            const og_conv& cv = P->enc_conv[i];
            if (!cv.weight || !cv.bias) return OG_E_INVALID;
            float* W = out + L.enc_w[i];
            float* b = out + L.enc_b[i];
            for (int o = 0; o < out_real; ++o) {
                double bb = cv.bias[o];
                for (int k = 0; k < in_real; ++k) {
                    const double w = cv.weight[(int64_t)o * in_real + k];
                    if (i == 0) W[(int64_t)o * L.enc_k[i] + k] = (float)w;
                    else { W[(int64_t)o * L.enc_k[i] + k] = (float)(w * g[k]); bb += w * c[k]; }
                }
                b[o] = (float)bb;
            }
            if (i == L.n_enc - 1 && L.enc_whl >= 0) {        // the same folded weights as hl32 rows of 256*w
                _Float16* Whl = (_Float16*)(out + L.enc_whl);
                std::vector<double> Wd((size_t)out_real * L.enc_k[i], 0.0);
                for (int o = 0; o < out_real; ++o)
                    for (int k = 0; k < in_real; ++k) Wd[(size_t)o * L.enc_k[i] + k] = (double)W[(int64_t)o * L.enc_k[i] + k];
                if (!put_matrix(Whl, Wd.data(), out_real, L.enc_k[i], out + L.scales + 1)) return OG_E_RANGE;
            }
            if (i < s.num_hidden) {
                if (s.flags & OG_FLAG_SIREN_ENCODER) {       // no BatchNorm between the layers: identity fold
                    g.assign(out_real, 1.0); c.assign(out_real, 0.0);
                } else {
                    const og_bn& bn = P->enc_bn[i];
                    if (!bn.weight || !bn.bias || !bn.running_mean || !bn.running_var) return OG_E_INVALID;
                    bn_affine(bn, out_real, g, c);
                }
            }
            in_real = out_real;
        }
    }

    // ---- GNN layers ----
    const int dh = D / s.num_heads;
    // attention.py:12 `* embed_dim ** -0.5`, times log2(e): the attention kernel's softmax is base 2.
    // Linear attention (attention.py:22-40) has no scale.
    const double qscale = (s.flags & OG_FLAG_LINEAR_ATTENTION) ? 1.0 : 1.4426950408889634 / sqrt((double)dh);
    const bool offset = s.flags & OG_FLAG_USE_OFFSET;
    const bool favor = s.flags & OG_FLAG_FAVOR_RELU;
    const int wq = qk_width(s);
    std::vector<double> Wm((size_t)D2 * D), prod((size_t)D2 * D), g, c;
    // every split-f16 matrix is assembled in double first: its power-of-two pre-scale depends on its largest |w| (put_matrix)
    std::vector<double> Wqd((size_t)qkv_width(s) * D), W0d((size_t)D2 * D2), W3d((size_t)D * D2);
    bool ok = true;                     // every split-f16 weight is finite
    for (int l = 0; l < 2 * s.num_stages; ++l) {
        if (!P->layers) return OG_E_INVALID;
        const og_layer_params& lp = P->layers[l];
        float* base = out + L.layer0 + (int64_t)l * L.layer_stride;
        _Float16* Wqkv = (_Float16*)(base + L.o_wqkv);
        float* bqkv = base + L.o_bqkv;
        const og_conv* proj[3] = {&lp.in_proj_q, &lp.in_proj_k, &lp.in_proj_v};
        for (int p = 0; p < 3; ++p) {
            if (!proj[p]->weight || !proj[p]->bias) return OG_E_INVALID;
            const int64_t row0 = p == 0 ? 0 : p == 1 ? wq : 2 * wq;      // q | k | v row blocks of the packed matrix
            if (favor && p < 2) {
                // randomized_kernel (attention.py:91-95): phi(x) = relu(P (x d^-1/4)) + eps on x = W t + b (d = D: one head)
                //   => relu((d^-1/4 P W) t + d^-1/4 P b) + eps: the projection folds into the 1x1 conv, 2D output rows; the ReLU is
                // the GEMM's epilogue, eps is added where the features are read (linear_attention.hip)
                if (!lp.favor_projection) return OG_E_INVALID;
                const double fs = pow((double)D, -0.25);
                std::vector<double> row(D);
                for (int f = 0; f < 2 * D; ++f) {
                    const float* pf = lp.favor_projection + (int64_t)f * D;
                    std::fill(row.begin(), row.end(), 0.0);
                    double bb = 0.0;
                    for (int o = 0; o < D; ++o) {
                        const double pw = fs * (double)pf[o];
                        const float* wr = proj[p]->weight + (int64_t)o * D;
                        for (int k = 0; k < D; ++k) row[k] += pw * (double)wr[k];
                        bb += pw * (double)proj[p]->bias[o];
                    }
                    for (int k = 0; k < D; ++k) Wqd[(size_t)(row0 + f) * D + k] = row[k];
                    bqkv[row0 + f] = (float)bb;
                }
                continue;
            }
            const double sc = (p == 0 && !favor) ? qscale : 1.0;
            for (int o = 0; o < D; ++o)
                for (int k = 0; k < D; ++k) Wqd[(size_t)(row0 + o) * D + k] = proj[p]->weight[(int64_t)o * D + k] * sc;
            for (int i = 0; i < D; ++i) bqkv[row0 + i] = (float)(proj[p]->bias[i] * sc);
        }
        float* scl = base + L.o_scale;      // {1 / S_qkv, 1 / S_0, 1 / S_3, 0}
        double Sq = OG_W_SCALE;
        ok &= put_matrix(Wqkv, Wqd.data(), qkv_width(s), D, scl + 0, &Sq);
        if (L.o_wqkvs >= 0) ok &= og_pack_proj_stream(qkv_width(s), D, Wqd.data(), base + L.o_wqkvs, Sq);
        if (L.o_wqkvb >= 0) ok &= og_pack_proj_stream_big(qkv_width(s), D, Wqd.data(), base + L.o_wqkvb, Sq);
        // fc.0 on y = [x ; msg] (or [x - msg ; msg] with use_offset, attention_gnn.py:51-54), msg = Wo O + bo:
        //   W0 y = W0a x + Wm (Wo O + bo),  Wm = W0b (- W0a)   ->  [W0a | Wm Wo] [x ; O] + (b0 + Wm bo)
        if (!lp.fc0.weight || !lp.fc0.bias || !lp.out_proj.weight || !lp.out_proj.bias || !lp.fc3.weight || !lp.fc3.bias)
            return OG_E_INVALID;
        _Float16* W0 = (_Float16*)(base + L.o_w0);
        float* b0 = base + L.o_b0;
        for (int o = 0; o < D2; ++o)
            for (int k = 0; k < D; ++k) {
                const double wa = lp.fc0.weight[(int64_t)o * D2 + k], wb = lp.fc0.weight[(int64_t)o * D2 + D + k];
                W0d[(size_t)o * D2 + k] = wa;
                Wm[(size_t)o * D + k] = offset ? wb - wa : wb;
            }
        std::fill(prod.begin(), prod.end(), 0.0);
        for (int o = 0; o < D2; ++o) {
            double* pr = &prod[(size_t)o * D];
            double bb = lp.fc0.bias[o];
            for (int k = 0; k < D; ++k) {
                const double w = Wm[(size_t)o * D + k];
                const float* wo = lp.out_proj.weight + (int64_t)k * D;
                for (int j = 0; j < D; ++j) pr[j] += w * (double)wo[j];
                bb += w * (double)lp.out_proj.bias[k];
            }
            for (int j = 0; j < D; ++j) W0d[(size_t)o * D2 + D + j] = pr[j];
            b0[o] = (float)bb;
        }
        double S0 = OG_W_SCALE, S3 = OG_W_SCALE;
        ok &= put_matrix(W0, W0d.data(), D2, D2, scl + 1, &S0);
        // fc.3 with BN(2D) folded in
        if (!lp.fc_bn.weight || !lp.fc_bn.bias || !lp.fc_bn.running_mean || !lp.fc_bn.running_var) return OG_E_INVALID;
        bn_affine(lp.fc_bn, D2, g, c);
        _Float16* W3 = (_Float16*)(base + L.o_w3);
        float* b3 = base + L.o_b3;
        for (int o = 0; o < D; ++o) {
            double bb = lp.fc3.bias[o];
            for (int k = 0; k < D2; ++k) {
                const double w = lp.fc3.weight[(int64_t)o * D2 + k];
                W3d[(size_t)o * D2 + k] = w * g[k];
                bb += w * c[k];
            }
            b3[o] = (float)bb;
        }
        ok &= put_matrix(W3, W3d.data(), D, D2, scl + 2, &S3);
        if (L.o_wmlp >= 0) ok &= og_pack_mlp_stream(D, W0d.data(), W3d.data(), base + L.o_wmlp, S0, S3);
    }

    // ---- tail ----
    if (!P->linear_proj.weight || !P->linear_proj.bias) return OG_E_INVALID;
    {   // final projection: hl32 rows of 256*w like the GNN matrices (it runs on the split-f16 kernel, reading the x rows of XO)
        _Float16* Wp = (_Float16*)(out + L.wp);
        std::vector<double> Wd((size_t)D * D);
        for (int64_t i = 0; i < (int64_t)D * D; ++i) Wd[i] = P->linear_proj.weight[i];
        ok &= put_matrix(Wp, Wd.data(), D, D, out + L.scales + 0);
    }
    memcpy(out + L.bp, P->linear_proj.bias, sizeof(float) * (size_t)D);
    if (s.flags & OG_FLAG_RESIDUAL) {
        if (!P->mix_coefs) return OG_E_INVALID;
        for (int i = 0; i < D; ++i) out[L.alpha + i] = (float)(1.0 / (1.0 + exp(-(double)P->mix_coefs[i])));   // superglue.py:60
    }
    out[L.dustbin] = P->dustbin_score;
    if (!ok) return OG_E_RANGE;
    for (int64_t i = 0; i < L.total; ++i)      // fp32 sections (encoder, biases): a non-finite fold (var + eps <= 0, inf weights)
        if (!std::isfinite(out[i])) {
            // the split-f16 sections hold pairs of halves, not floats: a NaN bit pattern there cannot occur for finite
            // halves with |hi| <= 65504 unless both halves have all-ones exponents, which put_split already excluded
            return OG_E_RANGE;
        }
    return 0;
}

namespace {

// Optional per-kernel-class timing with HIP events recorded on the launch stream (og_forward_profiled).
struct Profiler {
    hipStream_t st;
    std::vector<hipEvent_t> ev;     // start/stop pairs
    std::vector<int> cls;
    int begin(int c) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        ev.push_back(a); ev.push_back(b); cls.push_back(c);
        (void)hipEventRecord(a, st);
        return (int)cls.size() - 1;
    }
    void end(int i) { if (i >= 0) (void)hipEventRecord(ev[2 * i + 1], st); }
};
// roctx ranges around the stages of a forward (tracing): OG_ROCTX=1 makes every og_forward* call push / pop named ranges --
// "og_forward", "encoder", "gnn self l", "gnn cross l", "final_proj", "scores", "sinkhorn", "matches" -- which `rocprofv3 --marker-trace` records next to the
// kernel trace.  The marker library is looked up at run time (no link dependency; absent library or unset variable: the calls are no-ops).
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("OG_ROCTX");
        if (!e || e[0] != '1') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
struct Range {
    static const Roctx& rt() { static const Roctx r; return r; }
    bool on;
    explicit Range(const char* name) : on(rt().push != nullptr) { if (on) rt().push(name); }
    Range(const char* name, int idx) : on(rt().push != nullptr) {
        if (on) { char buf[48]; snprintf(buf, sizeof buf, "%s %d", name, idx); rt().push(buf); }
    }
    ~Range() { if (on) rt().pop(); }
    Range(const Range&) = delete;
};
struct Scope {
    Profiler* p; int i;
    Scope(Profiler* pp, int c) : p(pp), i(pp ? pp->begin(c) : -1) {}
    ~Scope() { if (p) p->end(i); }
};

int forward_impl(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                 const og_outputs* outp, void* stream, Profiler* prof, const RaggedDesc* rag = nullptr,
                 const EncoderRagged* er0 = nullptr, const EncoderRagged* er1 = nullptr, int tap = -1, float* tap_x = nullptr,
                 bool encoder_only = false) {
    static const og_outputs no_outputs{};
    if (encoder_only && !outp) outp = &no_outputs;          // og_keypoint_encoder: stops after tap 0, writes nothing but tap_x
    if (!shape || !in || !packed_dev || !workspace_dev || !outp) return OG_E_INVALID;
    if (int e = check_shape(shape)) return e;
    og_clear_status();
    const og_shape& s = *shape;
    if (!in->keypoints0 || !in->keypoints1 || !in->descriptors0 || !in->descriptors1 || (!outp->scores && !encoder_only)) return OG_E_INVALID;
    if (s.side_info > 0 && (!in->side_info0 || !in->side_info1)) return OG_E_INVALID;
    if ((outp->matches0 == nullptr) != (outp->matching_scores0 == nullptr)) return OG_E_INVALID;
    if ((outp->matches1 == nullptr) != (outp->matching_scores1 == nullptr)) return OG_E_INVALID;
    if (outp->matches1 && !outp->matches0) return OG_E_INVALID;
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)workspace_dev & 15) || ((uintptr_t)in->descriptors0 & 15) ||
        ((uintptr_t)in->descriptors1 & 15))
        return OG_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const PackedLayout L = packed_layout(s);
    const WorkspaceLayout W = workspace_layout(s);
    const float* pk = (const float*)packed_dev;
    float* ws = (float*)workspace_dev;
    const int D = s.desc_dim, D2 = 2 * D, B = s.batch, m = s.m, n = s.n;
    const bool favor = s.flags & OG_FLAG_FAVOR_RELU;
    const int WQ = qk_width(s), QW = qkv_width(s);      // q | k | v planes: columns [0, WQ) | [WQ, 2WQ) | [2WQ, QW); QW = 3D (5D: favor_relu)
    // uniform batch: B sets of m (n) tokens; ragged batch: packed sets, m and n are the maxima
    const int64_t T0 = rag ? rag->off0[B] : (int64_t)B * m, T1 = rag ? rag->off1[B] : (int64_t)B * n, T = T0 + T1;
    float* X32 = ws + W.x32; float* G = ws + W.g; float* Sb = ws + W.sbuf;
    const int D4 = 4 * D;
    _Float16* XO = (_Float16*)(ws + W.xo);             // [T] hl32 rows of [x | O]: 4D halves, x in the first 2D, O in the last 2D
    _Float16* QKVh = (_Float16*)(ws + W.qkvh); _Float16* QKVl = (_Float16*)(ws + W.qkvl);  // planes [T][QW]: q | k | v
    _Float16* Hb = (_Float16*)(ws + W.h);              // [T] hl32 rows of the hidden activations (4D halves)
    int rc;

    // exact-fp32 GEMM (encoder, final projection, score matrix)
    auto gemm = [&](const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                    const float* bias, int relu, const float* res, int64_t ldr, _Float16* Chl, int64_t ldch) -> int {
        GemmArgs g{};
        g.A = A; g.lda = lda; g.strideA = 0; g.B = Bm; g.ldb = ldb; g.strideB = 0; g.C = C; g.ldc = ldc; g.strideC = 0;
        g.M = (int)M; g.N = N; g.K = K; g.batch = 1; g.bias = bias; g.relu = relu; g.res = res; g.ldr = ldr; g.strideR = 0;
        g.alpha = nullptr; g.scale = 1.f; g.Ct = nullptr; g.ldct = 0; g.strideCt = 0; g.ct_rows = 1;
        g.Ch = Chl; g.Cl = Chl ? Chl + 32 : nullptr; g.ldch = ldch; g.c_hl = 1;      // hl32 copy for the f16x3 consumers
        Scope sc(prof, OG_STAGE_GEMM);
        return og_launch_gemm(g, st);
    };
    // split-f16 GEMM (the GNN's 1x1 convolutions): hl32 rows in; fp32 and/or hl32 rows (c_hl) or planes out
    auto gemmh = [&](const _Float16* A, const float* wbase, int64_t o_w, int64_t wrow0, int64_t M, int N, int K,
                     const float* bias, int relu, const _Float16* res_hl, float* C32, _Float16* Ch, _Float16* Cl, int64_t ldch,
                     int c_hl) -> int {
        GemmHArgs g{};
        g.scale_dev = wbase + L.o_scale + (o_w == L.o_wqkv ? 0 : o_w == L.o_w0 ? 1 : 2);      // the matrix's own 1 / pre-scale
        g.A = A; g.lda = D4;
        g.B = (const _Float16*)(wbase + o_w) + wrow0 * 2 * K; g.ldb = 2 * K;
        g.M = (int)M; g.N = N; g.K = K; g.scale = (float)(1.0 / OG_W_SCALE); g.bias = bias; g.relu = relu; g.res = nullptr; g.ldr = D; g.res_hl = res_hl; g.ldrh = D4;
        g.C32 = C32; g.ldc = D; g.Ch = Ch; g.Cl = Cl; g.ldch = ldch; g.c_hl = c_hl;
        Scope sc(prof, OG_STAGE_GEMM_F16X3);
        return og_launch_gemm_f16x3(g, st);
    };

    Range r_forward("og_forward");
    // ---- 1. keypoint encoder (superglue.py:41-55): x = desc + MLP([k^, side]) -> X32 and the x half of XO ----
    {
        Range r_stage("encoder");
        float* EI = ws + W.ei;         // [T][32]
        float* Ea = ws + W.ea;         // [T][enc_maxw]
        float* Eb = ws + W.eb;
        {
            Scope sc(prof, OG_STAGE_ENCODER_INPUT);
            if ((rc = og_launch_encoder_input(in->keypoints0, in->side_info0, T0, s.side_info, in->image0_wh[0], in->image0_wh[1], EI, st, er0))) return rc;
            if ((rc = og_launch_encoder_input(in->keypoints1, in->side_info1, T1, s.side_info, in->image1_wh[0], in->image1_wh[1], EI + T0 * 32, st, er1))) return rc;
        }
        const float* cur = EI; int64_t ldcur = 32;
        // The last conv carries 3/4 of the encoder's FLOPs and has no activation behind it: it runs on the split-f16 kernel
        // (fp32-class accuracy, 16x the matrix rate of exact fp32).  Its input, the last hidden activation, is written as hl32
        // rows by the epilogue of the conv before it -- into G, which is free until the final projection.
        const bool tail_f16 = L.enc_whl >= 0;
        const int last = L.n_enc - 1;
        _Float16* Ehl = (_Float16*)G;
        for (int i = 0; i < L.n_enc; ++i) {
            const float* Wi = pk + L.enc_w[i]; const float* bi = pk + L.enc_b[i];
            if (i < last) {
                float* dst = (i & 1) ? Eb : Ea;
                const int act = (s.flags & OG_FLAG_SIREN_ENCODER) ? 2 : 1;      // sin(30 x) or ReLU (+ folded BatchNorm)
                const bool to_hl = tail_f16 && i == last - 1;                   // only the hl32 rows are read again
                if ((rc = gemm(cur, ldcur, Wi, L.enc_k[i], to_hl ? nullptr : dst, L.enc_maxw, T, L.enc_out[i], L.enc_k[i], bi, act, nullptr, 0,
                               to_hl ? Ehl : nullptr, to_hl ? 2 * L.enc_out[i] : 0))) return rc;
                cur = dst; ldcur = L.enc_maxw;
            } else {
                const bool nd = s.flags & OG_FLAG_NO_DESCRIPTORS;
                const float* res[2] = {nd ? nullptr : in->descriptors0, nd ? nullptr : in->descriptors1};
                const int64_t r0[2] = {0, T0}, R[2] = {T0, T1};
                for (int side = 0; side < 2; ++side) {
                    if (tail_f16) {
                        const int K = L.enc_k[i];
                        GemmHArgs g{};
                        g.A = Ehl + r0[side] * 2 * K; g.lda = 2 * K;
                        g.B = (const _Float16*)(pk + L.enc_whl); g.ldb = 2 * K;
                        g.M = (int)R[side]; g.N = D; g.K = K; g.scale = (float)(1.0 / OG_W_SCALE); g.scale_dev = pk + L.scales + 1; g.bias = bi; g.relu = 0;
                        g.res = res[side]; g.ldr = D; g.res_hl = nullptr; g.ldrh = 0;
                        g.C32 = nullptr; g.ldc = D; g.Ch = XO + r0[side] * D4; g.Cl = g.Ch + 32; g.ldch = D4; g.c_hl = 1;
                        Scope sc(prof, OG_STAGE_GEMM_F16X3);
                        if ((rc = og_launch_gemm_f16x3(g, st))) return rc;
                    } else {
                        if ((rc = gemm(cur + r0[side] * ldcur, ldcur, Wi, L.enc_k[i], X32 + r0[side] * D, D, R[side], D, L.enc_k[i], bi, 0, res[side], D,
                                       XO + r0[side] * D4, D4))) return rc;
                    }
                }
            }
        }
    }

    // og_forward_tap: the residual stream x (all T token rows, fp32 [T][D]) after the encoder (tap 0) or after GNN layer tap - 1
    auto tap_here = [&](int idx) -> int {
        if (tap != idx || !tap_x) return 0;
        return og_launch_merge_f16_hl(XO, T, D, D4, tap_x, D, st);
    };
    if ((rc = tap_here(0))) return rc;
    if (encoder_only) return og_launch_status();

    // ---- 2. attentional GNN (attention_gnn.py:84-93) ----
    const int dh = D / s.num_heads;
    // the arrival counters of the key-split attention launches (one or two pairs) start at zero; the kernels re-arm them themselves
    if (hipMemsetAsync(ws + W.attn, 0, OG_ATTN_COUNTERS * sizeof(int), st) != hipSuccess) return OG_E_INVALID;
    auto attention = [&](int nz, int split, int64_t qb0, int64_t qs0, int nq0, int64_t kb0, int64_t ks0, int nk0,
                         int64_t qb1, int64_t qs1, int nq1, int64_t kb1, int64_t ks1, int nk1, int rag_mode) -> int {
        AttnArgs a{};
        a.rag = rag; a.rag_mode = rag_mode;
        a.qh = QKVh; a.ql = QKVl; a.ldq = QW; a.kh = QKVh + WQ; a.kl = QKVl + WQ; a.ldk = QW;
        a.vh = QKVh + 2 * WQ; a.vl = QKVl + 2 * WQ; a.ldv = QW;
        a.feat = favor ? WQ : 0;
        a.oh = XO + D2; a.ol = XO + D2 + 32; a.ldo = D4; a.o_hl = 1;        // O = channels D..2D-1 of the [x | O] rows
        a.nz = nz; a.num_heads = s.num_heads; a.dh = dh; a.split = split;
        a.counters = reinterpret_cast<int*>(ws + W.attn); a.partial = ws + W.attn + OG_ATTN_COUNTERS;
        a.q_base[0] = qb0; a.q_step[0] = qs0; a.nq[0] = nq0; a.kv_base[0] = kb0; a.kv_step[0] = ks0; a.nk[0] = nk0;
        a.q_base[1] = qb1; a.q_step[1] = qs1; a.nq[1] = nq1; a.kv_base[1] = kb1; a.kv_step[1] = ks1; a.nk[1] = nk1;
        Scope sc(prof, OG_STAGE_ATTENTION);
        if (favor) return og_launch_favor_attention(a, st);
        return (s.flags & OG_FLAG_LINEAR_ATTENTION) ? og_launch_linear_attention(a, st) : og_launch_attention(a, st);
    };
    // q / k / v projections of token rows [r0, r0 + R): columns [c0, c1) of the q | k | v planes = rows [c0, c1) of the packed
    // projection matrix.  One launch; with favor_relu the feature blocks (columns < 2 WQ) carry the ReLU of the feature map and the
    // value block does not, so a range that spans both is two launches.
    // Few rows (<= 8192 per launch, the single-pair regime): proj_small_kernel, 32-token workgroups over a fragment-major copy of the
    // matrix -- 17.5 us per launch of the 128-token tile GEMM otherwise, 36 launches per step.  OG_PROJ_SMALL=0 / 1 forces.
    auto proj_small_ok = [&](int64_t R) {
        static const int mode = [] { const char* e = getenv("OG_PROJ_SMALL"); return e ? atoi(e) : -1; }();
        return L.o_wqkvs >= 0 && !favor && (mode >= 0 ? mode != 0 : R <= 8192) && R < ((int64_t)1 << 30);
    };
    auto proj_small = [&](const float* lw, int64_t r0, int64_t R, int split_row, int a0, int a1, int b0, int b1) -> int {
        Scope sc(prof, OG_STAGE_GEMM_F16X3);
        return og_launch_proj_small(XO + r0 * D4, D4, (int)R, D, (const char*)(lw + L.o_wqkvs), lw + L.o_bqkv, lw + L.o_scale,
                                    QKVh + r0 * QW, QKVl + r0 * QW, QW, split_row, a0, a1, b0, b1, st);
    };
    // Batches (more than 8192 rows per launch) of the 128-d family: proj_stream_kernel, 128-token workgroups with the x fragments in registers and the
    // weights through an LDS ring (mlp_fused.hip).  At D = 128 the q | k | v matrix (N = 384) has no 256-tile form and the 128-token tile GEMM runs 4
    // k-stages per tile; at D = 256 the 256-tile GEMM stays faster in the whole step (og_proj_stream_wanted).  OG_PROJ_STREAM=0 / 1 forces.
    // The plane rows (QW halves) are whole 128-byte lines: QW = 3D, D a multiple of 64.
    auto proj_stream_ok = [&](int64_t R, bool full = false) {
        return L.o_wqkvb >= 0 && !favor && og_proj_stream_wanted((int)(R < ((int64_t)1 << 30) ? R : 0), D, full) && R < ((int64_t)1 << 30) && QW % 64 == 0 &&
               !(((uintptr_t)QKVh | (uintptr_t)QKVl) & 127);
    };
    auto proj_stream = [&](const float* lw, int64_t r0, int64_t R, int split_row, int a0, int a1, int b0, int b1) -> int {      // ranges in channels
        Scope sc(prof, OG_STAGE_GEMM_F16X3);
        return og_launch_proj_stream(XO + r0 * D4, D4, (int)R, D, (const char*)(lw + L.o_wqkvb), lw + L.o_bqkv, lw + L.o_scale,
                                     QKVh + r0 * QW, QKVl + r0 * QW, QW, split_row, a0 / 128, a1 / 128, b0 / 128, b1 / 128, st);
    };
    auto qkv_proj = [&](const float* lw, int64_t r0, int64_t R, int c0, int c1) -> int {
        if (proj_small_ok(R) && c0 % 32 == 0 && c1 % 32 == 0) return proj_small(lw, r0, R, 0, 0, 0, c0 / 32, c1 / 32);
        if (proj_stream_ok(R, c0 == 0 && c1 == QW) && c0 % 128 == 0 && c1 % 128 == 0 && ((r0 * QW * 2) % 128 == 0)) return proj_stream(lw, r0, R, 0, 0, 0, c0, c1);
        const int cut = favor ? 2 * WQ : c1;
        const int ca[2] = {c0, c0 < cut && cut < c1 ? cut : c1}, cb[2] = {ca[1], c1};
        for (int part = 0; part < 2; ++part) {
            const int a0 = part ? cb[0] : ca[0], a1 = part ? cb[1] : ca[1];
            if (a1 <= a0) continue;
            const int relu = favor && a0 < 2 * WQ;
            if (int e = gemmh(XO + r0 * D4, lw, L.o_wqkv, a0, R, a1 - a0, D, lw + L.o_bqkv + a0, relu, nullptr, nullptr, QKVh + r0 * QW + a0,
                              QKVl + r0 * QW + a0, QW, 0)) return e;
        }
        return 0;
    };
    // message MLP on token rows [r0, r0+R):  h = relu([x;O] W0'^T + b0') ; x += h W3'^T + b3'  (x kept in fp32 AND as planes)
    // message MLP on token rows [r0, r0+R):  h = relu([x;O] W0'^T + b0') ; x += h W3'^T + b3'.  x lives as hl32 (hi, lo)
    // rows (the residual is read from them: 2^-22 relative per layer); no fp32 copy is kept.
    const bool fused_mlp = og_mlp_fused_enabled(D) && L.o_wmlp >= 0;
    auto mlp = [&](const float* lw, int64_t r0, int64_t R) -> int {
        if (fused_mlp) {          // one launch, the hidden activation never leaves the registers (mlp_fused.hip)
            MlpFusedArgs a{};
            a.XO = XO + r0 * D4; a.ld = D4; a.M = (int)R; a.wstream = (const char*)(lw + L.o_wmlp);
            a.b0 = lw + L.o_b0; a.b3 = lw + L.o_b3; a.scale = (float)(1.0 / OG_W_SCALE); a.scales_dev = lw + L.o_scale + 1;
            Scope sc(prof, OG_STAGE_MLP_FUSED);
            return og_launch_mlp_fused(a, D, st);
        }
        int e = gemmh(XO + r0 * D4, lw, L.o_w0, 0, R, D2, D2, lw + L.o_b0, 1, nullptr, nullptr, Hb + r0 * D4, nullptr, D4, 1);
        if (e) return e;
        return gemmh(Hb + r0 * D4, lw, L.o_w3, 0, R, D, D2, lw + L.o_b3, 0, XO + r0 * D4, nullptr, XO + r0 * D4, nullptr, D4, 1);
    };
    for (int l = 0; l < s.num_stages; ++l) {
        // self layer 2l: both images through the same weights (attention_gnn.py:63-66)
        const float* lw = pk + L.layer0 + (int64_t)(2 * l) * L.layer_stride;
        {
            Range r_stage("gnn self", l);
            if ((rc = qkv_proj(lw, 0, T, 0, QW))) return rc;
            if ((rc = attention(2 * B, B, 0, m, m, 0, m, m, T0, n, n, T0, n, n, 1))) return rc;
            if ((rc = mlp(lw, 0, T))) return rc;
        }
        if ((rc = tap_here(2 * l + 1))) return rc;
        // cross layer 2l+1: image 0 first, then image 1 against the UPDATED image 0 (attention_gnn.py:74-77)
        lw = pk + L.layer0 + (int64_t)(2 * l + 1) * L.layer_stride;
        Range r_stage("gnn cross", l);
        for (int side = 0; side < 2; ++side) {
            const int64_t qr0 = side ? T0 : 0, qR = side ? T1 : T0;       // query rows
            if (side == 0) {
                // image 1 is still untouched: its k, v (for this half) and its q (for the second half) in ONE launch
                // ... and the q of image 0 with them when the shapes allow it (whole 256-row / 256-column tiles): rows < T0 of the launch
                // stop after the q columns.  [Two launches: 56 + 20 us at C2; one: the 512 tiles are exactly two rounds of the 256 CUs.]
                GemmHArgs g{};
                g.A = XO; g.lda = D4; g.B = (const _Float16*)(lw + L.o_wqkv); g.ldb = 2 * D;
                g.M = (int)T; g.N = QW; g.K = D; g.scale = (float)(1.0 / OG_W_SCALE); g.scale_dev = lw + L.o_scale; g.bias = lw + L.o_bqkv;
                g.Ch = QKVh; g.Cl = QKVl; g.ldch = QW; g.c_hl = 0; g.ldc = D; g.ldr = D; g.ldrh = D4;
                g.split_row = (int)T0; g.split_n = WQ;
                if (proj_small_ok(T) && T0 % 32 == 0 && WQ % 32 == 0) {      // rows of image 0: the q blocks only
                    if ((rc = proj_small(lw, 0, T, (int)T0, 0, WQ / 32, 0, QW / 32))) return rc;
                } else if (proj_stream_ok(T) && T0 % 128 == 0 && WQ % 128 == 0) {
                    if ((rc = proj_stream(lw, 0, T, (int)T0, 0, WQ, 0, QW))) return rc;
                } else if (!favor && !rag && T < (int64_t)1 << 30 && og_gemm_f16x3_row_split_ok(g)) {
                    Scope sc(prof, OG_STAGE_GEMM_F16X3);
                    if ((rc = og_launch_gemm_f16x3(g, st))) return rc;
                } else {
                    if ((rc = qkv_proj(lw, T0, T1, 0, QW))) return rc;
                    if ((rc = qkv_proj(lw, 0, T0, 0, WQ))) return rc;
                }
            } else {
                // k, v of the UPDATED image 0
                if ((rc = qkv_proj(lw, 0, T0, WQ, QW))) return rc;
            }
            if (side == 0) rc = attention(B, B, 0, m, m, T0, n, n, 0, 0, 0, 0, 0, 0, 2);
            else rc = attention(B, B, T0, n, n, 0, m, m, 0, 0, 0, 0, 0, 0, 3);
            if (rc) return rc;
            if ((rc = mlp(lw, qr0, qR))) return rc;
        }
        if ((rc = tap_here(2 * l + 2))) return rc;
    }

    // ---- 3. final projection + residual mix (superglue.py:58-62): G as hl32 rows (operands of the score GEMM) and the
    //         channel-first context_descriptors.  Split-f16 kernel on the x rows of XO (fp32-class, 16/3 the fp32-MFMA rate).
    _Float16* Gh = (_Float16*)G;                       // [T] hl32 rows of 2D halves (the slot holds T*D floats)
    for (int side = 0; side < 2; ++side) {
        Range r_stage("final_proj", side);
        const int64_t r0 = side ? T0 : 0, R = side ? T1 : T0;
        const bool resid = s.flags & OG_FLAG_RESIDUAL;
        GemmHArgs g{};
        g.A = XO + r0 * D4; g.lda = D4; g.B = (const _Float16*)(pk + L.wp); g.ldb = D2;
        g.M = (int)R; g.N = D; g.K = D; g.scale = (float)(1.0 / OG_W_SCALE); g.scale_dev = pk + L.scales; g.bias = pk + L.bp; g.relu = 0;
        g.res = resid ? (side ? in->descriptors1 : in->descriptors0) : nullptr; g.ldr = D;
        g.alpha = resid ? pk + L.alpha : nullptr;
        g.Ch = Gh + r0 * D2; g.Cl = g.Ch + 32; g.ldch = D2; g.c_hl = 1;
        g.Ct = side ? outp->context_descriptors1 : outp->context_descriptors0;
        g.ct_rows = side ? n : m; g.ldct = g.ct_rows;
        if (rag && g.Ct) { g.rag = rag; g.ct_rag = side ? 2 : 1; }     // ragged: per-pair [D][m_b] blocks, packed
        Scope sc(prof, OG_STAGE_GEMM_F16X3);
        if ((rc = og_launch_gemm_f16x3(g, st))) return rc;
    }

    // ---- 4. score matrix S = g0 g1^T * D^-1/2 (superglue.py:64, 81-86): one batched split-f16 launch over the pairs ----
    {
        GemmHArgs g{};
        g.A = Gh; g.lda = D2; g.strideA = (int64_t)m * D2; g.B = Gh + T0 * D2; g.ldb = D2; g.strideB = (int64_t)n * D2;
        g.C32 = Sb; g.ldc = W.lds; g.strideC32 = (int64_t)m * W.lds; g.M = m; g.N = n; g.K = D; g.batch = B;
        g.scale = (float)pow((double)D, -0.5);
        g.rag = rag;                       // ragged: pair z multiplies rows off0[z].. of G by rows T0 + off1[z].. of G
        if (rag) g.B = Gh;
        Range r_stage("scores");
        Scope sc(prof, OG_STAGE_GEMM_F16X3);
        if (B > 1 || rag) rc = og_launch_gemm_f16x3(g, st);
        else { g.batch = 0; rc = og_launch_gemm_f16x3(g, st); }
        if (rc) return rc;
    }

    // ---- 5. Sinkhorn with dustbins -> scores (superglue.py:88-111) ----
    {
        Range r_stage("sinkhorn");
        Scope sc(prof, OG_STAGE_SINKHORN);
        // with matches requested, the scores kernel also leaves every row's max / argmax in the extraction's workspace
        RowBest rb{nullptr, nullptr, 0};
        if (outp->matches0) rb = og_matches_row_best(ws + W.match, B, m, n);
        if ((rc = og_launch_sinkhorn(Sb, W.lds, pk + L.dustbin, 0.f, B, m, n, s.sinkhorn_iters, s.sinkhorn_reg, outp->scores,
                                     ws + W.sink, st, rag, outp->matches0 ? &rb : nullptr))) return rc;
    }

    // ---- 6. mutual-NN matches (matching_module.py:174-187) ----
    if (outp->matches0) {
        Range r_stage("matches");
        Scope sc(prof, OG_STAGE_MATCHES);
        if ((rc = og_launch_matches(outp->scores, B, m, n, s.match_threshold, outp->matches0, outp->matching_scores0,
                                    outp->matches1, outp->matching_scores1, ws + W.match, st, rag, true))) return rc;
    }
    return 0;
}

}  // namespace

extern "C" int og_forward(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                          const og_outputs* outp, void* stream) {
    return forward_impl(shape, in, packed_dev, workspace_dev, outp, stream, nullptr);
}

extern "C" int og_forward_ragged(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh,
                                 const float* image1_wh, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                                 const og_outputs* outp, void* stream);

// include/openglue_amd.h: og_forward plus a copy of the residual stream at one stage boundary (per-stage parity tests)
extern "C" int og_forward_tap(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                              const og_outputs* outp, void* stream, int32_t tap, float* tap_x) {
    if (!shape || tap < 0 || tap > 2 * shape->num_stages || !tap_x || ((uintptr_t)tap_x & 15)) return OG_E_INVALID;
    return forward_impl(shape, in, packed_dev, workspace_dev, outp, stream, nullptr, nullptr, nullptr, nullptr, tap, tap_x);
}

// include/openglue_amd.h: the keypoint-encoder stage on its own (SURVEY.md 8b names it among the per-stage entries)
extern "C" int og_keypoint_encoder(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev, float* x_out,
                                   void* stream) {
    if (!shape || !x_out || ((uintptr_t)x_out & 15)) return OG_E_INVALID;
    return forward_impl(shape, in, packed_dev, workspace_dev, nullptr, stream, nullptr, nullptr, nullptr, nullptr, 0, x_out, true);
}

// include/openglue_amd.h: the raw score matrices of a batch of pairs, exact fp32 (the whole path forms them from (hi, lo) rows it already holds)
extern "C" int og_scores(const float* g0, const float* g1, int32_t batch, int32_t m, int32_t n, int32_t D, float* S, int64_t lds, void* stream) {
    if (!g0 || !g1 || !S || batch <= 0 || m <= 0 || n <= 0 || D <= 0) return OG_E_INVALID;
    if ((D & 3) || lds < n || (lds & 3)) return OG_E_SHAPE;
    return og_gemm_nt(g0, D, (int64_t)m * D, g1, D, (int64_t)n * D, S, lds, (int64_t)m * lds, m, n, D, batch, nullptr, 0, nullptr, 0, nullptr,
                      (float)pow((double)D, -0.5), stream);
}

namespace {
int build_ragged(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh, const float* image1_wh,
                 const og_inputs* in, RaggedDesc& rd, EncoderRagged& er0, EncoderRagged& er1) {
    if (!shape || !lens0 || !lens1 || !in) return OG_E_INVALID;
    if (shape->batch <= 0 || shape->batch > OG_MAX_RAGGED) return OG_E_SHAPE;
    rd.B = er0.B = er1.B = shape->batch;
    rd.off0[0] = rd.off1[0] = 0;
    rd.soff[0] = 0;
    for (int b = 0; b < rd.B; ++b) {
        if (lens0[b] <= 0 || lens1[b] <= 0 || lens0[b] > shape->m || lens1[b] > shape->n) return OG_E_SHAPE;
        rd.off0[b + 1] = rd.off0[b] + lens0[b];
        rd.off1[b + 1] = rd.off1[b] + lens1[b];
        rd.soff[b + 1] = rd.soff[b] + (int64_t)(lens0[b] + 1) * (lens1[b] + 1);
        // every pair is normalised with ITS OWN image size (superglue.py:35-41 runs per call = per pair at B = 1)
        er0.wm1[b] = (image0_wh ? image0_wh[2 * b] : in->image0_wh[0]) - 1.f;
        er0.hm1[b] = (image0_wh ? image0_wh[2 * b + 1] : in->image0_wh[1]) - 1.f;
        er1.wm1[b] = (image1_wh ? image1_wh[2 * b] : in->image1_wh[0]) - 1.f;
        er1.hm1[b] = (image1_wh ? image1_wh[2 * b + 1] : in->image1_wh[1]) - 1.f;
    }
    for (int b = 0; b <= rd.B; ++b) { er0.off[b] = rd.off0[b]; er1.off[b] = rd.off1[b]; }
    return 0;
}

int profiled(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev, const og_outputs* outp,
             void* stream, float* stage_ms, int32_t* stage_launches, const RaggedDesc* rd, const EncoderRagged* er0,
             const EncoderRagged* er1);
}  // namespace

extern "C" int og_forward_profiled(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                                   const og_outputs* outp, void* stream, float* stage_ms, int32_t* stage_launches) {
    return profiled(shape, in, packed_dev, workspace_dev, outp, stream, stage_ms, stage_launches, nullptr, nullptr, nullptr);
}

extern "C" int og_forward_ragged_profiled(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh,
                                          const float* image1_wh, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                                          const og_outputs* outp, void* stream, float* stage_ms, int32_t* stage_launches) {
    RaggedDesc rd;
    EncoderRagged er0, er1;
    if (int e = build_ragged(shape, lens0, lens1, image0_wh, image1_wh, in, rd, er0, er1)) return e;
    return profiled(shape, in, packed_dev, workspace_dev, outp, stream, stage_ms, stage_launches, &rd, &er0, &er1);
}

namespace {
int profiled(const og_shape* shape, const og_inputs* in, const void* packed_dev, void* workspace_dev, const og_outputs* outp,
             void* stream, float* stage_ms, int32_t* stage_launches, const RaggedDesc* rd, const EncoderRagged* er0,
             const EncoderRagged* er1) {
    if (!stage_ms || !stage_launches) return OG_E_INVALID;
    Profiler prof{(hipStream_t)stream, {}, {}};
    int rc = forward_impl(shape, in, packed_dev, workspace_dev, outp, stream, &prof, rd, er0, er1);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    for (int c = 0; c < OG_NUM_STAGES; ++c) { stage_ms[c] = 0.f; stage_launches[c] = 0; }
    for (size_t i = 0; i < prof.cls.size(); ++i) {
        float ms = 0.f;
        if (e == hipSuccess && rc == 0 && hipEventElapsedTime(&ms, prof.ev[2 * i], prof.ev[2 * i + 1]) == hipSuccess) {
            stage_ms[prof.cls[i]] += ms;
            stage_launches[prof.cls[i]] += 1;
        }
        (void)hipEventDestroy(prof.ev[2 * i]);
        (void)hipEventDestroy(prof.ev[2 * i + 1]);
    }
    if (rc) return rc;
    return e == hipSuccess ? 0 : (int)e;
}
}  // namespace

extern "C" int og_forward_ragged(const og_shape* shape, const int32_t* lens0, const int32_t* lens1, const float* image0_wh,
                                 const float* image1_wh, const og_inputs* in, const void* packed_dev, void* workspace_dev,
                                 const og_outputs* outp, void* stream) {
    RaggedDesc rd;
    EncoderRagged er0, er1;
    if (int e = build_ragged(shape, lens0, lens1, image0_wh, image1_wh, in, rd, er0, er1)) return e;
    return forward_impl(shape, in, packed_dev, workspace_dev, outp, stream, nullptr, &rd, &er0, &er1);
}

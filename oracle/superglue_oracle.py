"""CPU oracle for the OpenGlue SuperGlue hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module, and only as the checker / the reported CPU baseline.  Nothing under
openglue_amd/ imports it; the product path fails loudly when the HIP library is missing.

What it is: a functional, token-major ([B, n, C] rows = keypoints) restatement in plain
torch CPU ops of the algorithm the reference implements with nn.Modules in channel-first
layout.  Every function cites the reference lines it follows (paths relative to the
reference repo root).  It takes the reference's own state-dict (SURVEY.md §3.6 names), so
any checkpoint of the reference can be fed to it unchanged.

Pinning status: the reference ships NO tests / golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, generated in the build
container by importing /root/reference (tests/golden/make_golden.py, committed together with
the fixtures it wrote).  tests/test_oracle_golden.py checks the oracle against those
fixtures; tests/test_oracle_vs_reference.py re-checks against the live reference whenever
/root/reference is present.

`dtype=torch.float64` gives a higher-precision "truth" used to measure the fp32 noise floor.
`attn_operand_dtype=torch.float16` rounds the QK^T / PV operands the way the HIP attention
kernel does (MFMA f16 inputs, f32 accumulate) -- used only to budget the tolerance.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm1d default, models/utils.py:55


def _w(sd: Mapping[str, torch.Tensor], name: str, dtype) -> torch.Tensor:
    t = sd[name]
    t = (t if t.requires_grad else t.detach()).to("cpu", dtype)       # parameters under autograd (training fixtures) stay attached
    return t[:, :, 0] if t.dim() == 3 else t  # Conv1d k=1 weight [out, in, 1] -> [out, in]


def conv1x1(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    """nn.Conv1d(kernel_size=1) on channel-first == a linear map on token rows."""
    return F.linear(x, _w(sd, prefix + ".weight", x.dtype), _w(sd, prefix + ".bias", x.dtype))


def batchnorm_eval(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    """nn.BatchNorm1d in eval mode (running statistics), per channel = last dim here."""
    dt = x.dtype
    mean, var = _w(sd, prefix + ".running_mean", dt), _w(sd, prefix + ".running_var", dt)
    gamma, beta = _w(sd, prefix + ".weight", dt), _w(sd, prefix + ".bias", dt)
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


# training-mode restatement (tests only): when this is a dict, every FeedForwardNet uses batch statistics
# (feed_forward_train below) and records the updated running statistics in it -- see superglue_forward(train_stats=...)
_TRAIN_STATS = None


def feed_forward(x: torch.Tensor, sd, prefix: str, n_conv: int) -> torch.Tensor:
    """FeedForwardNet: [Conv1d, ReLU, BatchNorm1d] * (n_conv-1) + Conv1d, in THAT order
    (models/utils.py:48-58).  Sequential indices: conv 3i, relu 3i+1, bn 3i+2."""
    if _TRAIN_STATS is not None:
        y, stats = feed_forward_train(x, {**sd, **_TRAIN_STATS}, prefix, n_conv)      # the second call of a module continues from the first
        for k, v in stats.items():          # a module called twice per step (image 0, image 1) updates its statistics twice
            _TRAIN_STATS[k] = v
        return y
    for i in range(n_conv - 1):
        x = conv1x1(x, sd, f"{prefix}.{3 * i}")
        x = torch.relu(x)
        x = batchnorm_eval(x, sd, f"{prefix}.{3 * i + 2}")
    return conv1x1(x, sd, f"{prefix}.{3 * (n_conv - 1)}")


def feed_forward_siren(x: torch.Tensor, sd, prefix: str, n_conv: int) -> torch.Tensor:
    """FeedForwardNetSiren: [Conv1d, Sine] * (n_conv-1) + Conv1d, Sine(x) = sin(30 x), no BatchNorm
    (models/utils.py:23-45).  Sequential indices: conv 2i, sine 2i+1."""
    for i in range(n_conv - 1):
        x = torch.sin(30 * conv1x1(x, sd, f"{prefix}.{2 * i}"))
    return conv1x1(x, sd, f"{prefix}.{2 * (n_conv - 1)}")


def normalize_keypoints(kpts: torch.Tensor, width: float, height: float) -> torch.Tensor:
    """superglue.py:74-78: 2*k / [W-1, H-1] - 1."""
    return 2 * kpts / torch.tensor([width - 1, height - 1], dtype=kpts.dtype) - 1.0


def keypoint_encoder(kpts_n: torch.Tensor, side: torch.Tensor, sd, config) -> torch.Tensor:
    """MLPPositionalEncoding.forward, positional_encoding.py:16-19: cat([xy, side_info]) -> MLP."""
    n_conv = len(config["positional_encoding"]["hidden_layers_sizes"]) + 1
    ff = feed_forward_siren if config["positional_encoding"].get("encoder_name") == "FeedForwardNetSiren" else feed_forward
    return ff(torch.cat([kpts_n, side], dim=-1), sd, "positional_encoding.encoder", n_conv)


def softmax_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int,
                      operand_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """attention.py:8-19 with the head split of attention_gnn.py:24-26: channel c belongs to head
    c // d (contiguous blocks, `.view(B, H, d, N)`), A = softmax_keys(Q K^T / sqrt(d)), O = A V."""
    B, nq, D = q.shape
    d = D // num_heads
    qh = q.view(B, nq, num_heads, d).transpose(1, 2)            # [B,H,nq,d]
    kh = k.view(B, -1, num_heads, d).transpose(1, 2)
    vh = v.view(B, -1, num_heads, d).transpose(1, 2)
    if operand_dtype is not None:  # emulate MFMA f16 operands (scale folded into Q before rounding)
        qh = (qh * d ** -0.5).to(operand_dtype).to(q.dtype)
        kh = kh.to(operand_dtype).to(q.dtype)
        vh = vh.to(operand_dtype).to(q.dtype)
        logits = qh @ kh.transpose(-1, -2)
        mx = logits.amax(-1, keepdim=True)
        p = torch.exp(logits - mx)
        l = p.sum(-1, keepdim=True)
        o = (p.to(operand_dtype).to(q.dtype) @ vh) / l
    else:
        att = (qh @ kh.transpose(-1, -2)) * d ** -0.5
        o = att.softmax(dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, nq, D)


def linear_attention_elu(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int) -> torch.Tensor:
    """attention.py:22-40: q' = elu(q)+1+eps, k' = elu(k)+1+eps (eps = 1e-6), kv = K'^T V per head,
    out = (Q' kv) / (Q' sum_keys k').  No d^-1/2 scale."""
    B, nq, D = q.shape
    d = D // num_heads
    qh = F.elu(q.view(B, nq, num_heads, d).transpose(1, 2)) + 1 + 1e-6          # [B,H,nq,d]
    kh = F.elu(k.view(B, -1, num_heads, d).transpose(1, 2)) + 1 + 1e-6
    vh = v.view(B, -1, num_heads, d).transpose(1, 2)
    kv = kh.transpose(-1, -2) @ vh                                               # [B,H,d,d]
    o = (qh @ kv) / (qh @ kh.sum(2, keepdim=True).transpose(-1, -2))
    return o.transpose(1, 2).reshape(B, nq, D)


def favor_relu_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, projection: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """attention = 'favor_relu' (__init__.py:19-25: GeneralizedFavorAttention(embed_dim, ReLU, 2 * embed_dim features, eps = 1e-8)):
    randomized_kernel (attention.py:91-95)  phi(x) = relu(P (x * d^-1/4)) + eps  with d = x.size(2) = the head size, applied to
    q and k, then linear_attention (attention.py:29-40): out = phi(q) (phi(k)^T v) / (phi(q) . sum_keys phi(k)).
    The reference multiplies the [2D, D] buffer with [B, H, d, N] tensors, which only type-checks for d == D: num_heads == 1
    (a 4-head config raises inside torch.matmul).  q, k, v token-major [B, n, D]."""
    d = q.shape[-1]
    P = projection.to(q.dtype)
    fq = torch.relu((q * d ** -0.25) @ P.T) + eps                # [B, nq, F]
    fk = torch.relu((k * d ** -0.25) @ P.T) + eps                # [B, nk, F]
    kv = fk.transpose(-1, -2) @ v                                # [B, F, D]
    norm = fq @ fk.sum(1, keepdim=True).transpose(-1, -2)        # [B, nq, 1]
    return (fq @ kv) / norm


def message_passing(xq: torch.Tensor, xkv: torch.Tensor, sd, prefix: str, num_heads: int,
                    use_offset: bool, attn_operand_dtype=None, attention: str = "softmax") -> torch.Tensor:
    """ResidualAttentionMessagePropagation.forward, attention_gnn.py:43-55, with
    MultiheadAttention.forward :22-32 inlined:  q + fc(cat[q, out_proj(MHA(q, kv, kv))])."""
    q = conv1x1(xq, sd, prefix + ".mha.in_proj_q")
    k = conv1x1(xkv, sd, prefix + ".mha.in_proj_k")
    v = conv1x1(xkv, sd, prefix + ".mha.in_proj_v")
    if attention == "linear":
        att = linear_attention_elu(q, k, v, num_heads)
    elif attention == "favor_relu":
        if num_heads != 1:
            raise ValueError("favor_relu: the reference only runs with num_heads == 1")
        att = favor_relu_attention(q, k, v, _w(sd, prefix + ".mha.attention_func.projection_matrix", q.dtype))
    else:
        att = softmax_attention(q, k, v, num_heads, attn_operand_dtype)
    msg = conv1x1(att, sd, prefix + ".mha.out_proj")
    y = torch.cat([xq - msg, msg], dim=-1) if use_offset else torch.cat([xq, msg], dim=-1)
    return xq + feed_forward(y, sd, prefix + ".fc", 2)


def attentional_gnn(x0: torch.Tensor, x1: torch.Tensor, sd, config, attn_operand_dtype=None, taps: Optional[list] = None):
    """GraphAttentionNet.forward, attention_gnn.py:84-93.  Layer 2l = self (both images through the
    SAME module, :63-66), layer 2l+1 = cross (:74-77): image 0 first, then image 1 attends to the
    UPDATED image-0 descriptors."""
    g = config["attention_gnn"]
    H, off = g["num_heads"], g.get("use_offset", False)
    att = g.get("attention", "softmax")
    if att not in ("softmax", "linear", "favor_relu"):
        raise ValueError(f"oracle: attention {att} not restated")
    for l in range(g["num_stages"]):
        ps, pc = f"attention_gnn.layers.{2 * l}.module", f"attention_gnn.layers.{2 * l + 1}.module"
        x0 = message_passing(x0, x0, sd, ps, H, off, attn_operand_dtype, att)
        x1 = message_passing(x1, x1, sd, ps, H, off, attn_operand_dtype, att)
        if taps is not None: taps.append((x0, x1))         # after attention_gnn.layers[2l] (self)
        x0 = message_passing(x0, x1, sd, pc, H, off, attn_operand_dtype, att)
        x1 = message_passing(x1, x0, sd, pc, H, off, attn_operand_dtype, att)
        if taps is not None: taps.append((x0, x1))         # after attention_gnn.layers[2l + 1] (cross)
    return x0, x1


def log_sinkhorn(log_a: torch.Tensor, log_b: torch.Tensor, Mx: torch.Tensor, num_iters: int,
                 reg: float) -> torch.Tensor:
    """log_otp_solver, optimal_transport.py:20-28: u first (with v = 0), then v with the NEW u;
    returns M/reg + u + v (not multiplied back by reg)."""
    Mx = Mx / reg
    u, v = torch.zeros_like(log_a), torch.zeros_like(log_b)
    for _ in range(num_iters):
        u = log_a - torch.logsumexp(Mx + v[:, None, :], dim=2)
        v = log_b - torch.logsumexp(Mx + u[:, :, None], dim=1)
    return Mx + u[:, :, None] + v[:, None, :]


def matching_log_probs(S: torch.Tensor, dustbin: torch.Tensor, num_iters: int, reg: float) -> torch.Tensor:
    """SuperGlue.get_matching_probs, superglue.py:88-111: dustbin row/column, log-marginals with
    norm = -log(m+n), Sinkhorn, minus norm."""
    B, m, n = S.shape
    S_aug = torch.empty(B, m + 1, n + 1, dtype=S.dtype)
    S_aug[:, :m, :n] = S
    S_aug[:, m, :] = dustbin
    S_aug[:, :, n] = dustbin
    norm = -math.log(m + n)
    log_a = torch.full((B, m + 1), norm, dtype=S.dtype)
    log_b = torch.full((B, n + 1), norm, dtype=S.dtype)
    log_a[:, -1] += math.log(n)
    log_b[:, -1] += math.log(m)
    return log_sinkhorn(log_a, log_b, S_aug, num_iters, reg) - norm


def _image_wh(data: Mapping, idx: int) -> Tuple[float, float]:
    """superglue.py:35-38: image tensor -> size()[-2:] = (H, W); else image{idx}_size = [W, H]."""
    if "image0" in data and "image1" in data:
        h, w = data[f"image{idx}"].shape[-2:]
        return float(w), float(h)
    w, h = data[f"image{idx}_size"][:2]
    return float(w), float(h)


def superglue_forward(sd: Mapping[str, torch.Tensor], config: Mapping, data: Mapping,
                      dtype: torch.dtype = torch.float32, attn_operand_dtype=None,
                      return_intermediates: bool = False, train_stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """SuperGlue.forward, superglue.py:29-72.  Returns the reference's dict: 'context_descriptors{0,1}' CHANNEL-FIRST [B, D, n]
    and 'scores' [B, m+1, n+1].  Eval mode by default; with `train_stats` (a dict) the MLPs run in TRAINING mode (batch-statistics
    BatchNorm), the updated running statistics are returned in it, and tensors that require grad stay attached to autograd."""
    global _TRAIN_STATS
    if train_stats is not None:
        _TRAIN_STATS = train_stats
        try:
            return superglue_forward(sd, config, data, dtype, attn_operand_dtype, return_intermediates, None)
        finally:
            _TRAIN_STATS = None
    cvt = lambda t: (t if t.requires_grad else t.detach()).to("cpu", dtype)
    k0, k1 = cvt(data["keypoints0"]), cvt(data["keypoints1"])
    d0, d1 = cvt(data["local_descriptors0"]), cvt(data["local_descriptors1"])
    s0, s1 = cvt(data["side_info0"]), cvt(data["side_info1"])
    pe0 = keypoint_encoder(normalize_keypoints(k0, *_image_wh(data, 0)), s0, sd, config)
    pe1 = keypoint_encoder(normalize_keypoints(k1, *_image_wh(data, 1)), s1, sd, config)
    if config.get("no_descriptors", False):                      # superglue.py:45-49
        x0, x1 = pe0, pe1
    else:                                                        # :52-55
        x0, x1 = d0 + pe0, d1 + pe1
    inter = {"x0_in": x0, "x1_in": x1, "pe0": pe0, "pe1": pe1}
    taps = [] if return_intermediates else None
    x0, x1 = attentional_gnn(x0, x1, sd, config, attn_operand_dtype, taps)
    if taps is not None: inter["layer_taps"] = taps
    inter.update(x0_gnn=x0, x1_gnn=x1)
    g0, g1 = conv1x1(x0, sd, "linear_proj"), conv1x1(x1, sd, "linear_proj")   # :58
    if config.get("residual", False):                            # :59-62, per-channel alpha
        alpha = torch.sigmoid(_w(sd, "mix_coefs", dtype)[:, 0])
        g0 = alpha * g0 + (1.0 - alpha) * d0
        g1 = alpha * g1 + (1.0 - alpha) * d1
    S = (g0 @ g1.transpose(1, 2)) * config["descriptor_dim"] ** -0.5           # :64, :81-86
    scores = matching_log_probs(S, _w(sd, "dustbin_score", dtype),
                                config["otp"]["num_iters"], config["otp"]["reg"])
    out = {
        "context_descriptors0": g0.transpose(1, 2).contiguous(),
        "context_descriptors1": g1.transpose(1, 2).contiguous(),
        "scores": scores,
    }
    if return_intermediates:
        inter["S"] = S
        out["_intermediates"] = inter
    return out


def extract_matches(scores: torch.Tensor, match_threshold: float) -> Dict[str, torch.Tensor]:
    """models/matching_module.py:174-187 (matches0 / matching_scores0) and inference.py:176-190
    (matches1 / matching_scores1): row/col max of scores[:, :-1, :-1], mutual check through
    gather, exp of the max, threshold, -1 fill.  torch.max on CPU returns the FIRST maximal index."""
    inner = scores[:, :-1, :-1]
    max0, max1 = inner.max(2), inner.max(1)
    idx0, idx1 = max0.indices, max1.indices
    ar0 = torch.arange(idx0.shape[1], dtype=idx0.dtype)[None]          # utils/misc.py:116-117
    ar1 = torch.arange(idx1.shape[1], dtype=idx1.dtype)[None]
    mutual0 = ar0 == idx1.gather(1, idx0)
    mutual1 = ar1 == idx0.gather(1, idx1)
    zero = scores.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values.exp(), zero)
    ms1 = torch.where(mutual1, ms0.gather(1, idx1), zero)
    valid0 = mutual0 & (ms0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, idx1)
    return {
        "matches0": torch.where(valid0, idx0, idx0.new_tensor(-1)),
        "matching_scores0": ms0,
        "matches1": torch.where(valid1, idx1, idx1.new_tensor(-1)),
        "matching_scores1": ms1,
        "_row_argmax": idx0, "_col_argmax": idx1,
        "_row_max": max0.values, "_col_max": max1.values,
    }


def match_pairs(sd, config, data, match_threshold: float = 0.2, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """MatchingTrainingModule.forward from the `scores` onwards (matching_module.py:171-187)."""
    out = superglue_forward(sd, config, data, dtype=dtype)
    out.update(extract_matches(out["scores"], match_threshold))
    return out


def ambiguous_rows(scores64: torch.Tensor, gap: float = 1e-4) -> Tuple[torch.Tensor, torch.Tensor]:
    """Near-tie detector for the index-parity tests (SURVEY.md §7 'index parity is ill-posed on
    near-ties'): rows / columns of the inner score block whose top-1 / top-2 gap (in the float64
    oracle) is below `gap`.  Returns boolean masks [B, m], [B, n]."""
    inner = scores64[:, :-1, :-1]
    t0 = inner.topk(2, dim=2).values
    t1 = inner.topk(2, dim=1).values
    return (t0[..., 0] - t0[..., 1]) < gap, (t1[:, 0] - t1[:, 1]) < gap


# ---------------------------------------------------------------------------------------------------
# The steps either side of the matcher.  kornia is NOT installed here and not vendored
# in the reference, so get_laf_scale / get_laf_center are restated from kornia's published source
# (kornia/feature/laf.py, requirements.txt:5 `kornia>=0.6.1`): parity for these two helpers is pinned to the
# published algorithm only, not to an execution of kornia.
def get_laf_scale(lafs: torch.Tensor) -> torch.Tensor:
    """kornia.feature.laf.get_laf_scale: sqrt(|det(A) + 1e-10|) of the 2x2 affine part, shape [B, N, 1, 1]."""
    det = lafs[..., 0:1, 0:1] * lafs[..., 1:2, 1:2] - lafs[..., 1:2, 0:1] * lafs[..., 0:1, 1:2] + 1e-10
    return det.abs().sqrt()


def laf_side_info(lafs: torch.Tensor, method: str) -> torch.Tensor:
    """models/laf_converter.py:22-128: 'none' | 'scale' | 'rotation' | 'scale_rotation' | 'affine'."""
    B, N = lafs.shape[:2]
    scale = get_laf_scale(lafs).squeeze(-1)                                   # [B, N, 1]
    log_scale = torch.log(scale)                                              # LAF2LogScale (:22-36) -- [B, N, 1]
    rot = torch.flip(lafs[..., 0, :-1], dims=(-1,)) / scale                   # LAF2SinCosOrientation (:39-55)
    aff = torch.flatten(lafs[..., :-1], start_dim=2) / scale                  # LAF2AffineGeom (:58-72)
    parts = {"none": [], "scale": [log_scale], "rotation": [rot], "scale_rotation": [log_scale, rot],
             "affine": [log_scale, aff]}
    if method.lower() not in parts:
        raise NameError("Unexpected name for the method: {}".format(method))
    p = parts[method.lower()]
    return torch.cat(p, dim=-1) if p else lafs.new_empty(B, N, 0)


def prepare_features_output(lafs, responses, desc, method="none", permute_desc=False, log_response=False):
    """models/features/utils.py:54-65."""
    kpts = lafs[:, :, :, -1]
    responses = responses.unsqueeze(-1)
    if log_response:
        responses = (responses + 0.1).log()
    return {"keypoints": kpts, "side_info": torch.cat([responses, laf_side_info(lafs, method)], dim=-1),
            "local_descriptors": desc.permute(0, 2, 1) if permute_desc else desc}


def compact_matches(matches0: torch.Tensor, matching_scores0: torch.Tensor, lafs0=None, lafs1=None):
    """inference.py:192-209: boolean-mask compaction of the valid matches (row-major over pair, keypoint)."""
    B, M = matches0.shape
    mask0 = matches0 != -1
    arange0 = torch.arange(M)[None].expand(B, -1)
    batch_idxs0 = torch.arange(B)[:, None].expand(-1, M)[mask0]
    idx0, idx1 = arange0[mask0], matches0[mask0]
    out = {"original_matching_idxs": torch.stack([idx0, idx1], dim=-1), "batch_indexes": batch_idxs0,
           "confidence": matching_scores0[mask0]}
    if lafs0 is not None:
        ml0, ml1 = lafs0[batch_idxs0, idx0][None], lafs1[batch_idxs0, idx1][None]
        out.update(lafs0=ml0, lafs1=ml1, keypoints0=ml0[0][..., 2], keypoints1=ml1[0][..., 2])   # kornia get_laf_center
    return out


def nll_criterion(scores: torch.Tensor, gt_matches0: torch.Tensor, gt_matches1: torch.Tensor) -> torch.Tensor:
    """The 'loss' entry of utils/losses.py:7-53 (margin=None): per pair, the mean negative log-assignment of the matched
    keypoints (gt >= 0), plus half the mean over the unmatched (gt == -1) keypoints of image 0 at the dustbin column and of
    image 1 at the dustbin row; IGNORE labels (-2) take no part; divided by the batch size."""
    def mean_w(batch_idx):
        _, inv, counts = torch.unique_consecutive(batch_idx, return_inverse=True, return_counts=True)
        return (1 / counts)[inv]
    b, i0 = torch.where(gt_matches0 >= 0)
    matched = (-scores[b, i0, gt_matches0[b, i0]] * mean_w(b)).sum()
    b, i0 = torch.where(gt_matches0 == -1)
    un0 = (-scores[b, i0, -1] * mean_w(b)).sum()
    b, i1 = torch.where(gt_matches1 == -1)
    un1 = (-scores[b, -1, i1] * mean_w(b)).sum()
    return (matched + 0.5 * (un0 + un1)) / scores.size(0)


def pairwise_cosine_dist(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """utils/misc.py:106-113: half of the cosine distance, (1 - cos) / 2 = |x1/|x1| - x2/|x2||^2 / 4, for every pair of rows."""
    a = torch.nn.functional.normalize(x1, dim=-1)
    b = torch.nn.functional.normalize(x2, dim=-1)
    return 0.25 * (a[..., :, None, :] - b[..., None, :, :]).pow(2).sum(-1)


def metric_criterion(ctx0: torch.Tensor, ctx1: torch.Tensor, gt_matches0: torch.Tensor, gt_matches1: torch.Tensor, margin: float) -> torch.Tensor:
    """The 'metric_loss' entry of utils/losses.py:7-93 for margin != None, on the channel-first context descriptors [B, D, n] the
    model returns: (a) matched keypoints: triplet loss against the closest NON-matching descriptor of the other image, both ways
    (losses.py:55-74; the negatives come from a detached copy of the distances with the positives masked by +inf);
    (b) unmatched keypoints of either image: hinge on the distance to their closest descriptor in the other image (losses.py:77-93);
    each term averaged per pair like the NLL, the total divided by the batch size."""
    dist = pairwise_cosine_dist(ctx0.transpose(2, 1).contiguous(), ctx1.transpose(2, 1).contiguous())        # [B, m, n]

    def mean_w(batch_idx):
        _, inv, counts = torch.unique_consecutive(batch_idx, return_inverse=True, return_counts=True)
        return (1 / counts)[inv]
    zero = torch.zeros((), dtype=dist.dtype, device=dist.device)
    b, i0 = torch.where(gt_matches0 >= 0)
    i1 = gt_matches0[b, i0]
    w = mean_w(b)
    d_ap = dist[b, i0, i1]
    dd = dist.detach().clone()
    dd[b, i0, i1] = float("inf")
    i0_closest_to_1 = torch.argmin(dd, dim=1)            # [B, n]
    i1_closest_to_0 = torch.argmin(dd, dim=2)            # [B, m]
    d_an0 = dist[b, i0, i1_closest_to_0[b, i0]]
    d_an1 = dist[b, i0_closest_to_1[b, i1], i1]
    matched = (torch.maximum(d_ap - d_an0 + margin, zero) * w).sum() + (torch.maximum(d_ap - d_an1 + margin, zero) * w).sum()
    b, i0 = torch.where(gt_matches0 == -1)
    un0 = (torch.maximum(-dist[b, i0, torch.argmin(dist, dim=2)[b, i0]] + margin, zero) * mean_w(b)).sum()
    b, i1 = torch.where(gt_matches1 == -1)
    un1 = (torch.maximum(-dist[b, torch.argmin(dist, dim=1)[b, i1], i1] + margin, zero) * mean_w(b)).sum()
    return (matched + un0 + un1) / ctx0.size(0)


def batchnorm_train(x: torch.Tensor, weight, bias, running_mean, running_var, momentum: float = 0.1, eps: float = 1e-5):
    """nn.BatchNorm1d in training mode (reference models/utils.py:55 inside FeedForwardNet; semantics of
    torch.nn.functional.batch_norm(training=True), restated with plain tensor ops) on token-major x [..., C] like the rest of
    this file (the reference's [B, C, N] tensor transposed): batch statistics over every token, biased variance for the
    normalisation, unbiased for the running estimate.  Returns (y, new_running_mean, new_running_var)."""
    xt = x.reshape(-1, x.shape[-1])
    n = xt.shape[0]
    mean = xt.mean(dim=0)
    var = ((xt - mean) ** 2).mean(dim=0)
    y = (x - mean) / torch.sqrt(var + eps)
    if weight is not None:
        y = y * weight
    if bias is not None:
        y = y + bias
    new_rm = (1 - momentum) * running_mean + momentum * mean
    new_rv = (1 - momentum) * running_var + momentum * var * (n / max(n - 1, 1))
    return y, new_rm, new_rv


def feed_forward_train(x: torch.Tensor, sd, prefix: str, n_conv: int, momentum: float = 0.1):
    """FeedForwardNet (models/utils.py:48-58) in TRAINING mode on token-major x [..., C]: (Conv1d -> ReLU -> BatchNorm1d[batch
    stats]) x (n_conv-1) -> Conv1d.  `prefix` like feed_forward ("" for a bare nn.Sequential state dict).  Returns
    (y, {running-stat name: new value}); `sd` is not modified."""
    dtype = x.dtype
    dot = prefix + "." if prefix else ""
    new_stats = {}
    for i in range(n_conv - 1):
        x = torch.relu(conv1x1(x, sd, f"{dot}{3 * i}"))
        bn = f"{dot}{3 * i + 2}"
        x, rm, rv = batchnorm_train(x, _w(sd, bn + ".weight", dtype), _w(sd, bn + ".bias", dtype), _w(sd, bn + ".running_mean", dtype),
                                    _w(sd, bn + ".running_var", dtype), momentum)
        new_stats[bn + ".running_mean"], new_stats[bn + ".running_var"] = rm, rv
    return conv1x1(x, sd, f"{dot}{3 * (n_conv - 1)}"), new_stats

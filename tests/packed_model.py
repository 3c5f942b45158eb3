"""Test helper: run the hot path on the CPU from the PACKED weight blob (og_pack_weights output),
following the exact launch sequence of og_forward (openglue_amd/csrc/api.hip) with torch CPU ops.
Validates the packing algebra (BatchNorm folds, out_proj folded into fc.0, q pre-scale, zero padding)
against the oracle without a GPU.  Test infrastructure only."""
import ctypes as C
import math

import torch

from openglue_amd import _lib
from oracle import superglue_oracle as orc


def layout_of(model):
    lib = _lib.load()
    L = _lib.og_packed_layout_t()
    shape = model._shape(1, 1, 1)
    _lib.check(lib.og_packed_layout(C.byref(shape), C.byref(L)), "og_packed_layout")
    return L


def forward_from_packed(model, data, dtype=torch.float64):
    L = layout_of(model)
    raw = model.pack_host()
    blob = torch.from_numpy(raw).to(dtype)
    half = torch.from_numpy(raw.view("float16").copy()).to(dtype)       # same bytes viewed as f16 (2 per float slot)

    def planes(off, r, c, inv_off):     # hl32 rows (gemm_f16x3.hip): [r][c/32][hi 32 | lo 32] of S * w, 1 / S stored at inv_off
        g = half[2 * off:2 * off + 2 * r * c].view(r, c // 32, 2, 32)
        return (g[:, :, 0] + g[:, :, 1]).reshape(r, c) * blob[inv_off]

    D, H, s = model.descriptor_dim, model.num_heads, model.side_info_size
    mat = lambda off, r, c: blob[off:off + r * c].view(r, c)
    vec = lambda off, r: blob[off:off + r]

    def encode(k, side, wh):
        kn = 2 * k.to(dtype) / torch.tensor([wh[0] - 1, wh[1] - 1], dtype=dtype) - 1
        x = torch.zeros(*k.shape[:-1], 32, dtype=dtype)
        x[..., :2] = kn
        x[..., 2:2 + s] = side.to(dtype)
        for i in range(L.n_enc):
            W, b = mat(L.enc_w[i], L.enc_out[i], L.enc_k[i]), vec(L.enc_b[i], L.enc_out[i])
            x = x @ W.T + b
            if i + 1 < L.n_enc:
                x = torch.sin(30 * x) if model.siren else torch.relu(x)
        return x

    d0, d1 = data["local_descriptors0"].to(dtype), data["local_descriptors1"].to(dtype)
    x0 = encode(data["keypoints0"], data["side_info0"], orc._image_wh(data, 0))
    x1 = encode(data["keypoints1"], data["side_info1"], orc._image_wh(data, 1))
    if not model.no_descriptors:
        x0, x1 = x0 + d0, x1 + d1

    def attn(q, k, v):      # q pre-scaled by d^-1/2 * log2(e): softmax in base 2
        if model.linear_attention:
            return orc.linear_attention_elu(q, k, v, H)
        q = q * math.log(2.0)
        B, nq, _ = q.shape
        d = D // H
        qh = q.view(B, nq, H, d).transpose(1, 2)
        kh = k.view(B, -1, H, d).transpose(1, 2)
        vh = v.view(B, -1, H, d).transpose(1, 2)
        o = (qh @ kh.transpose(-1, -2)).softmax(-1) @ vh
        return o.transpose(1, 2).reshape(B, nq, D)

    def prop(l, xq, xkv):
        base = L.layer0 + l * L.layer_stride
        if getattr(model, "favor_relu", False):
            # favor_relu (api.hip): rows [0, 2D) / [2D, 4D) = d^-1/4 P folded into in_proj_q / in_proj_k (ReLU in the GEMM epilogue,
            # + eps where the features are read), rows [4D, 5D) = in_proj_v; then linear attention over the 2D features
            Wqkv, bqkv = planes(base + L.o_wqkv, 5 * D, D, base + L.o_scale), vec(base + L.o_bqkv, 5 * D)
            fq = torch.relu(xq @ Wqkv[:2 * D].T + bqkv[:2 * D]) + 1e-8
            fk = torch.relu(xkv @ Wqkv[2 * D:4 * D].T + bqkv[2 * D:4 * D]) + 1e-8
            v = xkv @ Wqkv[4 * D:].T + bqkv[4 * D:]
            o = (fq @ (fk.transpose(-1, -2) @ v)) / (fq @ fk.sum(1, keepdim=True).transpose(-1, -2))
        else:
            Wqkv, bqkv = planes(base + L.o_wqkv, 3 * D, D, base + L.o_scale), vec(base + L.o_bqkv, 3 * D)
            q = xq @ Wqkv[:D].T + bqkv[:D]
            kv = xkv @ Wqkv[D:].T + bqkv[D:]
            o = attn(q, kv[..., :D], kv[..., D:])
        h = torch.relu(torch.cat([xq, o], -1) @ planes(base + L.o_w0, 2 * D, 2 * D, base + L.o_scale + 1).T + vec(base + L.o_b0, 2 * D))
        return xq + h @ planes(base + L.o_w3, D, 2 * D, base + L.o_scale + 2).T + vec(base + L.o_b3, D)

    for l in range(model.num_stages):
        x0, x1 = prop(2 * l, x0, x0), prop(2 * l, x1, x1)
        x0 = prop(2 * l + 1, x0, x1)
        x1 = prop(2 * l + 1, x1, x0)
    Wp, bp = planes(L.wp, D, D, L.scales), vec(L.bp, D)
    g0, g1 = x0 @ Wp.T + bp, x1 @ Wp.T + bp
    if model.residual:
        a = vec(L.alpha, D)
        g0, g1 = a * g0 + (1 - a) * d0, a * g1 + (1 - a) * d1
    S = g0 @ g1.transpose(1, 2) * D ** -0.5
    scores = orc.matching_log_probs(S, blob[L.dustbin], model.config["otp"]["num_iters"], model.config["otp"]["reg"])
    return {"scores": scores, "context_descriptors0": g0.transpose(1, 2), "context_descriptors1": g1.transpose(1, 2)}

"""Exploration (CPU, test infrastructure -- not collected by pytest, never on the product path): what would the parity budget look
like if the two CORRECTION passes of the split-f16 products (Ah.Bl + Al.Bh) ran on the 8-bit matrix path (twice the f16 MFMA rate,
DESIGN.md §9) instead of in f16?  Every GNN conv, Q K^T and P V of the oracle is replaced by

    a.b  ~=  Ah.Bh  +  q8(Ah).q8(Bl) + q8(Al).q8(Bh),        x = hi + lo, hi = f16(x), lo = f16(x - hi)

with q8 = identity (the kernels as built), MX-style fp8 e4m3 (one power-of-two scale per 32 elements along k), int8 with one scale
per row, or "drop" (plain f16 operands), everything accumulated in float64, and the log-scores are compared with the float64 oracle on the golden cases.

    python tests/emulate_cross_term_precision.py [case ...]
    python tests/emulate_cross_term_precision.py --attention-only [case ...]     # round 4: only the attention kernel's cross products in 8 bits
    python tests/emulate_cross_term_precision.py --mx [case ...]                 # round 5: the HARDWARE block-scaled fp8 MFMA
        (v_mfma_scale_f32_32x32x64_f8f6f4: e4m3 elements, one E8M0 scale per 32 elements along k applied by the matrix pipe itself, fp32 accumulation into
        the same accumulator as the f16 hi.hi product) for the two cross products of Q K^T K-concatenated ([Qh | Ql] . [Kl | Kh]^T), P V left at f16 x 3
        ("mxqk"), for the two cross products of P V only ("mxpv": P is made in the kernel, V would need fp8 copies; "mxpvc": the same with CONSTANT block scales 2^0 / 2^-11 -- P <= 1 and V is O(10), so no block maxima are needed), and for both contractions ("mxboth").  Kill criterion: worst case over all fixtures + `trained` <= 5e-4.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import superglue_oracle as orc          # noqa: E402
from tests.util import load_case                    # noqa: E402

MODE = "exact"


def split(x):
    hi = x.to(torch.float16).to(torch.float64)
    lo = (x - hi).to(torch.float16).to(torch.float64)
    return hi, lo


def q8(x):
    """MX fp8 e4m3: blocks of 32 along the last axis share a power-of-two scale that brings the block maximum under 448."""
    if MODE == "exact":
        return x
    if MODE == "drop":
        return torch.zeros_like(x)
    if MODE == "int8":      # v_mfma_i32_*_i8 has no block scales: ONE scale per row (the whole contraction), 7 bits + sign
        scale = x.abs().amax(-1, keepdim=True).clamp_min(1e-300) / 127.0
        return torch.round(x / scale) * scale
    K = x.shape[-1]
    pad = (-K) % 32
    xp = torch.nn.functional.pad(x, (0, pad)) if pad else x
    b = xp.reshape(*xp.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (b / scale).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * scale
    return q.reshape(xp.shape)[..., :K]


def mm(a, b):          # a [..., M, K] @ b [..., N, K]^T with the operand handling under test
    ah, al = split(a)
    bh, bl = split(b)
    out = ah @ bh.transpose(-1, -2)
    if MODE != "drop":
        out = out + q8(ah) @ q8(bl).transpose(-1, -2) + q8(al) @ q8(bh).transpose(-1, -2)
    return out


def conv1x1(x, sd, prefix):
    W, b = orc._w(sd, prefix + ".weight", x.dtype), orc._w(sd, prefix + ".bias", x.dtype)
    if not prefix.startswith("attention_gnn"):          # encoder / final projection: exact-fp32 MFMA or not under test here
        return x @ W.T + b
    if ATTN_ONLY:                                       # the convs as built: all three f16 passes
        xh, xl = split(x); wh, wl = split(W * 256.0)
        return (xh @ wh.T + xh @ wl.T + xl @ wh.T) / 256.0 + b
    return mm(x, W * 256.0) / 256.0 + b                 # weights are stored as hi/lo of 256 w (og_pack_weights)


ATTN_ONLY = False     # round 4: quantise the correction products of the ATTENTION kernel only, the convs stay split-f16 x3


def q_rows(x, mode):
    """8-bit operand of one MFMA accumulation: ONE scale per row over the whole contraction it takes part in (the integer / fp8 matrix
    instructions have no per-element scales: a scale that varies along k cannot be folded back after the sum)."""
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    if mode == "int8":
        scale = amax / 127.0
        return torch.round(x / scale) * scale
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))       # fp8 e4m3 with a power-of-two row scale
    return (x / scale).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * scale


def mx(x):
    """e4m3 elements with one power-of-two scale per 32 elements along the contraction (the last axis), as the block-scaled MFMA applies them."""
    global MODE
    keep, MODE = MODE, "fp8"
    try:
        return q8(x)
    finally:
        MODE = keep


def mm_attn(a, b, tile=None):
    """a [..., M, K] @ b [..., N, K]^T as the flash kernel would run it with 8-bit correction products: the hi.hi product in f16, the two
    cross products with 8-bit operands; tile = the key-tile length of P V (scales per (row, key tile): every tile is its own integer
    accumulation, converted and added to the fp32 accumulator), None = one accumulation over the whole K (Q K^T: K = head size)."""
    ah, al = split(a)
    bh, bl = split(b)
    out = ah @ bh.transpose(-1, -2)
    if MODE == "drop":
        return out
    if MODE == "exact" or (MODE == "mxqk" and tile is not None) or (MODE == "mxpv" and tile is None):
        return out + ah @ bl.transpose(-1, -2) + al @ bh.transpose(-1, -2)
    if MODE == "mxpvc" and tile is None:
        return out + ah @ bl.transpose(-1, -2) + al @ bh.transpose(-1, -2)
    if MODE == "mxpvc":                          # P V cross products with CONSTANT block scales: 2^0 for the hi parts, 2^-11 for the lo parts (no block maxima)
        f8 = lambda x, sc: (x / sc).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * sc
        return out + f8(ah, 1.0) @ f8(bl, 2.0 ** -11).transpose(-1, -2) + f8(al, 2.0 ** -11) @ f8(bh, 1.0).transpose(-1, -2)
    if MODE in ("mxqk", "mxpv", "mxboth"):       # block scales vary along k: no per-tile accumulation needed, the pipe applies them
        return out + mx(ah) @ mx(bl).transpose(-1, -2) + mx(al) @ mx(bh).transpose(-1, -2)
    if tile is None:
        return out + q_rows(ah, MODE) @ q_rows(bl, MODE).transpose(-1, -2) + q_rows(al, MODE) @ q_rows(bh, MODE).transpose(-1, -2)
    K = a.shape[-1]
    for k0 in range(0, K, tile):
        sl = slice(k0, min(K, k0 + tile))
        out = out + q_rows(ah[..., sl], MODE) @ q_rows(bl[..., sl], MODE).transpose(-1, -2) + q_rows(al[..., sl], MODE) @ q_rows(bh[..., sl], MODE).transpose(-1, -2)
    return out


def softmax_attention(q, k, v, num_heads, operand_dtype=None):
    B, nq, D = q.shape
    d = D // num_heads
    qh = q.view(B, nq, num_heads, d).transpose(1, 2) * d ** -0.5
    kh = k.view(B, -1, num_heads, d).transpose(1, 2)
    vh = v.view(B, -1, num_heads, d).transpose(1, 2)
    if ATTN_ONLY:
        logits = mm_attn(qh, kh)
        p = torch.exp(logits - logits.amax(-1, keepdim=True))
        o = mm_attn(p, vh.transpose(-1, -2), tile=64) / p.sum(-1, keepdim=True)
        return o.transpose(1, 2).reshape(B, nq, D)
    logits = mm(qh, kh)
    p = torch.exp(logits - logits.amax(-1, keepdim=True))
    o = mm(p, vh.transpose(-1, -2)) / p.sum(-1, keepdim=True)
    return o.transpose(1, 2).reshape(B, nq, D)


def main():
    global MODE, ATTN_ONLY
    args = sys.argv[1:]
    modes = ("exact", "fp8", "int8", "drop")
    if args and args[0] == "--mx":
        ATTN_ONLY = True
        modes = ("exact", "mxqk", "mxpv", "mxpvc", "mxboth")
        args = args[1:]
        print("block-scaled fp8 (MX e4m3, E8M0 scale per 32 along k) for the cross products of the ATTENTION kernel: mxqk = Q K^T only (P V f16 x 3), mxboth = both; "
              "kill criterion: worst case <= 5e-4")
    elif args and args[0] == "--attention-only":
        ATTN_ONLY = True
        args = args[1:]
        print("correction products of the ATTENTION kernel only (Qh.Kl + Ql.Kh per head, Ph.Vl + Pl.Vh per 64-key tile), convs split-f16 x3; kill criterion: flags <= 5e-4")
    cases = args or ["c1", "flags", "mid"]
    torch.set_num_threads(os.cpu_count() or 8)
    for name in cases:
        if name.startswith("trained"):       # tests/golden/trained.npz: dead BatchNorm channels, large weights; unit-norm ("unit") or x4 descriptors
            from openglue_amd import synthetic as syn
            cfg = syn.make_config(descriptor_dim=256, num_stages=2, num_heads=4, num_iters=20, side_info_size=1)
            sd = syn.make_trained_like_state_dict(cfg, seed=0)
            data = syn.make_batch(2, 140, 120, 256, 1, seed=31, desc_scale=4.0 if name.endswith("x4") else 1.0)
        else:
            z, cfg, sd, data = load_case(name)
        with torch.no_grad():
            ref = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)["scores"]
            keep = orc.conv1x1, orc.softmax_attention
            orc.conv1x1, orc.softmax_attention = conv1x1, softmax_attention
            try:
                for MODE in modes:
                    got = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)["scores"]
                    err = (got - ref).abs().max().item()
                    m_ref = orc.extract_matches(ref.float(), 0.2)["matches0"]
                    m_got = orc.extract_matches(got.float(), 0.2)["matches0"]
                    print(f"{name:6s} cross terms {MODE:5s}: max |scores - float64 oracle| = {err:.2e}; matches0 differ on {int((m_ref != m_got).sum())} rows")
            finally:
                orc.conv1x1, orc.softmax_attention = keep


if __name__ == "__main__":
    main()

"""Exploration (CPU, test infrastructure -- not collected by pytest, never on the product path): which SINGLE cross term of the
split-f16 products can be dropped inside the parity bar?    Every split product of the path is

    a.b ~= Ah.Bh + Ah.Bl + Al.Bh          x = hi + lo, hi = f16(x), lo = f16(x - hi)

and the attention kernel spends six MFMA passes per tile: QK^T = Qh.Kh + Qh.Kl + Ql.Kh, PV = Ph.Vh + Ph.Vl + Pl.Vh.  Here each
correction term (and pairs of them) is removed from the float64 oracle one at a time -- attention terms with the GNN convs exact,
and the two conv terms (Xh.Wl, Xl.Wh) with attention exact -- and the log-scores are compared with the unmodified float64 oracle
on the golden cases.  P is the unnormalised exp(s - rowmax) in [0, 1] as in the kernel.

    python tests/emulate_term_ablation.py [case ...]          (default: c1 mid flags c2)
"""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import superglue_oracle as orc          # noqa: E402
from tests.util import load_case                    # noqa: E402

DROP = set()       # subset of {"QhKl", "QlKh", "PhVl", "PlVh", "XhWl", "XlWh"}


def split(x):
    hi = x.to(torch.float16).to(torch.float64)
    lo = (x - hi).to(torch.float16).to(torch.float64)
    return hi, lo


def mm(a, b, name_hl, name_lh):          # a [..., M, K] @ b [..., N, K]^T
    ah, al = split(a)
    bh, bl = split(b)
    out = ah @ bh.transpose(-1, -2)
    if name_hl not in DROP: out = out + ah @ bl.transpose(-1, -2)
    if name_lh not in DROP: out = out + al @ bh.transpose(-1, -2)
    return out


def conv1x1(x, sd, prefix):
    W, b = orc._w(sd, prefix + ".weight", x.dtype), orc._w(sd, prefix + ".bias", x.dtype)
    if not prefix.startswith("attention_gnn"):
        return x @ W.T + b
    return mm(x, W * 256.0, "XhWl", "XlWh") / 256.0 + b


def softmax_attention(q, k, v, num_heads, operand_dtype=None):
    B, nq, D = q.shape
    d = D // num_heads
    qh = q.view(B, nq, num_heads, d).transpose(1, 2) * d ** -0.5
    kh = k.view(B, -1, num_heads, d).transpose(1, 2)
    vh = v.view(B, -1, num_heads, d).transpose(1, 2)
    logits = mm(qh, kh, "QhKl", "QlKh")
    p = torch.exp(logits - logits.amax(-1, keepdim=True))
    o = mm(p, vh.transpose(-1, -2), "PhVl", "PlVh") / p.sum(-1, keepdim=True)
    return o.transpose(1, 2).reshape(B, nq, D)


def main():
    global DROP
    cases = sys.argv[1:] or ["c1", "mid", "flags", "c2"]
    torch.set_num_threads(os.cpu_count() or 8)
    attn = ["QhKl", "QlKh", "PhVl", "PlVh"]
    variants = [()] + [(t,) for t in attn] + list(itertools.combinations(attn, 2)) + [("XhWl",), ("XlWh",), tuple(attn)]
    print("| dropped | " + " | ".join(cases) + " |")
    print("|---|" + "---|" * len(cases))
    rows = {v: [] for v in variants}
    for name in cases:
        z, cfg, sd, data = load_case(name)
        with torch.no_grad():
            ref = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)["scores"]
            m_ref = orc.extract_matches(ref.float(), 0.2)["matches0"]
            keep = orc.conv1x1, orc.softmax_attention
            orc.conv1x1, orc.softmax_attention = conv1x1, softmax_attention
            try:
                for v in variants:
                    DROP = set(v)
                    got = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)["scores"]
                    err = (got - ref).abs().max().item()
                    nd = int((orc.extract_matches(got.float(), 0.2)["matches0"] != m_ref).sum())
                    rows[v].append(f"{err:.1e} ({nd})")
            finally:
                orc.conv1x1, orc.softmax_attention = keep
    for v in variants:
        print("| " + ("none (as built)" if not v else " + ".join(v)) + " | " + " | ".join(rows[v]) + " |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Debug: which split term does the attention kernel mishandle?  Compares the kernel against a CPU emulation of the
same split arithmetic with individual lo planes zeroed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")

def split(x):
    hi = x.half(); lo = (x - hi.float()).half(); return hi, lo

def emulate(qh, ql, kh, kl, vh, vl, H, dh):
    Z, nq, D = qh.shape; nk = kh.shape[1]
    hd = lambda t, n: t.float().view(Z, n, H, dh).transpose(1, 2)
    qh, ql, kh, kl, vh, vl = hd(qh, nq), hd(ql, nq), hd(kh, nk), hd(kl, nk), hd(vh, nk), hd(vl, nk)
    S = qh @ kh.transpose(-1, -2) + qh @ kl.transpose(-1, -2) + ql @ kh.transpose(-1, -2)
    p = torch.exp2(S - S.max(-1, keepdim=True).values)
    ph, pl = split(p); ph, pl = ph.float(), pl.float()
    O = ph @ vh + ph @ vl + pl @ vh
    return (O / p.sum(-1, keepdim=True)).transpose(1, 2).reshape(Z, nq, D)

def run(qh, ql, kh, kl, vh, vl, H, dh):
    Z, nq, D = qh.shape; nk = kh.shape[1]
    t = [x.to(dev).contiguous() for x in (qh, ql, kh, kl, vh, vl)]
    oh = torch.empty(Z, nq, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
    rc = lib.og_attention(t[0].data_ptr(), t[1].data_ptr(), D, t[2].data_ptr(), t[3].data_ptr(), D, t[4].data_ptr(), t[5].data_ptr(), D,
                          oh.data_ptr(), ol.data_ptr(), D, Z, nq, nk, H, dh, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return oh.float().cpu(), ol.float().cpu()

for (H, dh, nq, nk) in [(4, 32, 129, 64), (4, 64, 129, 64), (4, 16, 129, 64), (4, 32, 64, 64), (4, 32, 129, 128)]:
    g = torch.Generator().manual_seed(H * 100 + dh + nq)
    D = H * dh
    q, k, v = torch.randn(2, nq, D, generator=g) * 3, torch.randn(2, nk, D, generator=g) * 3, torch.randn(2, nk, D, generator=g) * 2
    qs = q * dh ** -0.5 * 1.4426950408889634
    (qh, ql), (kh, kl), (vh, vl) = split(qs), split(k), split(v)
    z = torch.zeros_like
    for name, args in (("all", (qh, ql, kh, kl, vh, vl)), ("vl=0", (qh, ql, kh, kl, vh, z(vl))), ("ql=0", (qh, z(ql), kh, kl, vh, vl)),
                       ("kl=0", (qh, ql, kh, z(kl), vh, vl)), ("all lo=0", (qh, z(ql), kh, z(kl), vh, z(vl)))):
        oh, ol = run(*args, H, dh)
        em = emulate(*args, H, dh)
        d = (oh + ol - em).abs()
        i = d.argmax().item()
        print(f"H={H} dh={dh} nq={nq} nk={nk} {name:9s}: kernel vs emulation max {d.max():.2e} (hi only {(oh - em).abs().max():.2e}) at flat index {i} -> row {i // D % nq} col {i % D}")

#!/usr/bin/env python3
"""Round 6: A/B of the attention kernel's two forms at the C2 self-attention shape (64 problems x 4 heads x 1024 x 1024, dh = 64) and the C4 shape (dh = 32):
   f16 x 3 for both contractions  vs  MX (the two P V cross products as block-scaled e4m3 MFMAs, attention.hip).  The 8-bit V rows are made here with torch
   (float8_e4m3fn) exactly as the projection epilogue would write them, so the MX result is checked against a float64 softmax attention as well as timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
SV = int(os.environ.get("MX_SV", "3"))


def v8_rows(vh, vl, H):
    """[Z, n, D] f16 hi / lo planes -> the f16-typed plane whose head rows are [e4m3(vh 2^-sv) x dh | e4m3(vl 2^(11 - sv)) x dh]"""
    Z, n, D = vh.shape
    dh = D // H
    h8 = (vh.float() * 2.0 ** -SV).to(torch.float8_e4m3fn).view(torch.uint8).view(Z, n, H, dh)
    l8 = (vl.float() * 2.0 ** (11 - SV)).to(torch.float8_e4m3fn).view(torch.uint8).view(Z, n, H, dh)
    return torch.cat([h8, l8], dim=-1).contiguous().view(Z, n, 2 * D).view(torch.float16)


def case(Z, n, D, H, scales=(0.5, 2.0, 2.0), reps=20):
    g = torch.Generator().manual_seed(0)
    q, k, v = [(torch.randn(Z, n, D, generator=g) * s).to(dev) for s in scales]
    (qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
    v8 = v8_rows(vh, vl, H)
    assert v8.shape == vl.shape
    oh = torch.empty(Z, n, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
    st = torch.cuda.current_stream().cuda_stream

    def run(mx):
        if mx: os.environ["OG_ATTN_MX_SV"] = str(SV)
        else: os.environ.pop("OG_ATTN_MX_SV", None)
        lo = v8 if mx else vl
        rc = lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), lo.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, n, n, H, D // H, None, st)
        assert rc == 0, rc
    # float64 reference on a few problems (q arrives pre-scaled by d^-1/2 log2 e in the product; the stage entry takes q as it is and works in base 2)
    zs = [0, Z // 2, Z - 1]
    qq = (qh.double() + ql.double())[zs].view(len(zs), n, H, -1).transpose(1, 2)
    kk = (kh.double() + kl.double())[zs].view(len(zs), n, H, -1).transpose(1, 2)
    vv = (vh.double() + vl.double())[zs].view(len(zs), n, H, -1).transpose(1, 2)
    s = qq @ kk.transpose(-1, -2)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    ref = ((p @ vv) / p.sum(-1, keepdim=True)).transpose(1, 2).reshape(len(zs), n, D)
    out = {}
    for mx in (0, 1):
        run(mx); torch.cuda.synchronize()
        o = (oh.double() + ol.double())[zs]
        out[mx] = (o - ref).abs().max().item()
    print(f"Z={Z} n={n} D={D} H={H} (dh={D // H}) sv={SV}: max |O - float64|: f16x3 {out[0]:.3e}, MX {out[1]:.3e}  (|O| max {ref.abs().max().item():.2f})", flush=True)
    if os.environ.get("MX_NO_TIMING"):
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fl = 4.0 * Z * n * n * D
    for rnd in range(3):
        for mx in (0, 1):
            for _ in range(3): run(mx)
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps): run(mx)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"  round {rnd} {'MX   ' if mx else 'f16x3'}: {us:7.1f} us per launch, algorithmic {fl / us / 1e6:.0f} TFLOP/s", flush=True)


case(64, 1024, 256, 4)
case(32, 1024, 256, 4)
case(16, 4096, 128, 4)
case(4, 200, 256, 4, reps=5)          # a masked last tile

#!/usr/bin/env python3
"""Debug helper (round 6): where the pipelined attention loop differs from float64 -- per query / channel pattern for tiny problems."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
def case(nq, nk, D=64, H=1, Z=1, ks=2.0, vs=2.0, seed=1, empty=False):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(Z, nq, D, generator=g) * 0.5).to(dev); k = (torch.randn(Z, nk, D, generator=g) * ks).to(dev); v = (torch.randn(Z, nk, D, generator=g) * vs).to(dev)
    (qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
    oh = (torch.empty if empty else torch.zeros)(Z, nq, D, device=dev, dtype=torch.float16); ol = torch.zeros_like(oh)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, nq, nk, H, D // H, None, st)
    assert rc == 0
    torch.cuda.synchronize()
    qq = (qh.double() + ql.double()).view(Z, nq, H, -1).transpose(1, 2); kk = (kh.double() + kl.double()).view(Z, nk, H, -1).transpose(1, 2); vv = (vh.double() + vl.double()).view(Z, nk, H, -1).transpose(1, 2)
    s = qq @ kk.transpose(-1, -2); p = torch.exp2(s - s.amax(-1, keepdim=True)); ref = ((p @ vv) / p.sum(-1, keepdim=True)).transpose(1, 2).reshape(Z, nq, D)
    o = oh.double() + ol.double()
    err = (o - ref).abs()
    bad = ~(err < 1e-3)
    print(f"nq={nq} nk={nk} D={D} H={H}: bad {int(bad.sum())} of {bad.numel()}, nan {int(torch.isnan(o).sum())}; bad queries {sorted(set(torch.nonzero(bad)[:, 1].tolist()))[:40]}; bad channels {sorted(set(torch.nonzero(bad)[:, 2].tolist()))[:70]}")
    if bad.any():
        i = torch.nonzero(bad)[0].tolist()
        print("   first bad", i, "got", o[tuple(i)].item(), "ref", ref[tuple(i)].item(), "| row of got:", [round(x, 3) for x in o[i[0], i[1], :8].tolist()], "ref:", [round(x, 3) for x in ref[i[0], i[1], :8].tolist()])
case(1024, 1024, D=256, H=4, Z=64, seed=65024); case(1024, 1024, D=256, H=4, Z=8, seed=65024); case(256, 1024, D=256, H=4, Z=64, seed=65024); case(77, 192, D=128, H=4, Z=2, seed=2192); case(32, 192, D=32, H=1, Z=1, seed=2192)

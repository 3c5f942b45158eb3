#!/usr/bin/env python3
"""Micro-benchmark of og_attention at the C2 self-attention shape (64 problems x 4 heads x 1024 x 1024, d=64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
Z, n, D, H = 64, 1024, 256, 4
g = torch.Generator().manual_seed(0)
q, k, v = [(torch.randn(Z, n, D, generator=g) * s).to(dev) for s in (0.5, 2.0, 2.0)]
(qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
oh = torch.empty(Z, n, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
st = torch.cuda.current_stream().cuda_stream
def run(): assert lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, n, n, H, D // H, None, st) == 0
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 4.0 * Z * n * n * D
print(f"ablate={os.environ.get('OG_ATTN_ABLATE','0')}: {us:.1f} us  algorithmic {fl/us/1e6:.0f} TF, executed f16 MFMA {3*fl/us/1e6:.0f} TF")

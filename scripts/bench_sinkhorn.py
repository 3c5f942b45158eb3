#!/usr/bin/env python3
"""Micro-benchmark of og_sinkhorn at the C2 shape (32 pairs, 1024x1024, 100 iterations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, m, n, iters = [int(v) for v in os.environ.get('OG_SK_SHAPE', '32,1024,1024,100').split(',')]
S = (torch.randn(B, m, n, generator=torch.Generator().manual_seed(0)) * 4).to(dev)
ws = torch.empty(lib.og_sinkhorn_workspace_bytes(B, m, n), device=dev, dtype=torch.uint8)
out = torch.empty(B, m + 1, n + 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(): assert lib.og_sinkhorn(S.data_ptr(), n, 1.0, B, m, n, iters, 1.0, out.data_ptr(), ws.data_ptr(), st) == 0
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
if os.environ.get("OG_CAL"):
    for name, fn in (("clone", lambda: S.clone()), ("sum", lambda: S.sum()), ("amax(dim=2)", lambda: S.amax(2))):
        fn(); torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"calibration torch {name}: {us:.1f} us for {S.numel()*4/1e6:.0f} MB read -> {S.numel()*4/us/1e6:.2f} TB/s read side")
print(f"shape {B}x{m}x{n} ablate={os.environ.get('OG_SINKHORN_ABLATE','0')}: {ms:.3f} ms per 100 iterations = {ms*1e3/iters:.1f} us/iter ({B*m*n*4/(ms*1e3/iters)/1e6:.2f} TB/s of S); finite={bool(torch.isfinite(out).all())}")

#!/usr/bin/env python3
"""Experiment: phase breakdown of the on-chip-resident Sinkhorn kernel (needs the OG_SK_TRACE build:
scripts/build_ablation.sh sk_trace -DOG_SK_TRACE=1; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_sk_trace.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, m, n, iters = [int(v) for v in os.environ.get('OG_SK_SHAPE', '32,1024,1024,100').split(',')]
S = (torch.randn(B, m, n, generator=torch.Generator().manual_seed(0)) * 4).to(dev)
ws = torch.empty(lib.og_sinkhorn_workspace_bytes(B, m, n), device=dev, dtype=torch.uint8)
out = torch.empty(B, m + 1, n + 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.og_sinkhorn(S.data_ptr(), n, 1.0, B, m, n, iters, 1.0, out.data_ptr(), ws.data_ptr(), st) == 0
torch.cuda.synchronize()
buf = np.zeros((2, 8, 8, 16), np.uint32)
lib.og_debug_sk_trace.restype = C.c_int; lib.og_debug_sk_trace.argtypes = [C.c_void_p, C.c_size_t]
assert lib.og_debug_sk_trace(buf.ctypes.data, buf.nbytes) == 0
names = ["pass 1: E *= g, row sums", "row partials across waves + new u, f", "pass 2: E *= f, column partials", "workgroup column partials (LDS, 2 barriers)",
         "publish partials", "owner: sweep G partials, publish totals", "sweep totals", "new v, g -> LDS, LSE of v", "syncthreads_or"]
t = buf.astype(np.int64)
for wg in range(2):
    seg = (t[wg, :, :, 1:10] - t[wg, :, :, 0:9]) & 0xFFFFFFFF          # [wave][iter][9 segments]
    period = (t[wg, :, 1:, 0] - t[wg, :, :-1, 0]) & 0xFFFFFFFF
    print(f"shape {B}x{m}x{n}: workgroup {wg}: iteration period {np.median(period):.0f} cycles")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} median {np.median(seg[:, :, i]):7.0f}   per wave: " + " ".join(f"{np.median(seg[w, :, i]):6.0f}" for w in range(8)))

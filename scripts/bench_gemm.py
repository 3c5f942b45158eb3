#!/usr/bin/env python3
"""Micro-benchmark of the split-f16 GEMM kernel on the C2 shapes (T = 65536 tokens).
OG_GEMM_VARIANT selects the kernel variant (read once by the library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
T = 65536
shapes = [("qkv", T, 768, 256), ("fc0", T, 512, 512), ("fc3", T, 256, 512), ("q_cross", T // 2, 256, 256), ("kv_cross", T // 2, 512, 256)]
g = torch.Generator().manual_seed(0)
print("variant", os.environ.get("OG_GEMM_VARIANT", "1"))
tot = 0.0
for name, M, N, K in shapes:
    a = torch.randn(M, K, generator=g).to(dev); b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    ah, al = ops.split_f16(a); bh, bl = ops.split_f16(b)
    bias = torch.randn(N, generator=g).to(dev)
    ch = torch.empty(M, N, device=dev, dtype=torch.float16); cl = torch.empty_like(ch)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.og_gemm_nt_f16x3(ah.data_ptr(), al.data_ptr(), K, bh.data_ptr(), bl.data_ptr(), K, M, N, K, bias.data_ptr(), 1, None, N,
                                  None, N, ch.data_ptr(), cl.data_ptr(), N, st)
        assert rc == 0, rc
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    reps = 20
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ref = torch.relu(a[:256].double() @ b.double().T + bias.double())
    err = (ops.merge_f16(ch[:256], cl[:256]).double() - ref).abs().max().item()
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{name:9s} M={M} N={N} K={K}: {us:8.1f} us  {tf:7.1f} TF algorithmic ({3 * tf:7.1f} TF f16 MFMA executed)  err {err:.1e}")
    tot += us
print(f"sum {tot:.1f} us")

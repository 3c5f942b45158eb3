#!/usr/bin/env python3
"""Exact-fp32 GEMM at the shapes of the training step (4 pairs x 1024 keypoints: 4096 tokens per image): how much of a launch is fill."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import ops, train
dev = torch.device("cuda:0")
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N, K in [(4096, 256, 256), (8192, 256, 256), (8192, 768, 256), (4096, 512, 512), (8192, 512, 512), (8192, 256, 512), (65536, 256, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
    t = timeit(lambda: ops.gemm_nt(x, w, b))
    dz = torch.randn(M, N, device=dev)
    t2 = timeit(lambda: train._gemm_splitk(dz, x))
    t3 = timeit(lambda: train._conv_backward(x, w, dz, True, False))
    print(f"M={M} N={N} K={K}: forward {t:7.1f} us ({2*M*N*K/t*1e-6:6.1f} TF)   dW split-K {t2:7.1f} us   dx {t3:7.1f} us")

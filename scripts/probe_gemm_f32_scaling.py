import os, sys, torch
sys.path.insert(0, "/root/repo")
from openglue_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N, K in [(65536, 256, 256), (65536, 256, 1024), (65536, 256, 4096), (65536, 1024, 1024), (16384, 256, 4096)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
    t = timeit(lambda: ops.gemm_nt(x, w, b))
    print(f"M={M} N={N} K={K}: {t:8.1f} us  {2*M*N*K/t*1e-6:6.1f} TF")

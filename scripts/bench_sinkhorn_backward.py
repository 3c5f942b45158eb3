#!/usr/bin/env python3
"""Time the optimal-transport layer of the training step alone (forward keeping the trajectory + backward over the unrolled iterations) at
B pairs x N x N scores: OG_SK_BWD_ROWS_GRID = the row workgroups per pair of the backward iteration kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import train
B, N, IT = int(os.environ.get("B", 4)), int(os.environ.get("N", 1024)), int(os.environ.get("ITERS", 20))
dev = torch.device("cuda:0")
S = (torch.randn(B, N, N, device=dev) * 3).requires_grad_(True)
z = torch.tensor(1.0, device=dev, requires_grad=True)
G = torch.randn(B, N + 1, N + 1, device=dev)
def fb():
    out = train.matching_log_probs(S, z, IT)
    return out, torch.autograd.grad(out, (S, z), G)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
out = train.matching_log_probs(S, z, IT)
t_f = timeit(lambda: train.matching_log_probs(S, z, IT))
t_fb = timeit(fb)
print(f"B={B} N={N} iters={IT} rows_grid={os.environ.get('OG_SK_BWD_ROWS_GRID', 'default')}: forward {t_f:.3f} ms, forward + backward {t_fb:.3f} ms, backward {t_fb - t_f:.3f} ms")

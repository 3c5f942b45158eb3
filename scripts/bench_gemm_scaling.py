#!/usr/bin/env python3
"""How does the time of ONE 256 x 256 tile per CU depend on the number of busy CUs?  fc.0-shaped tiles (K = 512, ReLU, hl32 rows out),
M = 256 * tiles, N = 256: every block is one tile, blocks <= 256 so every block has its own CU.  A flat line = per-CU latency bound;
growth with the tile count = a shared resource (clock / power, L2 / fabric, HBM).  Run with OG_GEMM_TILE=256 (forces the 256-tile kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for K in (512, 256):
    N = 256
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b_hl = ops.split_f16_hl(b * 256.0)
    bias = torch.randn(N, generator=g).to(dev)
    for tiles in (8, 64, 128, 256, 512):
        M = 256 * tiles
        a_hl = ops.split_f16_hl(torch.randn(M, K, generator=g).to(dev))
        ch = torch.empty(M, 2 * N, device=dev, dtype=torch.float16)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            rc = lib.og_gemm_nt_f16x3_reshl(a_hl.data_ptr(), 2 * K, b_hl.data_ptr(), 2 * K, M, N, K, 1.0 / 256.0, bias.data_ptr(), 1, None, 2 * N,
                                            None, N, ch.data_ptr(), None, 2 * N, 1, st)
            assert rc == 0, rc
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        reps = 50
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"K={K} tiles={tiles:5d} (M={M:7d}): {us:7.1f} us per launch, {us / max(1.0, tiles / 256):7.1f} us per round of 256 tiles, "
              f"{2.0 * M * N * K * 3 / us / 1e6:7.1f} TF f16 executed")

#!/usr/bin/env python3
"""Micro-benchmark of the q / k / v projection kernels on the C2 / C4 launch forms (self: all rows x q | k | v; cross: rows of image 0 -> q,
rows of image 1 -> q | k | v; kv: half the rows x k | v).  OG_PROJ_STREAM=0 / 1 picks the tile GEMM path or proj_stream_kernel (read once per
process: run twice); OPENGLUE_AMD_LIB selects an experiment build (scripts/build_mlp_ablation.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for D, M in ((256, 65536), (128, 65536), (128, 131072)):
    N = 3 * D
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, D, generator=g) * 0.05
    b = (torch.randn(N, generator=g) * 0.3).to(dev)
    nb = lib.og_proj_block_stream_bytes(N, D)
    sh = torch.empty(nb, dtype=torch.uint8)
    _lib.check(lib.og_proj_block_pack(N, D, w.data_ptr(), sh.data_ptr()), "pack")
    sd = sh.to(dev)
    rows = ops.split_f16_hl((torch.randn(M, D, generator=g) * 1.5).to(dev))
    inv = torch.full((1,), 1.0 / 256.0, device=dev)
    yh = torch.zeros(M, N, device=dev, dtype=torch.float16); yl = torch.zeros_like(yh)
    w_hl = ops.split_f16_hl((w * 256.0).to(dev))
    forms = {"self": (M, 0, (0, 0), (0, N)), "cross": (M, M // 2, (0, D), (0, N)), "kv": (M // 2, 0, (0, 0), (D, N))}
    for name, (R, split, ca, cb) in forms.items():
        def blk():
            rc = lib.og_proj_block(rows.data_ptr(), 2 * D, R, D, sd.data_ptr(), b.data_ptr(), inv.data_ptr(), yh.data_ptr(), yl.data_ptr(), N,
                                   split, ca[0] // 32, ca[1] // 32, cb[0] // 32, cb[1] // 32, st)
            assert rc == 0, rc
        us = timed(blk)
        wr = (split * (ca[1] - ca[0]) + (R - split) * (cb[1] - cb[0])) * 4
        rd = R * D * 4
        line = f"D={D} {name:5s} R={R}: og_proj_block {us:7.1f} us  ({(wr + rd) / us / 1e6:5.2f} TB/s of {((wr + rd) / 1e6):.0f} MB)"
        if name != "cross" and os.environ.get("OG_PROJ_STREAM") != "1":      # the tile GEMM on the same form
            def tile():
                rc = lib.og_gemm_nt_f16x3_reshl(rows.data_ptr(), 2 * D, w_hl.data_ptr() + cb[0] * 2 * D * 2, 2 * D, R, cb[1] - cb[0], D, 1.0 / 256.0,
                                                b.data_ptr() + cb[0] * 4, 0, None, 0, None, 0, yh.data_ptr() + cb[0] * 2, yl.data_ptr() + cb[0] * 2, N, 0, st)
                assert rc == 0, rc
            line += f"   tile GEMM {timed(tile):7.1f} us"
        print(line, flush=True)

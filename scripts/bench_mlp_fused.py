#!/usr/bin/env python3
"""Micro-benchmark of the fused message-MLP kernel (csrc/mlp_fused.hip) against the two split-f16 GEMM launches it replaces, on
the C2 shapes (self layer: T = 65536 token rows, cross layer: 32768).  OPENGLUE_AMD_LIB selects an experiment build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
D = 256
g = torch.Generator().manual_seed(0)
w0 = torch.randn(2 * D, 2 * D, generator=g) * 0.04
w3 = torch.randn(D, 2 * D, generator=g) * 0.05
b0 = (torch.randn(2 * D, generator=g) * 0.3).to(dev)
b3 = (torch.randn(D, generator=g) * 0.3).to(dev)
stream_host = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
_lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), stream_host.data_ptr()), "pack")
wstream = stream_host.to(dev)
w0_hl, w3_hl = ops.split_f16_hl((w0 * 256.0).to(dev)), ops.split_f16_hl((w3 * 256.0).to(dev))
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, M in (("self", 65536), ("cross", 32768)):
    xo = torch.randn(M, 2 * D, generator=g) * 1.5
    rows0 = ops.split_f16_hl(xo.to(dev))              # [M][4D halves]
    rows = rows0.clone()
    hid = torch.empty(M, 4 * D, device=dev, dtype=torch.float16)

    def fused():
        rc = lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, wstream.data_ptr(), b0.data_ptr(), b3.data_ptr(), st)
        assert rc == 0, rc

    def two():
        rc = lib.og_gemm_nt_f16x3_reshl(rows.data_ptr(), 4 * D, w0_hl.data_ptr(), 4 * D, M, 2 * D, 2 * D, 1.0 / 256.0, b0.data_ptr(), 1,
                                        None, 0, None, 0, hid.data_ptr(), None, 4 * D, 1, st)
        assert rc == 0, rc
        rc = lib.og_gemm_nt_f16x3_reshl(hid.data_ptr(), 4 * D, w3_hl.data_ptr(), 4 * D, M, D, 2 * D, 1.0 / 256.0, b3.data_ptr(), 0,
                                        rows.data_ptr(), 4 * D, None, 0, rows.data_ptr(), None, 4 * D, 1, st)
        assert rc == 0, rc

    # correctness of one application against float64 on a slice, and against each other
    rows.copy_(rows0); fused(); torch.cuda.synchronize(); out_f = ops.merge_f16_hl(rows)[:, :D].cpu()
    rows.copy_(rows0); two(); torch.cuda.synchronize(); out_t = ops.merge_f16_hl(rows)[:, :D].cpu()
    xin = ops.merge_f16_hl(rows0[:512]).cpu().double()
    ref = xin[:, :D] + torch.relu(xin @ w0.double().T + b0.cpu().double()) @ w3.double().T + b3.cpu().double()
    print(f"{name}: fused err {(out_f[:512].double() - ref).abs().max():.2e}  two-launch err {(out_t[:512].double() - ref).abs().max():.2e}  "
          f"fused vs two {(out_f - out_t).abs().max():.2e}")
    # timing: repeated in-place application drifts the values (x grows), harmless for timing; reset between the two
    rows.copy_(rows0); us_f = timed(fused)
    rows.copy_(rows0); us_t = timed(two)
    fl = 2.0 * M * (2 * D * 2 * D + D * 2 * D)
    print(f"{name} M={M}: fused {us_f:7.1f} us ({fl / us_f / 1e6:6.1f} TF algorithmic)   two launches {us_t:7.1f} us ({fl / us_t / 1e6:6.1f} TF)")

#!/usr/bin/env python3
"""Micro-benchmark of the split-f16 GEMM kernel on the C2 shapes (T = 65536 tokens).
OPENGLUE_AMD_LIB selects an experiment build (scripts/build_ablation.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
T = 65536
shapes = [("qkv", T, 768, 256), ("fc0", T, 512, 512), ("fc3", T, 256, 512), ("q_cross", T // 2, 256, 256), ("kv_cross", T // 2, 512, 256),
          ("qkv_cross", T // 2, 768, 256), ("fc0_cross", T // 2, 512, 512), ("fc3_cross", T // 2, 256, 512)]
g = torch.Generator().manual_seed(0)
tot = 0.0
for name, M, N, K in shapes:
    a = torch.randn(M, K, generator=g).to(dev); b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    a_hl, b_hl = ops.split_f16_hl(a), ops.split_f16_hl(b * 256.0)
    bias = torch.randn(N, generator=g).to(dev)
    planes = name in ('qkv', 'q_cross', 'kv_cross', 'qkv_cross')       # as in og_forward: q/k/v leave as planes, the MLP as hl32 rows
    relu = name.startswith('fc0')                                      # fc.0: ReLU; fc.3: + the (hi, lo) residual stream; q/k/v: bias only
    res = ops.split_f16_hl(torch.randn(M, N, generator=g).to(dev)) if name.startswith('fc3') else None
    ch = torch.empty(M, N if planes else 2 * N, device=dev, dtype=torch.float16); cl = torch.empty_like(ch) if planes else None
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.og_gemm_nt_f16x3_reshl(a_hl.data_ptr(), 2 * K, b_hl.data_ptr(), 2 * K, M, N, K, 1.0 / 256.0, bias.data_ptr(), int(relu),
                                        res.data_ptr() if res is not None else None, 2 * N,
                                        None, N, ch.data_ptr(), cl.data_ptr() if planes else None, N if planes else 2 * N, 0 if planes else 1, st)
        assert rc == 0, rc
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    reps = 20
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ref = a[:256].double() @ b.double().T + bias.double()
    if relu: ref = torch.relu(ref)
    if res is not None: ref = ref + ops.merge_f16_hl(res[:256]).double()
    got = ops.merge_f16(ch[:256], cl[:256]) if planes else ops.merge_f16_hl(ch[:256])
    err = (got.double() - ref).abs().max().item()
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{name:9s} M={M} N={N} K={K}: {us:8.1f} us  {tf:7.1f} TF algorithmic ({3 * tf:7.1f} TF f16 MFMA executed)  err {err:.1e}")
    tot += us
print(f"sum {tot:.1f} us")

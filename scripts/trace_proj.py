#!/usr/bin/env python3
"""Experiment: where a 128-token tile of proj_stream_kernel spends its time (needs the OG_MLP_TRACE build: scripts/build_mlp_ablation.sh
trace -DOG_MLP_TRACE=1; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_trace.so, OG_PROJ_STREAM=1).  Per wave of every block: shader-cycle stamps at
entry, after the prologue, at every stage hand-over (before the DMA wait, after it, after the barrier), around every super-pair epilogue, at the end."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
lib.og_debug_mlp_trace.restype = C.c_int
lib.og_debug_mlp_trace.argtypes = [C.c_void_p, C.c_size_t]
st = torch.cuda.current_stream().cuda_stream
for D, M in ((256, 65536), (256, 32768), (128, 65536)):
    N = 3 * D
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, D, generator=g) * 0.05; b = (torch.randn(N, generator=g) * 0.3).to(dev)
    sh = torch.empty(lib.og_proj_block_stream_bytes(N, D), dtype=torch.uint8)
    _lib.check(lib.og_proj_block_pack(N, D, w.data_ptr(), sh.data_ptr()), "pack")
    sd = sh.to(dev)
    rows = ops.split_f16_hl((torch.randn(M, D, generator=g) * 1.5).to(dev))
    inv = torch.full((1,), 1.0 / 256.0, device=dev)
    yh = torch.zeros(M, N, device=dev, dtype=torch.float16); yl = torch.zeros_like(yh)
    def run():
        assert lib.og_proj_block(rows.data_ptr(), 2 * D, M, D, sd.data_ptr(), b.data_ptr(), inv.data_ptr(), yh.data_ptr(), yl.data_ptr(), N, 0, 0, 0, 0, N // 32, st) == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    buf = np.zeros((512, 8, 4, 64), np.uint32)
    assert lib.og_debug_mlp_trace(buf.ctypes.data, buf.nbytes) == 0
    nblk = min(512, M // 128)
    t = buf[:nblk].astype(np.int64)
    d = lambda x, y: (x - y) & 0xFFFFFFFF
    S = (N // 128) * (D // 64); NSP = N // 128; SPS = D // 64
    pre, post, bar = t[:, :, 0, :S], t[:, :, 1, :S], t[:, :, 2, :S]
    entry, prol, end = t[:, :, 3, 0], t[:, :, 3, 1], t[:, :, 3, 30]
    eb = np.stack([t[:, :, 3, 8 + 2 * i] for i in range(NSP)], -1); ee = np.stack([t[:, :, 3, 9 + 2 * i] for i in range(NSP)], -1)
    f = lambda x: f"{np.median(x):8.0f} (p10 {np.percentile(x, 10):7.0f} p90 {np.percentile(x, 90):7.0f})"
    print(f"\n=== D={D} M={M}: traced build {us:.1f} us per launch, {nblk} blocks traced, {S} stages of 24 MFMAs per wave")
    print(f"  block life (cycles, per wave)   : {f(d(end, entry))}   [pure MFMA issue, 2 waves per SIMD: {S * 48 * 32}]")
    print(f"  entry -> prologue done          : {f(d(prol, entry))}")
    per = d(pre[:, :, 1:S], pre[:, :, 0:S - 1])
    print(f"  stage period (hand-over to hand-over): {f(per)}   [MFMA-bound: 1536]")
    print(f"  DMA wait at hand-over           : {f(d(post[:, :, :S - 1], pre[:, :, :S - 1]))}")
    print(f"  barrier wait                    : {f(d(bar[:, :, :S - 1], post[:, :, :S - 1]))}")
    print(f"  epilogue of a super-pair        : {f(d(ee, eb))}")
    print(f"  last epilogue -> stores acknowledged: {f(d(end, ee[:, :, -1]))}")
    print("  median period by stage          :", " ".join(f"{int(np.median(per[:, :, i]))}" for i in range(per.shape[2])))
    print("  median DMA wait by stage        :", " ".join(f"{int(np.median(d(post[:, :, i], pre[:, :, i])))}" for i in range(S - 1)))
    print("  median barrier wait by stage    :", " ".join(f"{int(np.median(d(bar[:, :, i], post[:, :, i])))}" for i in range(S - 1)))
    print("  median epilogue by super-pair   :", " ".join(f"{int(np.median(d(ee[:, :, i], eb[:, :, i])))}" for i in range(NSP)))

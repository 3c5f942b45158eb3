#!/usr/bin/env python3
"""Experiment: where a 128-token tile of the fused message-MLP kernel spends its time (needs the OG_MLP_TRACE build:
scripts/build_mlp_ablation.sh trace -DOG_MLP_TRACE=1; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_trace.so).  Per wave of every block:
shader-cycle stamps at entry, at every stage hand-over (before the DMA wait, after it, after the barrier), at the end of the stage
loop, after the last store was issued and after the stores were acknowledged."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
lib.og_debug_mlp_trace.restype = C.c_int
lib.og_debug_mlp_trace.argtypes = [C.c_void_p, C.c_size_t]
D = int(os.environ.get("OG_TRACE_D", "256"))      # 256 or 128
g = torch.Generator().manual_seed(0)
w0 = torch.randn(2 * D, 2 * D, generator=g) * 0.04; w3 = torch.randn(D, 2 * D, generator=g) * 0.05
b0 = (torch.randn(2 * D, generator=g) * 0.3).to(dev); b3 = (torch.randn(D, generator=g) * 0.3).to(dev)
sh = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
_lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), sh.data_ptr()), "pack")
ws = sh.to(dev)
st = torch.cuda.current_stream().cuda_stream
for M in (int(a) for a in (sys.argv[1:] or ["65536", "32768", "16384"])):
    rows0 = ops.split_f16_hl((torch.randn(M, 2 * D, generator=g) * 1.5).to(dev)); rows = rows0.clone()
    def run():
        assert lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, ws.data_ptr(), b0.data_ptr(), b3.data_ptr(), st) == 0
    for _ in range(3): run()
    rows.copy_(rows0)
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
    S = 48 if D == 256 else 12
    pre, post, bar = t[:, :, 0, :S], t[:, :, 1, :S], t[:, :, 2, :S]
    entry, loop_end, st_iss, st_ack = t[:, :, 3, 0], t[:, :, 3, 1], t[:, :, 3, 2], t[:, :, 3, 3]
    f = lambda x: f"{np.median(x):8.0f} (p10 {np.percentile(x, 10):7.0f} p90 {np.percentile(x, 90):7.0f})"
    print(f"\n=== M={M}: traced build {us:.1f} us per launch, {nblk} blocks traced")
    print(f"  block life (cycles, per wave)   : {f(d(st_ack, entry))}   [pure MFMA issue, 2 waves per SIMD: {S * 48 * 32}]")
    print(f"  entry -> end of stage 0         : {f(d(bar[:, :, 0], entry))}")
    per = d(bar[:, :, 1:S - 1], bar[:, :, 0:S - 2])
    print(f"  stage period (all, steady)      : {f(per)}   [MFMA-bound: 1536]")
    print(f"  DMA wait at hand-over           : {f(d(post[:, :, :S - 1], pre[:, :, :S - 1]))}")
    print(f"  barrier wait                    : {f(d(bar[:, :, :S - 1], post[:, :, :S - 1]))}")
    print(f"  last barrier -> loop end        : {f(d(loop_end, bar[:, :, S - 2]))}")
    print(f"  epilogue until last store issued: {f(d(st_iss, loop_end))}, store drain {f(d(st_ack, st_iss))}")
    print("  median stage period by stage    :", " ".join(f"{int(np.median(per[:, :, i]))}" for i in range(per.shape[2])))
    print("  median DMA wait by stage        :", " ".join(f"{int(np.median(d(post[:, :, i], pre[:, :, i])))}" for i in range(S - 1)))
    print("  median barrier wait by stage    :", " ".join(f"{int(np.median(d(bar[:, :, i], post[:, :, i])))}" for i in range(S - 1)))

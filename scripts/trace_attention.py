#!/usr/bin/env python3
"""Experiment: per-segment shader-cycle breakdown of the attention kernel (needs the OG_ATTN_TRACE build:
scripts/build_ablation.sh attn_trace -DOG_ATTN_TRACE=1; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_attn_trace.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
Z, n, D, H = 64, 1024, 256, 4
g = torch.Generator().manual_seed(0)
q, k, v = [(torch.randn(Z, n, D, generator=g) * s).to(dev) for s in (0.5, 2.0, 2.0)]
(qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
oh = torch.empty(Z, n, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
st = torch.cuda.current_stream().cuda_stream
def run(): assert lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, n, n, H, D // H, None, st) == 0
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"traced build: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch")
buf = np.zeros((2, 4, 16, 8), np.uint32)
lib.og_debug_attn_trace.restype = C.c_int
lib.og_debug_attn_trace.argtypes = [C.c_void_p, C.c_size_t]
assert lib.og_debug_attn_trace(buf.ctypes.data, buf.nbytes) == 0
names = (["issue DMA", "QK^T + S ready", "softmax + split", "PV", "vmcnt(0)", "barrier", "-"] if os.environ.get("OG_ATTN_DMA", "1") != "0" else
         ["issue global loads", "QK^T + S ready", "PV(t-1) issue", "softmax + split", "barrier 1", "LDS staging stores", "barrier 2"])
for wg in range(2):
    for w in range(4):
        t = buf[wg, w].astype(np.int64)
        seg = (t[:, 1:] - t[:, :-1]) & 0xFFFFFFFF                     # 7 segments per tile
        tile = (t[1:, 0] - t[:-1, 0]) & 0xFFFFFFFF                     # tile period
        mid = slice(4, 12)
        print(f"wg {wg} wave {w}: tile period {tile[mid].mean():7.0f} cyc | " + " | ".join(f"{names[i]} {seg[mid, i].mean():6.0f}" for i in range(7)))


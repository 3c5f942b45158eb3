#!/usr/bin/env python3
"""Phase stamps of mlp_small_kernel (needs scripts/build_ablation.sh mlp_trace -DOG_MLP_TRACE=1 and OPENGLUE_AMD_LIB=openglue_amd/lib/libog_mlp_trace.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
D, M = 256, int(os.environ.get("OG_M", "2048"))
g = torch.Generator().manual_seed(0)
w0 = torch.randn(2 * D, 2 * D, generator=g) * 0.04; w3 = torch.randn(D, 2 * D, generator=g) * 0.05
b0 = (torch.randn(2 * D, generator=g) * 0.3).to(dev); b3 = (torch.randn(D, generator=g) * 0.3).to(dev)
sh = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
_lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), sh.data_ptr()), "pack")
ws = sh.to(dev)
rows0 = ops.split_f16_hl((torch.randn(M, 2 * D, generator=g) * 1.5).to(dev)); rows = rows0.clone()
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for rep in range(30):
    rows.copy_(rows0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, ws.data_ptr(), b0.data_ptr(), b3.data_ptr(), st) == 0
    e1.record(); torch.cuda.synchronize()
    if rep >= 10: tot += e0.elapsed_time(e1)
print(f"M = {M}: {tot / 20 * 1e3:.1f} us per launch (event pair)")
buf = np.zeros((512, 8, 4, 64), np.uint32)
lib.og_debug_mlp_trace.restype = C.c_int; lib.og_debug_mlp_trace.argtypes = [C.c_void_p, C.c_size_t]
assert lib.og_debug_mlp_trace(buf.ctypes.data, buf.nbytes) == 0
nb = (M + 31) // 32
t = buf[:nb, :, 0, :8].astype(np.int64)
names = ["prologue loads + tile copy + bias", "barrier", "fc.0 (32 k-steps)", "convert + fc.3", "barrier", "round 0", "round 1 + stores"]
seg = (t[:, :, 1:] - t[:, :, :-1]) & 0xFFFFFFFF
print("cycles per phase, mean over workgroups and waves (min .. max):")
for i, n in enumerate(names):
    print(f"  {n:36s} {seg[:, :, i].mean():8.0f}  ({seg[:, :, i].min()} .. {seg[:, :, i].max()})")
tot_c = (t[:, :, 7] - t[:, :, 0]) & 0xFFFFFFFF
print(f"  total {tot_c.mean():.0f} cycles per wave; first start to last end over the launch: {(t[:, :, 7].max() - t[:, :, 0].min()) & 0xFFFFFFFF} cycles")

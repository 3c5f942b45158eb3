#!/usr/bin/env python3
"""Timing of the fused message-MLP kernel with the token rows RESET before every launch (in-place repetition drifts the values, and on a
power-capped part the data decides the clock): C2 self-layer shape, one HIP-event pair per launch.  OPENGLUE_AMD_LIB selects a build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
D, M = 256, 65536
g = torch.Generator().manual_seed(0)
w0 = torch.randn(2 * D, 2 * D, generator=g) * 0.04; w3 = torch.randn(D, 2 * D, generator=g) * 0.05
b0 = (torch.randn(2 * D, generator=g) * 0.3).to(dev); b3 = (torch.randn(D, generator=g) * 0.3).to(dev)
sh = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
_lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), sh.data_ptr()), "pack")
ws = sh.to(dev)
rows0 = ops.split_f16_hl((torch.randn(M, 2 * D, generator=g) * 1.5).to(dev)); rows = rows0.clone()
st = torch.cuda.current_stream().cuda_stream
tot = 0.0; n = 0
for rep in range(60):
    rows.copy_(rows0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, ws.data_ptr(), b0.data_ptr(), b3.data_ptr(), st) == 0
    e1.record(); torch.cuda.synchronize()
    if rep >= 10: tot += e0.elapsed_time(e1); n += 1
print(f"{os.path.basename(os.environ.get('OPENGLUE_AMD_LIB', 'default'))}: {tot / n * 1e3:.1f} us per launch")

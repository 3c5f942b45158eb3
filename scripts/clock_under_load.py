#!/usr/bin/env python3
"""Which shader clock does the chip grant each hot kernel?  A one-wave sampler kernel (scripts/probes/clock_sampler_lib.hip ->
openglue_amd/lib/libprobe_clock_sampler.so, built by scripts/build_probes.sh) runs on its own stream and measures shader ticks per
20 us window of the constant 100 MHz counter while the kernel under test loops on the main stream (C2 shapes)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
probe = C.CDLL(os.path.join(ROOT, "openglue_amd", "lib", "libprobe_clock_sampler.so"))
probe.clock_sampler_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
side = torch.cuda.Stream()
N, GAP = 400, 2000                      # 400 windows of 20 us = 8 ms

def sample(fn, name, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    out = torch.zeros(2 * N, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    assert probe.clock_sampler_launch(out.data_ptr(), N, GAP, side.cuda_stream) == 0
    e0.record()
    for _ in range(reps): fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    o = out.cpu().numpy().reshape(N, 2).astype(np.float64)
    ghz = o[:, 1] / o[:, 0] * 0.1       # ticks per 10 ns
    busy = int(min(N, ms * 1e3 / (GAP / 100.0)))          # windows that overlap the loop
    w = ghz[5:max(6, busy - 5)]
    print(f"{name:34s} {ms / reps * 1e3:8.1f} us / launch   shader clock while it runs: median {np.median(w):.3f} GHz (min {w.min():.3f}, max {w.max():.3f}); idle tail {np.median(ghz[busy + 5:]) if busy + 10 < N else float('nan'):.3f} GHz")

g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
# nothing running
sample(lambda: None, "idle", 1)
# attention, C2 self layer
Z, n, D, H = 64, 1024, 256, 4
q, k, v = [(torch.randn(Z, n, D, generator=g) * s).to(dev) for s in (0.5, 2.0, 2.0)]
(qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
oh = torch.empty(Z, n, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
def attn(): assert lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, n, n, H, D // H, None, st) == 0
sample(attn, "attention_dma_kernel<64> (C2)", 40)
# fused MLP, C2 self layer
w0 = torch.randn(2 * D, 2 * D, generator=g) * 0.04; w3 = torch.randn(D, 2 * D, generator=g) * 0.05
b0 = (torch.randn(2 * D, generator=g) * 0.3).to(dev); b3 = (torch.randn(D, generator=g) * 0.3).to(dev)
sh = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
_lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), sh.data_ptr()), "pack")
ws = sh.to(dev)
M = 65536
rows = ops.split_f16_hl((torch.randn(M, 2 * D, generator=g) * 0.5).to(dev))
def mlp(): assert lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, ws.data_ptr(), b0.data_ptr(), b3.data_ptr(), st) == 0
sample(mlp, "mlp_fused_kernel<256> (C2 self)", 60)
# the q/k/v projection: 65536 x 256 -> 768
wq = ops.split_f16_hl((torch.randn(3 * D, D, generator=g) * 0.05 * 256.0).to(dev))
xr = ops.split_f16_hl((torch.randn(M, D, generator=g) * 0.5).to(dev))
bq = torch.zeros(3 * D, device=dev)
yh = torch.empty(M, 3 * D, device=dev, dtype=torch.float16); yl = torch.empty_like(yh)
def qkv():
    rc = lib.og_gemm_nt_f16x3_reshl(xr.data_ptr(), 2 * D, wq.data_ptr(), 2 * D, M, 3 * D, D, 1.0 / 256.0, bq.data_ptr(), 0, None, 0, None, 0, yh.data_ptr(), yl.data_ptr(), 3 * D, 0, st)
    assert rc == 0, rc
try:
    sample(qkv, "gemm_nt_f16x3 q/k/v (65536x256x768)", 80)
except Exception as e:
    print("qkv gemm skipped:", e)
# Sinkhorn, C2
S = (torch.randn(32, 1024, 1024, generator=g) * 2).to(dev)
sample(lambda: ops.sinkhorn(S, 1.0, 100), "sinkhorn stage (C2, resident)", 8)

# the whole C2 step (bench.py's workload): clock timeline over ~2 steps, 10 us windows
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
kw = dict(syn.CONFIGS["C2"]); (m_, n_), B = kw.pop("kpts"), kw.pop("batch")
cfg = syn.make_config(**kw)
model = SuperGlue(cfg).eval(); model.load_state_dict(syn.make_state_dict(cfg, seed=0), strict=True); model.to(dev)
data = syn.make_batch(B, m_, n_, kw["descriptor_dim"], kw["side_info_size"], seed=0, device=dev)
for _ in range(5): model.match(data, 0.2, both_sides=True)
torch.cuda.synchronize()
N2, GAP2 = 3000, 1000
out = torch.zeros(2 * N2, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): model.match(data, 0.2, both_sides=True)       # warm: sustained load before the sampled steps
assert probe.clock_sampler_launch(out.data_ptr(), N2, GAP2, side.cuda_stream) == 0
e0.record()
for _ in range(4): model.match(data, 0.2, both_sides=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
o = out.cpu().numpy().reshape(N2, 2).astype(np.float64)
ghz = o[:, 1] / o[:, 0] * 0.1
t = np.cumsum(o[:, 0]) / 100.0       # us
busy = t < ms * 4 * 1e3 - 50
print(f"C2 step {ms:.3f} ms; shader clock over 4 steps: mean {ghz[busy].mean():.3f} GHz, median {np.median(ghz[busy]):.3f}, p10 {np.percentile(ghz[busy], 10):.3f}, p90 {np.percentile(ghz[busy], 90):.3f}")
# timeline of the second step in 100 us buckets
sel = (t > ms * 1e3) & (t < 2 * ms * 1e3)
tb = ((t[sel] - ms * 1e3) // 100).astype(int)
line = " ".join(f"{ghz[sel][tb == b].mean():.2f}" for b in range(tb.max() + 1))
print("clock per 100 us of one step [GHz]:", line)

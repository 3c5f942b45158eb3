#!/usr/bin/env python3
"""Experiment: where a 256x256 tile of the split-f16 GEMM spends its time (needs the OG_GEMM_TRACE build:
scripts/build_ablation.sh gemm_trace -DOG_GEMM_TRACE=1; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_gemm_trace.so).
Per wave of every block: shader-cycle stamps at kernel entry, after the prologue barrier, at every stage hand-over
(before the DMA wait, after it, after the barrier), at the end of the k-loop, after the last store was issued and after
the stores were acknowledged; plus the CU the block ran on and its real-time entry / exit."""
import ctypes as C, os, sys, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
lib.og_debug_gemm_trace.restype = C.c_int
lib.og_debug_gemm_trace.argtypes = [C.c_void_p, C.c_size_t]
T = 65536
shapes = [("qkv", T, 768, 256, True), ("fc0", T, 512, 512, False), ("fc3", T, 256, 512, False)]
if os.environ.get("OG_TRACE_SMALL"):      # few blocks: is the epilogue store rate a per-CU or a whole-chip limit?
    shapes = [("fc3_32blocks", 8192, 256, 512, False), ("fc3_64blocks", 16384, 256, 512, False), ("fc3_128blocks", 32768, 256, 512, False)]
g = torch.Generator().manual_seed(0)
W = 64
for name, M, N, K, planes in shapes:
    a = torch.randn(M, K, generator=g).to(dev); b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    a_hl, b_hl = ops.split_f16_hl(a), ops.split_f16_hl(b * 256.0)
    bias = torch.randn(N, generator=g).to(dev)
    ch = torch.empty(M, N if planes else 2 * N, device=dev, dtype=torch.float16); cl = torch.empty_like(ch) if planes else None
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.og_gemm_nt_f16x3(a_hl.data_ptr(), 2 * K, b_hl.data_ptr(), 2 * K, M, N, K, 1.0 / 256.0, bias.data_ptr(), 1, None, N,
                                  None, N, ch.data_ptr(), cl.data_ptr() if planes else None, N if planes else 2 * N, 0 if planes else 1, st)
        assert rc == 0, rc
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    buf = np.zeros((1024, 8, W), np.uint32)
    assert lib.og_debug_gemm_trace(buf.ctypes.data, buf.nbytes) == 0
    nblk = (M // 256) * (N // 256)
    t = buf[:nblk].astype(np.int64)
    nk = int(t[0, 0, 5]); ge = 8 + 3 * (nk - 1)
    d = lambda x, y: (x - y) & 0xFFFFFFFF
    pro = d(t[:, :, 7], t[:, :, 6])
    kts = np.arange(nk - 1)
    wait_dma = d(t[:, :, 9 + 3 * kts], t[:, :, 8 + 3 * kts])
    wait_bar = d(t[:, :, 10 + 3 * kts], t[:, :, 9 + 3 * kts])
    period = d(t[:, :, 10 + 3 * kts[1:]], t[:, :, 10 + 3 * kts[:-1]])
    first = d(t[:, :, 10], t[:, :, 7])
    last = d(t[:, :, ge], t[:, :, 10 + 3 * (nk - 2)])
    epi = d(t[:, :, ge + 1], t[:, :, ge]); drain = d(t[:, :, ge + 2], t[:, :, ge + 1])
    total = d(t[:, :, ge + 2], t[:, :, 6])
    print(f"\n=== {name} M={M} N={N} K={K}: traced build {us:.1f} us per launch, {nblk} blocks, nk={nk}")
    f = lambda x: f"{np.median(x):8.0f} (p10 {np.percentile(x, 10):7.0f} p90 {np.percentile(x, 90):7.0f})"
    print(f"  block life (cycles, per wave): {f(total)}   [pure MFMA issue at 2 waves/SIMD: {nk * 96 * 32}]")
    print(f"  prologue entry->stage0 ready : {f(pro)}")
    print(f"  stage 0 (no prefetched frags): {f(first)}")
    print(f"  stage period (steady)        : {f(period)}   [MFMA-bound: 3072]")
    print(f"    of which DMA wait at hand-over {f(wait_dma)}, barrier wait {f(wait_bar)}")
    print(f"  last stage                   : {f(last)}")
    print(f"  epilogue until last store issued {f(epi)}, store drain {f(drain)}")
    # per-stage profile of the median block
    print("  median stage period by kt:", " ".join(f"{int(np.median(period[:, :, i]))}" for i in range(period.shape[2])))
    print("  median DMA wait by kt    :", " ".join(f"{int(np.median(wait_dma[:, :, i]))}" for i in range(wait_dma.shape[2])))
    # CU timeline from the real-time counter (100 MHz): blocks per CU, gap between a block's exit and the next entry
    cu = collections.defaultdict(list)
    for bi in range(nblk):
        hw, xcc = int(t[bi, 0, 1]), int(t[bi, 0, 2]) & 0xF
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)       # xcc, se, sh, cu
        cu[key].append((int(t[bi, :, 3].min()), int(t[bi, :, 4].max()), bi))
    per = collections.Counter(len(v) for v in cu.values())
    gaps, lifes = [], []
    t_first = min(v[0] for vs in cu.values() for v in vs); t_last = max(v[1] for vs in cu.values() for v in vs)
    for vs in cu.values():
        vs.sort()
        lifes += [(e - s_) & 0xFFFFFFFF for s_, e, _ in vs]
        gaps += [(vs[i + 1][0] - vs[i][1]) & 0xFFFFFFFF for i in range(len(vs) - 1)]
    print(f"  CUs used {len(cu)}, blocks per CU {dict(per)}; kernel span (first entry -> last exit) {(t_last - t_first) * 0.01:.1f} us")
    print(f"  block life {np.median(lifes) * 0.01:.1f} us (max {max(lifes) * 0.01:.1f}); gap between consecutive blocks on a CU: "
          f"{(np.median(gaps) * 0.01 if gaps else 0):.2f} us (max {(max(gaps) * 0.01 if gaps else 0):.2f})")
    starts = sorted((v[0] - t_first) * 0.01 for vs in cu.values() for v in vs)
    print(f"  block entry times (us after the first): p10 {np.percentile(starts, 10):.1f} p50 {np.percentile(starts, 50):.1f} p90 {np.percentile(starts, 90):.1f} max {starts[-1]:.1f}")

#!/usr/bin/env python3
"""Round 6: the attention kernel's software-pipelined tile loop (OG_ATTN_PIPE=1, read once per process) against a float64 softmax attention over the tile-count
edge cases (1 .. 5 tiles, partial last tiles, dh = 64 and 32, a spike that forces the running max to move mid-way), then timing at the C2 / C4 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
mode = os.environ.get("OG_ATTN_PIPE", "default") + ("+p16" if os.environ.get("OG_ATTN_P16", "0") == "1" else "")


def run_case(Z, nq, nk, D, H, spike=False, reps=0, scales=(0.5, 2.0, 2.0)):
    g = torch.Generator().manual_seed(Z * 1000 + nk)
    q = (torch.randn(Z, nq, D, generator=g) * scales[0]).to(dev)
    k = (torch.randn(Z, nk, D, generator=g) * scales[1]).to(dev)
    v = (torch.randn(Z, nk, D, generator=g) * scales[2]).to(dev)
    if spike:          # one key far above the others late in the sequence: the running max must move by much more than 2^11 in a later tile
        k[:, (2 * nk) // 3] = q[:, 5] * 40.0
        k[:, nk - 1] = q[:, 7] * 90.0
    (qh, ql), (kh, kl), (vh, vl) = ops.split_f16(q), ops.split_f16(k), ops.split_f16(v)
    oh = torch.empty(Z, nq, D, device=dev, dtype=torch.float16); ol = torch.empty_like(oh)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D, oh.data_ptr(), ol.data_ptr(), D, Z, nq, nk, H, D // H, None, st)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    zs = sorted(set([0, Z // 2, Z - 1]))
    qq = (qh.double() + ql.double())[zs].view(len(zs), nq, H, -1).transpose(1, 2)
    kk = (kh.double() + kl.double())[zs].view(len(zs), nk, H, -1).transpose(1, 2)
    vv = (vh.double() + vl.double())[zs].view(len(zs), nk, H, -1).transpose(1, 2)
    s = qq @ kk.transpose(-1, -2)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    ref = ((p @ vv) / p.sum(-1, keepdim=True)).transpose(1, 2).reshape(len(zs), nq, D)
    err = ((oh.double() + ol.double())[zs] - ref).abs().max().item()
    line = f"[pipe={mode}] Z={Z} nq={nq} nk={nk} dh={D // H}{' spike' if spike else ''}: max |O - float64| = {err:.3e} (|O| max {ref.abs().max().item():.2f})"
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): run()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        fl = 4.0 * Z * nq * nk * D
        line += "  | " + " / ".join(f"{t:.1f}" for t in ts) + f" us per launch, best {fl / min(ts) / 1e6:.0f} TFLOP/s algorithmic"
    print(line, flush=True)
    return err


worst = 0.0
for nk in (1, 40, 64, 65, 128, 150, 192, 256, 300, 320, 333, 1000, 1024):
    worst = max(worst, run_case(3, 130, nk, 256, 4))
    worst = max(worst, run_case(2, 77, nk, 128, 4))
worst = max(worst, run_case(4, 256, 640, 256, 4, spike=True))
worst = max(worst, run_case(4, 256, 1000, 128, 4, spike=True))
worst = max(worst, run_case(8, 1024, 1024, 256, 4))
print(f"[pipe={mode}] worst error over the edge cases: {worst:.3e}")
if os.environ.get("OG_CHECK_NO_TIMING"):
    sys.exit(0)
run_case(64, 1024, 1024, 256, 4, reps=20)
run_case(32, 1024, 1024, 256, 4, reps=20)
run_case(64, 2048, 2048, 256, 4, reps=10)
run_case(16, 4096, 4096, 128, 4, reps=10)

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import ops
from oracle import superglue_oracle as orc
dev = torch.device("cuda:0")
for (H, dh, nq, nk) in [(4, 32, 64, 128), (4, 32, 64, 100), (4, 32, 300, 257), (4, 16, 300, 257), (4, 64, 300, 257), (2, 32, 40, 65), (4, 32, 64, 192)]:
    g = torch.Generator().manual_seed(1)
    D = H * dh
    q, k, v = torch.randn(2, nq, D, generator=g), torch.randn(2, nk, D, generator=g), torch.randn(2, nk, D, generator=g)
    ref = orc.softmax_attention(q.double(), k.double(), v.double(), H)
    out = ops.attention((q * dh ** -0.5).to(dev), k.to(dev), v.to(dev), H).cpu()
    d = (out.double() - ref).abs()
    bad = (d > 1e-3).nonzero()
    print(f"H={H} dh={dh} nq={nq} nk={nk}: max err {d.max():.2e}; bad {len(bad)}; first bad {bad[:3].tolist()}")

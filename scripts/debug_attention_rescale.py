import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from openglue_amd import ops
import numpy as np
dev = torch.device("cuda:0")
H, dh, nq, nk = 1, 64, 33, 128
g = torch.Generator().manual_seed(H * 100 + dh + nq)
def _rand(g, *shape, scale=1.0): return torch.randn(*shape, generator=g) * scale
D = H * dh
q, k, v = _rand(g, 2, nq, D, scale=3.0), _rand(g, 2, nk, D, scale=3.0), _rand(g, 2, nk, D, scale=2.0)
s = torch.einsum("bqd,bkd->bqk", q.double(), k.double()) * dh ** -0.5
ref = torch.softmax(s, -1) @ v.double()
out, lse = ops.attention((q * dh ** -0.5).to(dev), k.to(dev), v.to(dev), H, return_lse=True)
out = out.cpu().double(); lse = lse.cpu().double()
print("lse err", (lse[:, 0] - torch.logsumexp(s, -1)).abs()[0])
err = (out - ref).abs().amax(-1)
print("row errors", err)
s2 = s * 1.4426950408889634
m0 = s2[:, :, :64].amax(-1); m1 = s2[:, :, 64:].amax(-1)
print("tile0 max", m0[0]); print("tile1 max - tile0 max", (m1 - m0)[0])
bad = (err > 1e-3).nonzero()
print(bad[:10])

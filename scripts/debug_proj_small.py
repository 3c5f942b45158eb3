import sys, torch
sys.path.insert(0, "/root/repo")
from openglue_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for M, kw in ((8192, {}), (96, dict(cols_b=(0, 256))), (2048, dict(cols_b=(0, 512))), (2048, {})):
    x = torch.randn(M, 256, generator=g); w = torch.randn(768, 256, generator=g) * 0.05; b = torch.randn(768, generator=g)
    try:
        out = ops.proj_block(x.to(dev), w.to(dev), b.to(dev), **kw)
        torch.cuda.synchronize()
        ref = x.double() @ w.double().T + b.double()
        c = kw.get("cols_b", (0, 768))
        print(M, kw, "ok err", (out.cpu().double() - ref)[:, c[0]:c[1]].abs().max().item(), flush=True)
    except Exception as e:
        print(M, kw, "EXC", e, flush=True)

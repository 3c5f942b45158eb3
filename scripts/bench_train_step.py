#!/usr/bin/env python3
"""Time one training step (forward in train() mode + NLL loss + backward) of the BASELINE-config-2 model on HIP kernels.
Informational: the training path is functional, not tuned.  Round 4 captured the same step (with nll_static below as
the loss) into ONE hipGraph and replayed it: 37.8 ms against 39.2 ms eager (profiles/r04_train_step_graph.log) -- the ~2000 launches of a
step cost GPU-side time (kernel boundaries and dependent few-microsecond kernels), not host time; a graph does not remove that."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
from oracle import superglue_oracle as orc            # only nll_criterion (the loss is the caller's code in the reference too)

B = int(os.environ.get("B", 4)); N = int(os.environ.get("N", 1024)); IT = int(os.environ.get("ITERS", 20))
dev = torch.device("cuda:0")
cfg = syn.make_config(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=IT)
model = SuperGlue(cfg); model.load_state_dict(syn.make_state_dict(cfg, seed=0)); model = model.to(dev).train()
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.make_batch(B, N, N, 256, 1, seed=1).items()}
g = torch.Generator().manual_seed(0)
gt0 = torch.full((B, N), -1, dtype=torch.long); gt1 = torch.full((B, N), -1, dtype=torch.long)
for b in range(B):
    i = torch.randperm(N, generator=g)[: N // 2]; j = torch.randperm(N, generator=g)[: N // 2]
    gt0[b, i] = j; gt1[b, j] = i
gt0, gt1 = gt0.to(dev), gt1.to(dev)
def step():
    model.zero_grad(set_to_none=True)
    out = model(data)
    loss = orc.nll_criterion(out["scores"], gt0, gt1)
    loss.backward()
    return loss
def nll_static(scores):
    """utils/losses.py:7-53 (margin None) with masks instead of torch.where: no data-dependent shapes, no host synchronisation -- the
    form a captured step needs.  Same value as orc.nll_criterion."""
    inner = scores[:, :-1, :-1]
    m0 = (gt0 >= 0); u0 = (gt0 == -1); u1 = (gt1 == -1)
    picked = inner.gather(2, gt0.clamp_min(0)[:, :, None])[:, :, 0]
    per = lambda val, mask: (-(val * mask).sum(1) / mask.sum(1).clamp_min(1)).sum()
    return (per(picked, m0.float()) + 0.5 * (per(scores[:, :-1, -1], u0.float()) + per(scores[:, -1, :-1], u1.float()))) / scores.size(0)
step(); torch.cuda.synchronize()
t0 = time.time(); n = 3
for _ in range(n): loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"training step B={B} pairs x {N} kpts, 9 stages, {IT} Sinkhorn iterations: {dt * 1e3:.1f} ms per step ({B / dt:.1f} pairs/s); loss {loss.item():.4f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

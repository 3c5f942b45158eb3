#!/usr/bin/env python3
"""A training loop around the drop-in module, the way the reference's Lightning module drives it (models/matching_module.py:93-105 training_step,
:142-170 configure_optimizers: Adam, lr 1e-4 in config/config.yaml): `SuperGlue(config).train()`, the NLL of utils/losses.py:7-53 on the returned
`scores`, `loss.backward()`, `optimizer.step()`.  Everything between the input tensors and the parameter gradients runs on the HIP kernels of
openglue_amd (openglue_amd/train.py); the optimizer is torch's and updates the parameters in place.

    python examples/train_loop.py [--steps 20] [--pairs 4] [--kpts 512]     # needs an MI355X; synthetic pairs with known correspondences"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue


def nll(scores, gt0, gt1):
    """utils/losses.py:7-53 with margin None, written with masks (no data-dependent shapes): matched pairs, unmatched keypoints of either side"""
    B = scores.size(0)
    m0, u0, u1 = (gt0 >= 0).float(), (gt0 == -1).float(), (gt1 == -1).float()
    per = lambda val, mask: (-(val * mask).sum(1) / mask.sum(1).clamp_min(1)).sum()
    picked = scores[:, :-1, :-1].gather(2, gt0.clamp_min(0)[:, :, None])[:, :, 0]
    return (per(picked, m0) + 0.5 * (per(scores[:, :-1, -1], u0) + per(scores[:, -1, :-1], u1))) / B


def make_pairs(B, N, D, dev, seed=0):
    """image 1 = a shuffled, jittered copy of 60 % of image 0's keypoints and descriptors + fresh points: the correspondences are known"""
    g = torch.Generator().manual_seed(seed)
    data = syn.make_batch(B, N, N, D, 1, seed=seed)
    gt0 = torch.full((B, N), -1, dtype=torch.long); gt1 = torch.full((B, N), -1, dtype=torch.long)
    for b in range(B):
        keep = torch.randperm(N, generator=g)[: int(0.6 * N)]
        dst = torch.randperm(N, generator=g)[: keep.numel()]
        data["keypoints1"][b, dst] = data["keypoints0"][b, keep] + 2.0 * torch.randn(keep.numel(), 2, generator=g)
        d = data["local_descriptors0"][b, keep] + 0.05 * torch.randn(keep.numel(), D, generator=g)
        data["local_descriptors1"][b, dst] = d / d.norm(dim=-1, keepdim=True)
        gt0[b, keep] = dst; gt1[b, dst] = keep
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}, gt0.to(dev), gt1.to(dev)


def run(steps=20, pairs=4, kpts=512, dim=256, stages=9, lr=1e-4, log=print):
    dev = torch.device("cuda:0")
    cfg = syn.make_config(descriptor_dim=dim, num_stages=stages, num_heads=4, num_iters=20)
    model = SuperGlue(cfg)
    model.load_state_dict(syn.make_state_dict(cfg, seed=0))
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    data, gt0, gt1 = make_pairs(pairs, kpts, dim, dev)
    losses = []
    t0 = time.perf_counter()
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = nll(model(data)["scores"], gt0, gt1)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
        if s % 5 == 0 or s + 1 == steps:
            log(f"step {s:3d}  loss {losses[-1]:.4f}")
    torch.cuda.synchronize(dev)
    log(f"{steps} steps of {pairs} pairs x {kpts} keypoints: {(time.perf_counter() - t0) / steps * 1e3:.1f} ms per step incl. Adam and the loss read-back")
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--pairs", type=int, default=4); ap.add_argument("--kpts", type=int, default=512)
    a = ap.parse_args()
    run(a.steps, a.pairs, a.kpts)

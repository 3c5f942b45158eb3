#!/usr/bin/env python3
"""CPU emulation (float64): how much of an error injected into the exponentials of the EARLY Sinkhorn iterations survives to the
log-scores after the last iteration?  (The idea: a cheap low-accuracy exp for iterations 1..T0, the exact one for the rest --
Sinkhorn contracts towards its fixed point, but the reference's output after `iters` iterations is a point of the trajectory,
not the fixed point.)  Score matrices: the C2-shaped synthetic problem of the bench (seeded), produced by the oracle's GNN."""
import os, sys, math
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import superglue_oracle as orc
from openglue_amd import synthetic as syn

torch.set_num_threads(16)


def sinkhorn_noisy(S, dustbin, iters, reg, t0, rel, seed=0):
    """matching_log_probs with multiplicative noise (1 + U(-rel, rel)) on every exponential of iterations < t0."""
    g = torch.Generator().manual_seed(seed)
    B, m, n = S.shape
    Sa = torch.empty(B, m + 1, n + 1, dtype=torch.float64)
    Sa[:, :m, :n] = S; Sa[:, m, :] = dustbin; Sa[:, :, n] = dustbin
    Sa = Sa / reg
    norm = -math.log(m + n)
    la = torch.full((B, m + 1), norm, dtype=torch.float64); la[:, -1] += math.log(n)
    lb = torch.full((B, n + 1), norm, dtype=torch.float64); lb[:, -1] += math.log(m)
    u, v = torch.zeros_like(la), torch.zeros_like(lb)

    def lse(x, dim, noisy):
        mx = x.amax(dim, keepdim=True)
        e = torch.exp(x - mx)
        if noisy:
            e = e * (1.0 + (torch.rand(e.shape, generator=g, dtype=torch.float64) * 2 - 1) * rel)
        return (mx + torch.log(e.sum(dim, keepdim=True))).squeeze(dim)

    for it in range(iters):
        u = la - lse(Sa + v[:, None, :], 2, it < t0)
        v = lb - lse(Sa + u[:, :, None], 1, it < t0)
    return Sa + u[:, :, None] + v[:, None, :] - norm


def main():
    cases = []
    cfg = syn.make_config()                                       # C2 model
    sd = {k: v.double() if v.dtype.is_floating_point else v for k, v in syn.make_state_dict(cfg, seed=0).items()}
    data = syn.make_batch(2, 1024, 1024, 256, 1, seed=1)
    data = {k: (v.double() if torch.is_tensor(v) and v.dtype.is_floating_point else v) for k, v in data.items()}
    with torch.no_grad():
        out = orc.superglue_forward(sd, cfg, data, return_intermediates=True) if "return_intermediates" in orc.superglue_forward.__code__.co_varnames else None
    if out is not None and "S" in out:
        S = out["S"]
    else:                                                        # recompute S from the context descriptors
        with torch.no_grad():
            o = orc.superglue_forward(sd, cfg, data)
        g0, g1 = o["context_descriptors0"].transpose(1, 2), o["context_descriptors1"].transpose(1, 2)
        S = g0 @ g1.transpose(1, 2) * 256 ** -0.5
    z = float(sd["dustbin_score"])
    print("S range", float(S.min()), float(S.max()))
    exact = sinkhorn_noisy(S, z, 100, 1.0, 0, 0.0)
    for rel in (1e-3, 1e-4):
        for t0 in (50, 80, 90, 95, 99, 100):
            got = sinkhorn_noisy(S, z, 100, 1.0, t0, rel)
            print(f"rel {rel:g}  noisy iterations 1..{t0}: max |d scores| = {float((got - exact).abs().max()):.3e}")


if __name__ == "__main__":
    main()

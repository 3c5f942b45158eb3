#!/usr/bin/env python3
"""CPU emulation (fp32 arithmetic, float64 reference) of the LINEAR-DOMAIN form of the on-chip-resident Sinkhorn iterations
(csrc/sinkhorn_resident2.hip): after the max-subtracted first iteration the plan entries E_ij = 2^(s_ij + u_i + v_j) are <= 1, and
instead of re-evaluating the exponential of every entry in every iteration the resident matrix is E itself, rescaled in place:
    pass 1:  E_ij *= g_j ;  rowsum_i = sum_j E_ij (+ the dustbin-column entry in closed form) ;  du_i = log2 a_i - log2 rowsum_i
    pass 2:  E_ij *= 2^du_i ;  colsum_j = sum_i E_ij (+ the dustbin-row entry) ;  dv_j = log2 b_j - log2 colsum_j ;  g_j = 2^dv_j
Same recursion as optimal_transport.py:24-26 in exact arithmetic; here the question is what fp32 rounding of the in-place products
(two per entry and iteration, never re-synchronised with the duals) does to the log-scores after 100 iterations, and what happens
to entries that underflow.  Duals are accumulated as u = u1 + sum du (the increments in their own accumulator).

Usage: python tests/emulate_sinkhorn_linear.py [--c2]     (--c2 adds the C2-shaped synthetic problem through the oracle's GNN)"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import superglue_oracle as orc

LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453
F = torch.float32


def first_iteration_fp32(S, z, reg):
    """u1, v1 (natural units, fp32) after one max-subtracted iteration, dustbins included; returns u [B,m+1], v [B,n+1]."""
    B, m, n = S.shape
    Sa = torch.empty(B, m + 1, n + 1, dtype=F)
    Sa[:, :m, :n] = S; Sa[:, m, :] = z; Sa[:, :, n] = z
    Sa = Sa / reg
    norm = -math.log(m + n)
    la = torch.full((B, m + 1), norm, dtype=F); la[:, -1] += math.log(n)
    lb = torch.full((B, n + 1), norm, dtype=F); lb[:, -1] += math.log(m)
    u = la - torch.logsumexp(Sa, dim=2)
    v = lb - torch.logsumexp(Sa + u[:, :, None], dim=1)
    return u, v, la, lb, Sa


def linear_sinkhorn(S, z, iters, reg, refresh_every=0, flush_denormals=True, drift_refresh_bits=0.0):
    """fp32 emulation; returns scores [B, m+1, n+1] (float32), and diagnostics."""
    B, m, n = S.shape
    u1, v1, la, lb, Sa = first_iteration_fp32(S, z, reg)
    if iters == 1:
        return (Sa + u1[:, :, None] + v1[:, None, :] + math.log(m + n)), {}
    tiny = torch.finfo(F).tiny
    s2 = (Sa[:, :m, :n] * LOG2E).to(F)                       # base-2 scores
    zr2 = np.float32(z / reg * LOG2E)
    la2, lb2 = la * LOG2E, lb * LOG2E
    u0, v0 = (u1 * LOG2E).to(F), (v1 * LOG2E).to(F)          # base-2 duals at hand-over
    Du, Dv = torch.zeros_like(u0), torch.zeros_like(v0)      # accumulated increments
    def fresh():
        E = torch.exp2((s2 + (v0 + Dv)[:, None, :n]) + (u0 + Du)[:, :m, None])
        if flush_denormals: E = torch.where(E < tiny, torch.zeros_like(E), E)
        return E
    E = fresh()
    g = torch.ones(B, n, dtype=F)
    drift = torch.zeros(B, dtype=F)                          # bound on the growth (bits) of any entry since its last refresh
    n_refresh = 0
    min_live = float("inf")
    for it in range(iters - 1):
        u, v = u0 + Du, v0 + Dv
        # pass 1
        E = E * g[:, None, :]
        if flush_denormals: E = torch.where(E < tiny, torch.zeros_like(E), E)
        pd = torch.exp2(zr2 + v[:, n:n + 1] + u[:, :m])       # dustbin-column entries of the rows (old u, current v_N)
        rowsum = E.sum(dim=2) + pd
        du = la2[:, :m] - torch.log2(rowsum)
        f = torch.exp2(du)
        # dustbin row: u_M' = log2 a_M - (z + LSE2 v)
        uM = la2[:, m] - (zr2 + torch.logsumexp(v * LN2, dim=1) * LOG2E)
        # pass 2
        E = E * f[:, :, None]
        if flush_denormals: E = torch.where(E < tiny, torch.zeros_like(E), E)
        colsum = E.sum(dim=1) + torch.exp2(zr2 + v[:, :n] + uM[:, None])
        dcolsum = (pd * f).sum(dim=1) + torch.exp2(zr2 + v[:, n] + uM)
        dv = lb2[:, :n] - torch.log2(colsum)
        dvN = lb2[:, n] - torch.log2(dcolsum)
        g = torch.exp2(dv)
        Du[:, :m] += du
        Du[:, m] = uM - u0[:, m]
        Dv[:, :n] += dv
        Dv[:, n] += dvN
        drift += du.clamp_min(0).amax(dim=1) + dv.clamp_min(0).amax(dim=1)
        do_refresh = (refresh_every and (it + 1) % refresh_every == 0) or (drift_refresh_bits and bool((drift > drift_refresh_bits).any()))
        if do_refresh:
            E = fresh()
            g = torch.ones(B, n, dtype=F)
            drift.zero_()
            n_refresh += 1
    u, v = (u0 + Du) * LN2, (v0 + Dv) * LN2
    scores = (Sa + u[:, :, None]) + v[:, None, :] + math.log(m + n)
    return scores, {"refreshes": n_refresh, "max_Du": float(Du.abs().max()), "max_Dv": float(Dv.abs().max()), "zeros": float((E == 0).float().mean())}


def lazy_sinkhorn(S, z, iters, reg, drift_bits=40.0):
    """The form the kernel runs since round 4: E stays as evaluated; the row factors F_i = 2^(sum du_i) and column factors
    C_j = 2^(sum dv_j) since the evaluation are kept beside it and applied inside the sums (2 fmas per entry and iteration, no
    store).  Two-sided drift bound: re-evaluate when sum_t (max_i |du_i| + max_j |dv_j|) exceeds drift_bits."""
    B, m, n = S.shape
    u1, v1, la, lb, Sa = first_iteration_fp32(S, z, reg)
    if iters == 1:
        return (Sa + u1[:, :, None] + v1[:, None, :] + math.log(m + n)), {}
    tiny = torch.finfo(F).tiny
    s2 = (Sa[:, :m, :n] * LOG2E).to(F)
    zr2 = np.float32(z / reg * LOG2E)
    la2, lb2 = la * LOG2E, lb * LOG2E
    u, v = (u1 * LOG2E).to(F), (v1 * LOG2E).to(F)
    def fresh():
        E = torch.exp2((s2 + v[:, None, :n]) + u[:, :m, None])
        return torch.where(E < tiny, torch.zeros_like(E), E)
    E = fresh(); Fr = torch.ones(B, m, dtype=F); C = torch.ones(B, n, dtype=F)
    drift = torch.zeros(B, dtype=F); nref = 0
    for it in range(iters - 1):
        pd = torch.exp2(zr2 + v[:, n:n + 1] + u[:, :m])
        rowsum = Fr * (E * C[:, None, :]).sum(dim=2) + pd
        un = u[:, :m] + la2[:, :m] - torch.log2(rowsum)
        du = un - u[:, :m]
        f = torch.exp2(du)
        Fr = Fr * f
        uM = la2[:, m] - (zr2 + torch.logsumexp(v * LN2, dim=1) * LOG2E)
        T = (E * Fr[:, :, None]).sum(dim=1)
        colsum = C * T + torch.exp2(zr2 + v[:, :n] + uM[:, None])
        dcolsum = (pd * f).sum(dim=1) + torch.exp2(zr2 + v[:, n] + uM)
        vn = v[:, :n] + lb2[:, :n] - torch.log2(colsum)
        dv = vn - v[:, :n]
        C = C * torch.exp2(dv)
        u = torch.cat([un, uM[:, None]], dim=1)
        v = torch.cat([vn, (v[:, n] + lb2[:, n] - torch.log2(dcolsum))[:, None]], dim=1)
        drift += du.abs().amax(dim=1) + dv.abs().amax(dim=1)
        if bool((drift > drift_bits).any()):
            E = fresh(); Fr = torch.ones_like(Fr); C = torch.ones_like(C); drift.zero_(); nref += 1
    un, vn = u * LN2, v * LN2
    return (Sa + un[:, :, None]) + vn[:, None, :] + math.log(m + n), {"refreshes": nref}


def rand_scores(B, m, n, scale, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, m, n, generator=g, dtype=F) * scale


def report(tag, S, z, iters, reg, **kw):
    ref = orc.matching_log_probs(S.double(), torch.tensor(z, dtype=torch.float64), iters, reg)
    if not kw:
        lz, li = lazy_sinkhorn(S, z, iters, reg)
        print(f"[{tag}] it={iters} reg={reg} LAZY factors (the kernel's form): err {(lz.double() - ref).abs().max().item():.2e} {li}")
    out, info = linear_sinkhorn(S, z, iters, reg, **kw)
    err = (out.double() - ref).abs().max().item()
    # the log-domain fp32 solver of the reference itself as the yardstick
    ref32 = orc.matching_log_probs(S, torch.tensor(z, dtype=F), iters, reg)
    e32 = (ref32.double() - ref).abs().max().item()
    print(f"[{tag}] it={iters} reg={reg} {kw or ''}: linear-form err {err:.2e} | reference fp32 log-domain err {e32:.2e} | max|score| {ref.abs().max():.0f} | {info}")
    return err


def main():
    report("rand 4.0  257x1000", rand_scores(2, 257, 1000, 4.0, 1), 0.7, 10, 1.0)
    report("rand 4.0  512x512 ", rand_scores(2, 512, 512, 4.0, 2), 0.7, 100, 1.0)
    report("rand 4.0  1024x1024", rand_scores(1, 1024, 1024, 4.0, 3), 0.7, 100, 1.0)
    for (scale, reg, z) in [(25.0, 0.5, 0.7), (8.0, 0.1, -30.0), (60.0, 1.0, 50.0), (1e-3, 1.0, 0.0)]:
        S = rand_scores(2, 96, 200, scale, int(scale * 10) + 3)
        S[0, 5, :] = -4.0 * scale
        S[1, :, 7] = 4.0 * scale
        for kw in ({}, {"drift_refresh_bits": 40.0}):
            report(f"extreme scale={scale} z={z}", S, z, 30, reg, **kw)
    S = rand_scores(2, 200, 900, 25.0, 5); S[0, 5, :] = -100.0; S[1, :, 7] = 100.0
    for kw in ({}, {"drift_refresh_bits": 40.0}):
        report("extreme resident test", S, 0.7, 40, 0.5, **kw)
    # slowly converging: a permutation-like cost with a conflict chain (mass has to travel along it)
    m = 256
    S = torch.full((1, m, m), -40.0, dtype=F)
    idx = torch.arange(m)
    S[0, idx, idx] = 0.0
    S[0, idx[:-1], idx[1:]] = 0.5
    for kw in ({}, {"drift_refresh_bits": 40.0}, {"refresh_every": 16}):
        report("conflict chain", S, -20.0, 100, 1.0, **kw)
    if "--c2" in sys.argv:
        from openglue_amd import synthetic as syn
        torch.set_num_threads(8)
        cfg = syn.make_config()
        sd = syn.make_state_dict(cfg, seed=0)
        data = syn.make_batch(1, 1024, 1024, 256, 1, seed=1)
        with torch.no_grad():
            o = orc.superglue_forward(sd, cfg, data)
        g0, g1 = o["context_descriptors0"].transpose(1, 2), o["context_descriptors1"].transpose(1, 2)
        S = (g0 @ g1.transpose(1, 2) * 256 ** -0.5).to(F)
        print("C2 synthetic S range", float(S.min()), float(S.max()))
        for kw in ({}, {"refresh_every": 32}):
            report("C2 synthetic", S, float(sd["dustbin_score"]), 100, 1.0, **kw)


if __name__ == "__main__":
    main()

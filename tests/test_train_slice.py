"""Training slice: the optimal-transport layer with HIP forward + backward against gradients produced by
the REFERENCE under torch autograd (tests/golden/train_ot.npz, written by tests/golden/make_golden_train.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import superglue_oracle as orc
from tests.util import GOLDEN

Z = dict(np.load(os.path.join(GOLDEN, "train_ot.npz")))
CASES = ["a", "b", "c"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_nll_and_autograd_match_the_reference(name):
    """CPU: the restated criterion + the oracle's Sinkhorn under autograd reproduce the reference's loss and gradients."""
    B, m, n, iters = (int(v) for v in Z[f"{name}_meta"]); reg, z = (float(v) for v in Z[f"{name}_reg_z"])
    S = torch.from_numpy(Z[f"{name}_S"]).clone().requires_grad_(True)
    dust = torch.tensor(z, requires_grad=True)
    scores = orc.matching_log_probs(S, dust, iters, reg)
    assert np.abs(scores.detach().numpy() - Z[f"{name}_scores"]).max() < 2e-5
    loss = orc.nll_criterion(scores, torch.from_numpy(Z[f"{name}_gt0"]), torch.from_numpy(Z[f"{name}_gt1"]))
    loss.backward()
    assert abs(loss.item() - float(Z[f"{name}_nll"])) < 1e-5
    assert np.abs(S.grad.numpy() - Z[f"{name}_dS_nll"]).max() < 1e-6
    assert abs(dust.grad.item() - float(Z[f"{name}_dz_nll"])) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_sinkhorn_backward_against_reference_autograd(gpu_device, name):
    from openglue_amd.train import matching_log_probs
    B, m, n, iters = (int(v) for v in Z[f"{name}_meta"]); reg, z = (float(v) for v in Z[f"{name}_reg_z"])
    for kind in ("dense", "nll"):
        S = torch.from_numpy(Z[f"{name}_S"]).to(gpu_device).requires_grad_(True)
        dust = torch.tensor(z, device=gpu_device, requires_grad=True)
        scores = matching_log_probs(S, dust, iters, reg)
        assert np.abs(scores.detach().cpu().numpy() - Z[f"{name}_scores"]).max() < 1e-4
        if kind == "dense":
            loss = (scores * torch.from_numpy(Z[f"{name}_R"]).to(gpu_device)).sum()
        else:
            loss = orc.nll_criterion(scores, torch.from_numpy(Z[f"{name}_gt0"]).to(gpu_device), torch.from_numpy(Z[f"{name}_gt1"]).to(gpu_device))
            assert abs(loss.item() - float(Z[f"{name}_nll"])) < 1e-4
        loss.backward()
        want_dS, want_dz = Z[f"{name}_dS_{kind}"], float(Z[f"{name}_dz_{kind}"])
        got_dS, got_dz = S.grad.cpu().numpy(), dust.grad.item()
        scale = np.abs(want_dS).max()
        err = np.abs(got_dS - want_dS).max()
        print(f"[train_ot {name} {kind}] dS err {err:.2e} (max |dS| {scale:.2e}); d dustbin {got_dz:.6f} vs {want_dz:.6f}")
        assert err < 1e-3 * scale + 1e-7                         # parameter / input gradients to rel. 1e-3
        assert abs(got_dz - want_dz) < 1e-3 * abs(want_dz) + 1e-6


@pytest.mark.gpu
def test_sinkhorn_backward_c2_sized(gpu_device):
    """One 1024 x 1024 pair, 100 iterations (the BASELINE config 2 Sinkhorn): HIP backward vs the oracle under autograd on the CPU."""
    from openglue_amd.train import matching_log_probs
    g = torch.Generator().manual_seed(3)
    B, m, n, iters = 1, 1024, 1024, 100
    S0 = torch.randn(B, m, n, generator=g) * 3.0
    gt0 = torch.full((B, m), -1, dtype=torch.long); gt1 = torch.full((B, n), -1, dtype=torch.long)
    i = torch.randperm(m, generator=g)[:600]; j = torch.randperm(n, generator=g)[:600]
    gt0[0, i] = j; gt1[0, j] = i
    S = S0.clone().requires_grad_(True); dust = torch.tensor(1.0, requires_grad=True)
    ref = orc.nll_criterion(orc.matching_log_probs(S, dust, iters, 1.0), gt0, gt1)
    ref.backward()
    Sg = S0.to(gpu_device).requires_grad_(True); dg = torch.tensor(1.0, device=gpu_device, requires_grad=True)
    loss = orc.nll_criterion(matching_log_probs(Sg, dg, iters, 1.0), gt0.to(gpu_device), gt1.to(gpu_device))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-3
    scale = S.grad.abs().max().item()
    err = (Sg.grad.cpu() - S.grad).abs().max().item()
    print(f"[train_ot C2-sized] nll {loss.item():.5f} vs {ref.item():.5f}; dS err {err:.2e} (max {scale:.2e}); d dustbin {dg.grad.item():.6f} vs {dust.grad.item():.6f}")
    assert err < 1e-3 * scale
    assert abs(dg.grad.item() - dust.grad.item()) < 1e-3 * abs(dust.grad.item()) + 1e-6

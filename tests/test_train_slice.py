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


# ---------------------------------------------------------------------------------------------------------------
# train-mode BatchNorm / FeedForwardNet forward (tests/golden/train_mlp.npz: the reference's FeedForwardNet in train() mode)
M = dict(np.load(os.path.join(GOLDEN, "train_mlp.npz")))
MLP_CASES = ["enc", "msg"]


def _mlp_state(name, which, device=None):
    pre = f"{name}_{which}_"
    sd = {k[len(pre):]: torch.from_numpy(v.copy()) for k, v in M.items() if k.startswith(pre)}
    return {k: (v.to(device) if device is not None else v) for k, v in sd.items()}


@pytest.mark.parametrize("name", MLP_CASES)
def test_oracle_train_mode_mlp_matches_the_reference(name):
    """CPU: the restated train-mode BatchNorm inside FeedForwardNet reproduces the reference's outputs step by step and its
    running statistics after the last step."""
    meta = [int(v) for v in M[f"{name}_meta"]]
    sizes, (B, N, steps) = meta[:-3], meta[-3:]
    sd = _mlp_state(name, "before")
    for st in range(steps):
        x = torch.from_numpy(M[f"{name}_x{st}"]).permute(0, 2, 1)                   # [B, C, N] -> token-major [B, N, C]
        y, new_stats = orc.feed_forward_train(x, sd, "", len(sizes) - 1)
        assert np.abs(y.permute(0, 2, 1).numpy() - M[f"{name}_y{st}"]).max() < 2e-5
        sd.update(new_stats)
    after = _mlp_state(name, "after")
    for k, v in after.items():
        if "running" in k:
            assert np.abs(sd[k].numpy() - v.numpy()).max() < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("name", MLP_CASES)
def test_train_mode_mlp_forward_against_reference(gpu_device, name):
    """HIP: exact-fp32 GEMM (+bias, ReLU) -> og_batchnorm_train_forward chain on token-major activations vs the reference's
    FeedForwardNet in train() mode; the running statistics are updated in place like torch's."""
    from openglue_amd.train import feed_forward_train
    meta = [int(v) for v in M[f"{name}_meta"]]
    sizes, (B, N, steps) = meta[:-3], meta[-3:]
    sd = _mlp_state(name, "before", gpu_device)
    for st in range(steps):
        x = torch.from_numpy(M[f"{name}_x{st}"]).to(gpu_device)                       # [B, C, N]
        xt = x.permute(0, 2, 1).reshape(B * N, sizes[0]).contiguous()                  # token-major
        y = feed_forward_train(xt, sd).reshape(B, N, sizes[-1]).permute(0, 2, 1)
        err = (y.cpu() - torch.from_numpy(M[f"{name}_y{st}"])).abs().max().item()
        print(f"[train_mlp {name} step {st}] y err {err:.2e} (|y| up to {np.abs(M[f'{name}_y{st}']).max():.1f})")
        assert err < 1e-4
    after = _mlp_state(name, "after")
    for k, v in after.items():
        if "running" in k:
            e = (sd[k].cpu() - v).abs().max().item()
            assert e < 1e-5, (k, e)


@pytest.mark.gpu
def test_batchnorm_train_large_and_offset_channels(gpu_device):
    """65536 tokens x 256 channels (the C2 activation shape), channels whose mean is 1000x their spread: the shifted sums keep
    the variance; statistics against float64."""
    from openglue_amd.train import batch_norm_train
    g = torch.Generator().manual_seed(5)
    T, C = 65536, 256
    x = torch.randn(T, C, generator=g)
    x[:, :16] = x[:, :16] * 1e-2 + 10.0                      # mean >> std
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    xd = x.to(gpu_device)
    rmd, rvd = rm.to(gpu_device), rv.to(gpu_device)
    y, mean, invstd = batch_norm_train(xd, w.to(gpu_device), b.to(gpu_device), rmd, rvd, 0.1, 1e-5, return_stats=True)
    x64 = x.double()
    m64, v64 = x64.mean(0), x64.var(0, unbiased=False)
    want = (x64 - m64) / torch.sqrt(v64 + 1e-5) * w.double() + b.double()
    assert (mean.cpu().double() - m64).abs().max() < 1e-5
    assert ((invstd.cpu().double() - 1 / torch.sqrt(v64 + 1e-5)) * torch.sqrt(v64 + 1e-5)).abs().max() < 1e-4
    err = (y.cpu().double() - want).abs().max().item()
    print(f"[batchnorm_train 65536 x 256] y err {err:.2e}")
    assert err < 2e-3                                         # offset channels: (x - mean) carries fp32 rounding of x itself (1e-6 * 10 / 1e-2)
    assert (y.cpu().double()[:, 16:] - want[:, 16:]).abs().max() < 2e-5
    assert (rmd.cpu().double() - 0.1 * m64).abs().max() < 1e-5
    assert (rvd.cpu().double() - (0.9 + 0.1 * x64.var(0, unbiased=True))).abs().max() < 1e-5


def _grad_case(name, device=None):
    meta = [int(v) for v in M[f"{name}_meta"]]
    sizes, (B, N, steps) = meta[:-3], meta[-3:]
    sd = _mlp_state(name, "after", device)                   # parameters are the same before / after; only running statistics moved
    x = torch.from_numpy(M[f"{name}_gx"]).permute(0, 2, 1).contiguous()       # token-major [B, N, C]
    R = torch.from_numpy(M[f"{name}_gR"]).permute(0, 2, 1).contiguous()
    return sizes, B, N, sd, x, R


@pytest.mark.parametrize("name", MLP_CASES)
def test_oracle_train_mode_mlp_gradients_match_the_reference(name):
    """CPU: autograd through the restated train-mode block reproduces the reference's gradients (input and every parameter)."""
    sizes, B, N, sd, x, R = _grad_case(name)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    full = dict(sd); full.update(params)
    xr = x.clone().requires_grad_(True)
    y, _ = orc.feed_forward_train(xr, full, "", len(sizes) - 1)
    assert np.abs(y.detach().permute(0, 2, 1).numpy() - M[f"{name}_gy"]).max() < 2e-5
    (y * R).sum().backward()
    want = M[f"{name}_grad_x"]
    assert np.abs(xr.grad.permute(0, 2, 1).numpy() - want).max() < 1e-4 * np.abs(want).max() + 1e-6
    for k, p in params.items():
        want = M[f"{name}_grad_{k}"]
        assert np.abs(p.grad.numpy() - want).max() < 1e-4 * np.abs(want).max() + 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("name", MLP_CASES)
def test_train_mode_mlp_backward_against_reference_autograd(gpu_device, name):
    """HIP: conv / ReLU / train-mode BatchNorm backward (exact-fp32 GEMMs, og_batchnorm_train_backward) vs the gradients the
    reference's FeedForwardNet produced under torch autograd."""
    from openglue_amd.train import feed_forward_train_autograd
    sizes, B, N, sd, x, R = _grad_case(name, gpu_device)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if "running" in k}
    xr = x.reshape(B * N, sizes[0]).to(gpu_device).requires_grad_(True)
    y = feed_forward_train_autograd(xr, params, buffers)
    err_y = (y.detach().reshape(B, N, sizes[-1]).permute(0, 2, 1).cpu() - torch.from_numpy(M[f"{name}_gy"])).abs().max().item()
    assert err_y < 1e-4
    (y * R.reshape(B * N, sizes[-1]).to(gpu_device)).sum().backward()
    want = M[f"{name}_grad_x"]
    got = xr.grad.reshape(B, N, sizes[0]).permute(0, 2, 1).cpu().numpy()
    worst = np.abs(got - want).max() / np.abs(want).max()
    for k, p in params.items():
        want = M[f"{name}_grad_{k}"]
        e = np.abs(p.grad.cpu().numpy().reshape(want.shape) - want).max() / max(np.abs(want).max(), 1e-6)
        worst = max(worst, e)
        assert e < 1e-3, (k, e)                                  # gradients to rel. 1e-3
    print(f"[train_mlp {name} backward] forward err {err_y:.2e}; worst relative gradient error {worst:.2e}")
    assert worst < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# the whole module in training mode (tests/golden/train_model.npz: the reference SuperGlue in train(), NLL loss, backward)
G = dict(np.load(os.path.join(GOLDEN, "train_model.npz")))
MODEL_CASES = {"base": dict(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=8),
               "flags": dict(descriptor_dim=64, num_stages=1, num_heads=2, num_iters=5, use_offset=True, residual=True)}


def _model_case(name):
    from openglue_amd import synthetic as syn
    cfg = syn.make_config(**MODEL_CASES[name])
    sd = syn.make_state_dict(cfg, seed=len(name))
    B, m, n = (int(v) for v in G[f"{name}_meta"])
    data = syn.make_batch(B, m, n, cfg["descriptor_dim"], 1, seed=3 + len(name))
    return cfg, sd, data, torch.from_numpy(G[f"{name}_gt0"]), torch.from_numpy(G[f"{name}_gt1"])


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_oracle_training_step_matches_the_reference(name):
    """CPU: the oracle in training mode (batch-statistics BatchNorm) under autograd reproduces the reference's scores, loss,
    parameter / descriptor gradients and updated running statistics."""
    cfg, sd, data, gt0, gt1 = _model_case(name)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    data["local_descriptors0"].requires_grad_(True); data["local_descriptors1"].requires_grad_(True)
    stats = {}
    out = orc.superglue_forward(params, cfg, data, train_stats=stats)
    assert np.abs(out["scores"].detach().numpy() - G[f"{name}_scores"]).max() < 1e-4
    loss = orc.nll_criterion(out["scores"], gt0, gt1)
    assert abs(loss.item() - float(G[f"{name}_loss"])) < 1e-3
    loss.backward()
    for key, got in (("desc0", data["local_descriptors0"].grad), ("desc1", data["local_descriptors1"].grad)):
        want = G[f"{name}_grad_{key}"]
        assert np.abs(got.numpy() - want).max() < 1e-3 * np.abs(want).max() + 1e-7, key
    checked = 0
    for k, p in params.items():
        if f"{name}_grad_{k}" in G and p.requires_grad:
            want = G[f"{name}_grad_{k}"]
            got = p.grad.numpy() if p.grad is not None else np.zeros_like(want)
            assert np.abs(got - want).max() < 1e-3 * np.abs(want).max() + 1e-6, k
            checked += 1
    assert checked >= 40
    for k, v in stats.items():
        assert np.abs(v.detach().numpy() - G[f"{name}_buf_{k}"]).max() < 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_training_step_against_reference_autograd(gpu_device, name):
    """HIP: SuperGlue(config).train() forward + loss.backward() -- every GEMM, BatchNorm, softmax, Sinkhorn forward/backward kernel
    of openglue_amd.train -- vs the reference's training step: scores, loss, gradients of every parameter and of the descriptors,
    BatchNorm running statistics after the step."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1 = _model_case(name)
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).train()
    dd = {k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}
    dd["local_descriptors0"].requires_grad_(True); dd["local_descriptors1"].requires_grad_(True)
    out = model(dd)
    err_s = np.abs(out["scores"].detach().cpu().numpy() - G[f"{name}_scores"]).max()
    assert err_s < 1e-3
    assert np.abs(out["context_descriptors0"].detach().cpu().numpy() - G[f"{name}_ctx0"]).max() < 1e-4
    loss = orc.nll_criterion(out["scores"], gt0.to(gpu_device), gt1.to(gpu_device))
    assert abs(loss.item() - float(G[f"{name}_loss"])) < 1e-3 * abs(float(G[f"{name}_loss"]))
    loss.backward()
    worst, worst_k = 0.0, ""
    for key, got in (("desc0", dd["local_descriptors0"].grad), ("desc1", dd["local_descriptors1"].grad)):
        want = G[f"{name}_grad_{key}"]
        e = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        if e > worst: worst, worst_k = e, key
    n_checked = 0
    for k, p in model.named_parameters():
        want = G[f"{name}_grad_{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-8)
        if np.abs(want).max() > 1e-7 and e > worst: worst, worst_k = e, k
        n_checked += 1
    print(f"[train_model {name}] scores err {err_s:.2e}; loss {loss.item():.5f} vs {float(G[f'{name}_loss']):.5f}; "
          f"{n_checked} parameter gradients, worst relative error {worst:.2e} ({worst_k})")
    assert worst < 1e-3                                      # parameter gradients to rel. 1e-3
    for k, b in model.named_buffers():
        if "running" in k:
            assert np.abs(b.cpu().numpy() - G[f"{name}_buf_{k}"]).max() < 1e-5, k


# ----------------------------------------------------------------------------- the training step at the size it is benchmarked on
GC2 = np.load(os.path.join(GOLDEN, "train_c2.npz"))


def _c2_case():
    from openglue_amd import synthetic as syn
    cfg = syn.make_config(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=20)
    B, m, n = (int(v) for v in GC2["meta"])
    return cfg, syn.make_state_dict(cfg, seed=0), syn.make_batch(B, m, n, 256, 1, seed=1), torch.from_numpy(GC2["gt0"]), torch.from_numpy(GC2["gt1"])


C2_TOL = {"norm": 1e-3, "max": 5e-3, "head": 2e-2}


def _check_c2_gradients(named_grads):
    """per parameter: L2 norm, largest |gradient| and the first 64 entries (relative to the largest) against the digests the reference left in
    train_c2.npz.  Tolerances C2_TOL: two fp32 CPU implementations of this step (the reference and the oracle) already differ by 1.9e-4 /
    1.3e-3 / 6.5e-3 in these three measures -- bias gradients are sums over 8192 tokens that cancel to a thousandth of their terms.  Gradients
    that are rounding noise in the reference too (|g| < 1e-7: the k biases -- softmax does not see a per-query constant) are skipped."""
    worst = {"norm": (0.0, ""), "max": (0.0, ""), "head": (0.0, "")}
    checked = 0
    for k, g in named_grads:
        gmax = float(GC2[f"gmax_{k}"])
        if gmax < 1e-7:
            continue
        flat = g.reshape(-1).double()
        errs = {"norm": abs(flat.norm().item() - float(GC2[f"gnorm_{k}"])) / float(GC2[f"gnorm_{k}"]),
                "max": abs(flat.abs().max().item() - gmax) / gmax,
                "head": float(np.abs(flat[:64].numpy() - GC2[f"ghead_{k}"].astype(np.float64)).max() / gmax)}
        for name, e in errs.items():
            if e > worst[name][0]:
                worst[name] = (e, k)
        checked += 1
    assert checked >= 230, checked
    for name, (e, k) in worst.items():
        assert e < C2_TOL[name], (name, e, k)
    return worst, checked


def test_oracle_training_step_c2_sized_matches_the_reference():
    """CPU: the oracle's training step at 4 pairs x 1024 x 1024 keypoints on the C2 model vs the digests of the reference's (train_c2.npz)."""
    cfg, sd, data, gt0, gt1 = _c2_case()
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    stats = {}
    out = orc.superglue_forward(params, cfg, data, train_stats=stats)
    loss = orc.nll_criterion(out["scores"], gt0, gt1)
    assert abs(loss.item() - float(GC2["loss"])) < 1e-4 * float(GC2["loss"])
    loss.backward()
    _check_c2_gradients((k, p.grad.detach()) for k, p in params.items() if p.requires_grad and p.grad is not None and f"gmax_{k}" in GC2.files)
    for k, v in stats.items():
        assert np.abs(v.detach().numpy() - GC2[f"buf_{k}"]).max() < 1e-5, k


@pytest.mark.gpu
def test_training_step_c2_sized_against_reference(gpu_device):
    """HIP: the training step bench.py and scripts/bench_train_step.py time (C2 model, 4 pairs x 1024 x 1024 keypoints, 20 Sinkhorn iterations)
    against the reference's own step on the same inputs: loss, score statistics, every parameter gradient (digests), running statistics."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1 = _c2_case()
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).train()
    out = model({k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()})
    sc = out["scores"].detach().double()
    assert abs(sc.mean().item() - GC2["scores_mean_std"][0]) < 1e-4 and abs(sc.std().item() - GC2["scores_mean_std"][1]) < 1e-4
    loss = orc.nll_criterion(out["scores"], gt0.to(gpu_device), gt1.to(gpu_device))
    assert abs(loss.item() - float(GC2["loss"])) < 1e-4 * float(GC2["loss"])
    loss.backward()
    worst, checked = _check_c2_gradients((k, p.grad.detach().cpu()) for k, p in model.named_parameters() if p.grad is not None)
    print(f"[train_c2] loss {loss.item():.6f} vs {float(GC2['loss']):.6f}; {checked} parameter gradients, worst relative errors: "
          + ", ".join(f"{n_} {e:.1e} ({k})" for n_, (e, k) in worst.items()))
    for k, b in model.named_buffers():
        if "running" in k:
            assert np.abs(b.cpu().numpy() - GC2[f"buf_{k}"]).max() < 1e-5, k


@pytest.mark.gpu
def test_training_loop_with_adam_reduces_the_loss(gpu_device):
    """examples/train_loop.py on a small model: the drop-in module under torch.optim.Adam (in-place parameter updates between steps, BatchNorm
    running statistics moving) -- the NLL on pairs with known correspondences must fall, every step must stay finite."""
    from examples.train_loop import run
    losses = run(steps=12, pairs=2, kpts=128, dim=64, stages=2, lr=1e-3, log=lambda *_: None)
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses


# ----------------------------------------------------------------------------- metric loss (criterion with margin): gradients through context_descriptors
GM = np.load(os.path.join(GOLDEN, "train_margin.npz"))


def _margin_case():
    from openglue_amd import synthetic as syn
    name, kw = next(iter(MODEL_CASES.items()))
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=len(name))
    B, m, n = (int(v) for v in GM["meta"])
    data = syn.make_batch(B, m, n, cfg["descriptor_dim"], 1, seed=3 + len(name))
    margin, wn, wm = (float(v) for v in GM["weights"])
    return cfg, sd, data, torch.from_numpy(GM["gt0"]), torch.from_numpy(GM["gt1"]), margin, wn, wm


def test_oracle_metric_loss_matches_the_reference():
    """CPU: criterion(..., margin=0.2) of the reference (utils/losses.py:7-93) = oracle nll_criterion + metric_criterion: both loss
    values, and the gradients of L = nll_weight * loss + metric_weight * metric_loss w.r.t. every parameter and the descriptors --
    part of which enters through context_descriptors{0,1} directly (matching_module.py:99-105)."""
    cfg, sd, data, gt0, gt1, margin, wn, wm = _margin_case()
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    data["local_descriptors0"].requires_grad_(True); data["local_descriptors1"].requires_grad_(True)
    out = orc.superglue_forward(params, cfg, data, train_stats={})
    nll = orc.nll_criterion(out["scores"], gt0, gt1)
    met = orc.metric_criterion(out["context_descriptors0"], out["context_descriptors1"], gt0, gt1, margin)
    assert abs(nll.item() - float(GM["loss"])) < 1e-3 and abs(met.item() - float(GM["metric_loss"])) < 1e-4
    assert float(GM["metric_loss"]) > 0.5                        # the hinge terms are active: the fixture does exercise the path
    (wn * nll + wm * met).backward()
    for key, got in (("desc0", data["local_descriptors0"].grad), ("desc1", data["local_descriptors1"].grad)):
        want = GM[f"grad_{key}"]
        assert np.abs(got.numpy() - want).max() < 1e-3 * np.abs(want).max() + 1e-7, key
    checked = 0
    for k, p in params.items():
        if f"grad_{k}" in GM and p.requires_grad:
            want = GM[f"grad_{k}"]
            got = p.grad.numpy() if p.grad is not None else np.zeros_like(want)
            assert np.abs(got - want).max() < 1e-3 * np.abs(want).max() + 1e-6, k
            checked += 1
    assert checked >= 40


@pytest.mark.gpu
def test_training_step_with_metric_loss_against_reference_autograd(gpu_device):
    """HIP: the training step with the reference's metric loss on top of the NLL.  The loss itself is the caller's (torch ops on the
    model's outputs, like utils/losses.py); what is checked is that the gradient arriving at `context_descriptors{0,1}` -- an OUTPUT
    of the HIP forward that the margin=None fixtures only use through `scores` -- flows back through the final projection, the GNN and
    the encoder to every parameter exactly as under the reference's autograd."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1, margin, wn, wm = _margin_case()
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).train()
    dd = {k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}
    dd["local_descriptors0"].requires_grad_(True); dd["local_descriptors1"].requires_grad_(True)
    out = model(dd)
    assert np.abs(out["context_descriptors1"].detach().cpu().numpy() - GM["ctx1"]).max() < 1e-4
    g0, g1 = gt0.to(gpu_device), gt1.to(gpu_device)
    nll = orc.nll_criterion(out["scores"], g0, g1)
    met = orc.metric_criterion(out["context_descriptors0"], out["context_descriptors1"], g0, g1, margin)
    assert abs(met.item() - float(GM["metric_loss"])) < 1e-3 * abs(float(GM["metric_loss"]))
    (wn * nll + wm * met).backward()
    worst, worst_k = 0.0, ""
    for key, got in (("desc0", dd["local_descriptors0"].grad), ("desc1", dd["local_descriptors1"].grad)):
        want = GM[f"grad_{key}"]
        e = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        if e > worst: worst, worst_k = e, key
    for k, p in model.named_parameters():
        want = GM[f"grad_{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-8)
        if np.abs(want).max() > 1e-7 and e > worst: worst, worst_k = e, k
    print(f"[train margin] nll {nll.item():.4f} metric {met.item():.4f}; worst relative gradient error {worst:.2e} ({worst_k})")
    assert worst < 1e-3


# ----------------------------------------------------------------------------- og_gemm_kmajor: the backward products on the operands as they lie
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 100, 52, 64, False), (2, 132, 256, 37, True), (1, 64, 12, 301, True), (5, 260, 68, 1024, True),
                                   (2, 33, 260, 128, False)])
def test_gemm_kmajor_against_float64(gpu_device, shape):
    """C[z] = A[z] B[z] (B stored [K][N]) and C[z] = A[z]^T B[z] (A stored [K][M] too): ragged tile edges (row strides stay multiples of 4: the contract), K not a multiple of 4 in
    the doubly k-major form, batched; vs float64 (exact-fp32 MFMA: 1e-6 relative)."""
    from openglue_amd import train
    Z, M, N, K, a_km = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((Z, K, M) if a_km else (Z, M, K), generator=g).to(gpu_device)
    Bm = torch.randn(Z, K, N, generator=g).to(gpu_device)
    got = (train._bmm_tn(A, Bm) if a_km else train._bmm_nn(A, Bm)).cpu().double()
    want = (A.cpu().double().transpose(1, 2) if a_km else A.cpu().double()) @ Bm.cpu().double()
    assert (got - want).abs().max() < 2e-6 * want.abs().max() * max(1.0, K ** 0.5 / 8)


@pytest.mark.gpu
@pytest.mark.parametrize("T,Cout,Cin", [(4096, 256, 256), (8192, 512, 512), (1000, 64, 4), (301, 128, 36), (77, 32, 8)])
def test_split_k_weight_gradient(gpu_device, T, Cout, Cin):
    """dW = dz^T x contracted over all T tokens as ONE batched split-K launch on the token-major operands (the last chunk is ragged:
    k_total); vs float64."""
    from openglue_amd import train
    g = torch.Generator().manual_seed(T + Cout)
    dz = torch.randn(T, Cout, generator=g).to(gpu_device)
    x = torch.randn(T, Cin, generator=g).to(gpu_device)
    got = train._gemm_splitk(dz, x).cpu().double()
    want = dz.cpu().double().T @ x.cpu().double()
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-6 * want.abs().max() * max(1.0, T ** 0.5 / 8)
    dW, db = train._gemm_splitk(dz, x, with_colsum=True)            # the bias gradient (column sums of dz) out of the same launch
    assert (dW.cpu().double() - want).abs().max() < 2e-6 * want.abs().max() * max(1.0, T ** 0.5 / 8)
    wb = dz.cpu().double().sum(0)
    assert db.shape == (Cout,) and (db.cpu().double() - wb).abs().max() < 2e-6 * max(wb.abs().max(), T ** 0.5)


# ----------------------------------------------------------------------------- flash attention backward (no Nq x Nk tensor)
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 53, 4, 16), (1, 64, 64, 2, 32), (2, 130, 97, 2, 64), (1, 200, 333, 4, 64), (3, 5, 3, 1, 16), (2, 70, 90, 2, 128)])
@pytest.mark.parametrize("flash_bwd", ["1", "0"])
def test_softmax_attention_backward_against_float64(gpu_device, shape, flash_bwd, monkeypatch):
    """train.SoftmaxAttention (forward: the split-f16 flash kernel; backward: og_attention_train_lse + og_attention_backward, P
    recomputed in registers -- or with OG_TRAIN_FLASH_BWD=0 the GEMM-by-GEMM path) vs torch autograd in float64 of
    softmax(q k^T / sqrt(d)) v per head (attention.py:8-19): ragged tile edges in queries and keys, all head sizes (128, round 6: forward on the
    register-staged kernel, backward GEMM by GEMM -- the flash backward covers 16 / 32 / 64)."""
    from openglue_amd import train
    monkeypatch.setenv("OG_TRAIN_FLASH_BWD", flash_bwd)
    B, Nq, Nk, H, d = shape
    D = H * d
    g = torch.Generator().manual_seed(Nq * 3 + Nk)
    q, k, v = (torch.randn(B, n_, D, generator=g) for n_ in (Nq, Nk, Nk))
    R = torch.randn(B, Nq, D, generator=g)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.reshape(B, -1, H, d).transpose(1, 2) for t in (qd, kd, vd))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(B, Nq, D)
    (ref * R.double()).sum().backward()
    qg, kg, vg = (t.to(gpu_device).requires_grad_(True) for t in (q, k, v))
    out = train.SoftmaxAttention.apply(qg, kg, vg, H)
    (out * R.to(gpu_device)).sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 2e-5
    for name, got, want in (("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err < 2e-5, (name, float(err))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 53, 4, 16), (1, 130, 97, 2, 32), (2, 200, 333, 4, 64)])
def test_row_log_sum_exp_from_both_kernels(gpu_device, shape):
    """The row log-sum-exp the flash backward needs: from the forward kernel's online-softmax state (ops.attention(return_lse=True),
    split-f16 scores) and from the exact-fp32 pass og_attention_train_lse, both vs float64."""
    from openglue_amd import ops, _lib
    B, Nq, Nk, H, d = shape
    D = H * d
    g = torch.Generator().manual_seed(Nq + 7 * Nk)
    q, k, v = (torch.randn(B, n_, D, generator=g) * 1.5 for n_ in (Nq, Nk, Nk))
    qh, kh = (t.double().reshape(B, -1, H, d).transpose(1, 2) for t in (q, k))
    want = torch.logsumexp(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)                        # [B, H, Nq]
    qg, kg, vg = (t.to(gpu_device) for t in (q, k, v))
    _, lse_fwd = ops.attention(qg * d ** -0.5, kg, vg, H, return_lse=True)
    lse_bwd = torch.empty(B, H, Nq, device=gpu_device, dtype=torch.float32)
    _lib.check(_lib.load().og_attention_train_lse(qg.data_ptr(), kg.data_ptr(), B, Nq, Nk, H, d, d ** -0.5, lse_bwd.data_ptr(),
                                                  torch.cuda.current_stream(gpu_device).cuda_stream), "og_attention_train_lse")
    assert (lse_fwd.cpu().double() - want).abs().max() < 2e-5
    assert (lse_bwd.cpu().double() - want).abs().max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 37, 4, 16, True), (2, 130, 130, 2, 64, True), (2, 130, 97, 2, 64, False), (1, 200, 333, 4, 32, False)])
def test_projected_attention_on_column_ranges_against_float64(gpu_device, shape):
    """train.ProjectedAttention (round 5: q, k, v stay inside the [tokens, 3D] / [tokens, 2D] projection matrices -- og_split_f16_rows,
    og_attention on strided planes, og_attention_delta, og_attention_backward_ld writing dk, dv into the gradient matrix) vs torch autograd
    in float64 of the projections + softmax attention: self form (one stacked launch) and cross form, ragged tile edges, every gradient."""
    from openglue_amd import train
    B, Nq, Nk, H, d, is_self = shape
    D = H * d
    g = torch.Generator().manual_seed(Nq * 5 + Nk + d)
    xq = torch.randn(B * Nq, D, generator=g)
    xkv = None if is_self else torch.randn(B * Nk, D, generator=g)
    Ws = [torch.randn(D, D, generator=g) * D ** -0.5 for _ in range(3)]
    bs = [torch.randn(D, generator=g) * 0.1 for _ in range(3)]
    R = torch.randn(B * Nq, D, generator=g)
    leaves64 = [t.double().requires_grad_(True) for t in ([xq] + ([] if is_self else [xkv]) + Ws + bs)]
    xq64, rest = leaves64[0], leaves64[1:]
    xkv64 = xq64 if is_self else rest.pop(0)
    W64, b64 = rest[:3], rest[3:]
    q, k, v = (x @ W.T + b for x, W, b in zip((xq64, xkv64, xkv64), W64, b64))
    heads = lambda t, n_: t.reshape(B, n_, H, d).transpose(1, 2)
    ref = (torch.softmax(heads(q, Nq) @ heads(k, Nk).transpose(-1, -2) * d ** -0.5, -1) @ heads(v, Nk)).transpose(1, 2).reshape(B * Nq, D)
    (ref * R.double()).sum().backward()
    leaves = [t.to(gpu_device).requires_grad_(True) for t in ([xq] + ([] if is_self else [xkv]) + Ws + bs)]
    xg, rest = leaves[0], leaves[1:]
    xkvg = None if is_self else rest.pop(0)
    Wg, bg = rest[:3], rest[3:]
    out = train.ProjectedAttention.apply(xg, xkvg, Wg[0], bg[0], Wg[1], bg[1], Wg[2], bg[2], B, Nq, Nk, H)
    (out * R.to(gpu_device)).sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 3e-5
    for i, (got, want) in enumerate(zip(leaves, leaves64)):
        if want.grad.abs().max() < 1e-9:        # the k bias: softmax is invariant to a per-query constant, its gradient is rounding noise
            continue
        err = (got.grad.cpu().double() - want.grad).abs().max() / want.grad.abs().max()
        assert err < 3e-5, (i, float(err))


@pytest.mark.gpu
def test_training_glue_kernels(gpu_device):
    """The single-launch glue of the training step (ABI v9) against plain tensor algebra: og_split_f16_rows (strided, q columns scaled by two
    factors in turn) + og_merge_f16, og_splitk_reduce (dW and the bias column out of padded partial products), og_attention_delta."""
    from openglue_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    g = torch.Generator().manual_seed(11)
    x = torch.randn(300, 96 + 8, generator=g).to(gpu_device)                     # a [300, 96] matrix inside rows of 104 floats
    hi = torch.empty(300, 96, device=gpu_device, dtype=torch.float16); lo = torch.empty_like(hi)
    _lib.check(lib.og_split_f16_rows(x.data_ptr(), 104, 300, 96, 32, 0.125 ** 0.5, 1.4426950408889634, hi.data_ptr(), lo.data_ptr(), 96, st), "split")
    want = x[:, :96].clone()
    want[:, :32] = (want[:, :32] * (0.125 ** 0.5)) * 1.4426950408889634
    wh, wl = ops.split_f16(want.contiguous())
    assert torch.equal(hi, wh) and torch.equal(lo, wl)
    out = torch.empty(300, 96, device=gpu_device)
    _lib.check(lib.og_merge_f16(hi.data_ptr(), lo.data_ptr(), hi.numel(), out.data_ptr(), st), "merge")
    assert torch.equal(out, hi.float() + lo.float())
    for parts in (1, 3, 4, 21):
        part = torch.randn(parts, 40, 28 + 4, generator=g).to(gpu_device)
        dW = torch.empty(40, 28, device=gpu_device); db = torch.empty(40, device=gpu_device)
        _lib.check(lib.og_splitk_reduce(part.data_ptr(), parts, 40, 32, 28, dW.data_ptr(), db.data_ptr(), st), "reduce")
        ref = part.double().sum(0)
        assert (dW.double() - ref[:, :28]).abs().max() < 1e-5 and (db.double() - ref[:, 28]).abs().max() < 1e-5
        dW2 = torch.empty(40, 28, device=gpu_device)
        _lib.check(lib.og_splitk_reduce(part.data_ptr(), parts, 40, 32, 28, dW2.data_ptr(), None, st), "reduce")
        assert torch.equal(dW2, dW)
    a, b = torch.randn(77, 4 * 16, generator=g).to(gpu_device), torch.randn(77, 4 * 16, generator=g).to(gpu_device)
    delta = torch.empty(77 * 4, device=gpu_device)
    _lib.check(lib.og_attention_delta(a.data_ptr(), b.data_ptr(), 77, 4, 16, delta.data_ptr(), st), "delta")
    assert (delta.reshape(77, 4).double() - (a.double() * b.double()).reshape(77, 4, 16).sum(-1)).abs().max() < 1e-5


# ----------------------------------------------------------------------------- the other attentions / encoder in training mode
GV = dict(np.load(os.path.join(GOLDEN, "train_variants.npz")))
VARIANT_CASES = {"linear": dict(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=6, attention="linear"),
                 "favor": dict(descriptor_dim=64, num_stages=1, num_heads=1, num_iters=6, attention="favor_relu"),
                 "siren": dict(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=6, encoder_name="FeedForwardNetSiren", use_offset=True),
                 "square": dict(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=6, use_offset=True)}   # m == n: merged self layers


def _variant_case(name):
    from openglue_amd import synthetic as syn
    cfg = syn.make_config(**VARIANT_CASES[name])
    sd = syn.make_state_dict(cfg, seed=len(name) + 20)
    B, m, n = (int(v) for v in GV[f"{name}_meta"])
    data = syn.make_batch(B, m, n, cfg["descriptor_dim"], 1, seed=5 + len(name))
    return cfg, sd, data, torch.from_numpy(GV[f"{name}_gt0"]), torch.from_numpy(GV[f"{name}_gt1"])


@pytest.mark.parametrize("name", list(VARIANT_CASES))
def test_oracle_training_step_variants_match_the_reference(name):
    """CPU: the oracle in training mode with attention 'linear' / 'favor_relu' and with the Siren encoder vs the reference's training
    step (tests/golden/make_golden_train.py variants): scores, loss, gradients."""
    cfg, sd, data, gt0, gt1 = _variant_case(name)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k and "projection" not in k else v.clone())
              for k, v in sd.items()}
    data["local_descriptors0"].requires_grad_(True); data["local_descriptors1"].requires_grad_(True)
    stats = {}
    out = orc.superglue_forward(params, cfg, data, train_stats=stats)
    # the bar of the path (1e-3): sin(30 x) amplifies fp32 rounding of the first conv (F.linear here, Conv1d there) to 3e-4 on |scores| ~ 80
    assert np.abs(out["scores"].detach().numpy() - GV[f"{name}_scores"]).max() < 1e-3
    loss = orc.nll_criterion(out["scores"], gt0, gt1)
    assert abs(loss.item() - float(GV[f"{name}_loss"])) < 1e-3
    loss.backward()
    for key, got in (("desc0", data["local_descriptors0"].grad), ("desc1", data["local_descriptors1"].grad)):
        want = GV[f"{name}_grad_{key}"]
        assert np.abs(got.numpy() - want).max() < 1e-3 * np.abs(want).max() + 1e-7, key
    checked = 0
    for k, p in params.items():
        if f"{name}_grad_{k}" in GV and p.requires_grad:
            want = GV[f"{name}_grad_{k}"]
            got = p.grad.numpy() if p.grad is not None else np.zeros_like(want)
            assert np.abs(got - want).max() < 1e-3 * np.abs(want).max() + 1e-6, k
            checked += 1
    assert checked >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANT_CASES))
def test_training_step_variants_against_reference_autograd(gpu_device, name):
    """HIP: SuperGlue(config).train() with attention 'linear' (elu + 1 feature map), 'favor_relu' (ReLU random features) -- both through
    train.LinearAttentionCore, O(N), no N x N matrix -- and with the Siren keypoint encoder: scores, loss, gradients of every parameter
    and of the descriptors, running statistics after the step, vs the reference's training step."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1 = _variant_case(name)
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).train()
    dd = {k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}
    dd["local_descriptors0"].requires_grad_(True); dd["local_descriptors1"].requires_grad_(True)
    out = model(dd)
    err_s = np.abs(out["scores"].detach().cpu().numpy() - GV[f"{name}_scores"]).max()
    assert err_s < 1e-3
    assert np.abs(out["context_descriptors0"].detach().cpu().numpy() - GV[f"{name}_ctx0"]).max() < 1e-4
    loss = orc.nll_criterion(out["scores"], gt0.to(gpu_device), gt1.to(gpu_device))
    assert abs(loss.item() - float(GV[f"{name}_loss"])) < 1e-3 * abs(float(GV[f"{name}_loss"]))
    loss.backward()
    worst, worst_k = 0.0, ""
    for key, got in (("desc0", dd["local_descriptors0"].grad), ("desc1", dd["local_descriptors1"].grad)):
        want = GV[f"{name}_grad_{key}"]
        e = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        if e > worst: worst, worst_k = e, key
    n_checked = 0
    for k, p in model.named_parameters():
        want = GV[f"{name}_grad_{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-8)
        if np.abs(want).max() > 1e-7 and e > worst: worst, worst_k = e, k
        n_checked += 1
    print(f"[train_variants {name}] scores err {err_s:.2e}; loss {loss.item():.5f} vs {float(GV[f'{name}_loss']):.5f}; "
          f"{n_checked} parameter gradients, worst relative error {worst:.2e} ({worst_k})")
    assert worst < 1e-3
    for k, b in model.named_buffers():
        if "running" in k:
            assert np.abs(b.cpu().numpy() - GV[f"{name}_buf_{k}"]).max() < 1e-5, k


# ----------------------------------------------------------------------------- eval mode under autograd
GE = np.load(os.path.join(GOLDEN, "eval_grad.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_eval_mode_autograd_against_reference(gpu_device, name):
    """The reference's forward is differentiable in eval() (BatchNorm on running statistics: fine-tuning on frozen statistics).
    With model.eval_autograd = True so is this one, through the same HIP forward / backward Functions as the training path:
    scores, loss, gradients of every parameter and of the descriptors vs the reference (tests/golden/make_golden_train.py eval_grad);
    the running statistics must not move."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1 = _model_case(name)
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).eval()
    model.eval_autograd = True
    before = {k: b.clone() for k, b in model.named_buffers()}
    dd = {k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}
    dd["local_descriptors0"].requires_grad_(True); dd["local_descriptors1"].requires_grad_(True)
    out = model(dd)
    assert np.abs(out["scores"].detach().cpu().numpy() - GE[f"{name}_scores"]).max() < 1e-3
    loss = orc.nll_criterion(out["scores"], gt0.to(gpu_device), gt1.to(gpu_device))
    assert abs(loss.item() - float(GE[f"{name}_loss"])) < 1e-3 * abs(float(GE[f"{name}_loss"]))
    loss.backward()
    worst, worst_k = 0.0, ""
    for key, got in (("desc0", dd["local_descriptors0"].grad), ("desc1", dd["local_descriptors1"].grad)):
        want = GE[f"{name}_grad_{key}"]
        e = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        if e > worst: worst, worst_k = e, key
    for k, p in model.named_parameters():
        want = GE[f"{name}_grad_{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-8)
        if np.abs(want).max() > 1e-7 and e > worst: worst, worst_k = e, k
    print(f"[eval_grad {name}] loss {loss.item():.5f} vs {float(GE[f'{name}_loss']):.5f}; worst relative gradient error {worst:.2e} ({worst_k})")
    assert worst < 1e-3
    for k, b in model.named_buffers():
        assert torch.equal(b, before[k]), k


@pytest.mark.gpu
def test_eval_mode_fast_path_is_never_silently_detached(gpu_device):
    """Default eval(): the fused inference kernels.  With autograd enabled the outputs stay attached to a node that RAISES when a
    gradient is asked of it; under torch.no_grad() they are plain tensors; both give the same numbers."""
    from openglue_amd.superglue import SuperGlue
    cfg, sd, data, gt0, gt1 = _model_case("base")
    model = SuperGlue(cfg)
    model.load_state_dict(sd)
    model = model.to(gpu_device).eval()
    dd = {k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}
    with torch.no_grad():
        plain = model(dd)
    assert not plain["scores"].requires_grad
    out = model(dd)
    assert out["scores"].requires_grad and torch.equal(out["scores"].detach(), plain["scores"])
    with pytest.raises(RuntimeError, match="not differentiable"):
        out["scores"].sum().backward()
    for p in model.parameters():
        p.requires_grad_(False)
    assert not model(dd)["scores"].requires_grad          # nothing asks for a gradient: plain tensors


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_oracle_eval_mode_autograd_matches_the_reference(name):
    """CPU: the oracle in eval mode under autograd vs the reference's eval-mode gradients (eval_grad.npz)."""
    cfg, sd, data, gt0, gt1 = _model_case(name)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    data["local_descriptors0"].requires_grad_(True); data["local_descriptors1"].requires_grad_(True)
    out = orc.superglue_forward(params, cfg, data)
    assert np.abs(out["scores"].detach().numpy() - GE[f"{name}_scores"]).max() < 1e-4
    loss = orc.nll_criterion(out["scores"], gt0, gt1)
    loss.backward()
    for key, got in (("desc0", data["local_descriptors0"].grad), ("desc1", data["local_descriptors1"].grad)):
        want = GE[f"{name}_grad_{key}"]
        assert np.abs(got.numpy() - want).max() < 1e-3 * np.abs(want).max() + 1e-7, key
    checked = 0
    for k, p in params.items():
        if f"{name}_grad_{k}" in GE and p.requires_grad:
            want = GE[f"{name}_grad_{k}"]
            got = p.grad.numpy() if p.grad is not None else np.zeros_like(want)
            assert np.abs(got - want).max() < 1e-3 * np.abs(want).max() + 1e-6, k
            checked += 1
    assert checked >= 40

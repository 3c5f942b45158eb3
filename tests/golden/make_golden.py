#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which does not exist on the GPU
box).  It imports the reference's own `models.superglue.superglue.SuperGlue` and sub-functions
unchanged, feeds them the seeded synthetic inputs / weights of openglue_amd/synthetic.py and
stores inputs + outputs as .npz.  The reference has no tests or known-answer vectors of its
own, so these files are what pins both the CPU oracle (oracle/) and the HIP
path.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Match extraction (models/matching_module.py:174-187) lives in a module that cannot be imported
here (pytorch_lightning etc. are absent), so the golden match indices are produced by a
brute-force per-row Python loop below (independent of the oracle's vectorised restatement) from
the REFERENCE's scores.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from models.superglue.superglue import SuperGlue as RefSuperGlue           # noqa: E402
from models.superglue.optimal_transport import log_otp_solver             # noqa: E402
from models.superglue.attention import softmax_attention                  # noqa: E402
from openglue_amd import synthetic as syn                                 # noqa: E402

MATCH_THRESHOLD = 0.2  # config/config.yaml:40

CASES = {
    # name: (config kwargs, m, n, batch, data seed, how much of `scores` to store)
    "c1": (dict(syn.CONFIGS["C1"]), 64, 64, 1, 1, "full"),
    "mid": (dict(descriptor_dim=128, num_stages=3, num_heads=4, num_iters=20, side_info_size=6), 300, 257, 2, 2, "full"),
    "flags": (dict(descriptor_dim=64, num_stages=2, num_heads=2, num_iters=10, side_info_size=3,
                   residual=False, use_offset=True, reg=0.5, dustbin_score_init=0.3), 96, 130, 2, 3, "full"),
    "nodesc": (dict(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5, side_info_size=1,
                    no_descriptors=True), 70, 33, 1, 4, "full"),
    "siren": (dict(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5, side_info_size=2,
                   encoder_name="FeedForwardNetSiren", hidden_layers_sizes=(32, 64)), 80, 75, 2, 6, "full"),
    "linear": (dict(descriptor_dim=128, num_stages=2, num_heads=4, num_iters=8, side_info_size=1, attention="linear"),
               150, 97, 2, 7, "full"),
    "favor": (dict(descriptor_dim=128, num_stages=2, num_heads=1, num_iters=8, side_info_size=1, attention="favor_relu"),
              140, 101, 2, 8, "full"),
    "c2": (dict(syn.CONFIGS["C2"]), 1024, 1024, 2, 5, "sub8"),
    # round 5 (VERDICT r4 missing 1, 5): the reference's 128-d family at its own operating point (sift_opencv.yaml:2-4: 128-d, config.yaml:53:
    # 20 iterations), whole path, 9 stages, s = 6 ...
    "d128": (dict(descriptor_dim=128, num_stages=9, num_heads=4, num_iters=20, side_info_size=6), 640, 512, 2, 41, "full"),
    # ... and the LARGE BASELINE shapes from the reference itself (until round 4 only the oracle port was checked there):
    "c3": (dict(syn.CONFIGS["C3"]), 2048, 2048, 2, 43, "sub8"),
    "c4": (dict(syn.CONFIGS["C4"]), 4096, 4096, 2, 47, "sub8"),
}


def brute_force_matches(scores: torch.Tensor, thr: float):
    """Per-row loops: first maximal index wins (torch.max CPU semantics), mutual check, exp, threshold."""
    s = scores[:, :-1, :-1].numpy()
    B, m, n = s.shape
    matches0 = np.full((B, m), -1, np.int64)
    ms0 = np.zeros((B, m), np.float32)
    for b in range(B):
        col_best = [int(np.argmax(s[b, :, j])) for j in range(n)]     # np.argmax: first max
        for i in range(m):
            j = int(np.argmax(s[b, i, :]))
            if col_best[j] == i:
                ms0[b, i] = np.exp(np.float32(s[b, i, j]))
                if ms0[b, i] > thr:
                    matches0[b, i] = j
    return matches0, ms0


def full_case(name, kw, m, n, batch, seed, store):
    kw = {k: v for k, v in kw.items() if k not in ("kpts", "batch")}
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=0)
    ref = RefSuperGlue(cfg)
    ref.load_state_dict(sd, strict=True)      # also proves the state-dict names/shapes of synthetic.py
    ref.eval()
    data = syn.make_batch(batch, m, n, cfg["descriptor_dim"], cfg["positional_encoding"]["side_info_size"], seed=seed)
    with torch.no_grad():
        out = ref(data)
        # encoder alone (positional_encoding.py:16-19) on image 0, channel-first [B, D, m]
        k0n = RefSuperGlue.normalize_keypoints(data["keypoints0"], (syn.IMAGE_WH[1], syn.IMAGE_WH[0]))
        enc0 = ref.positional_encoding(k0n, data["side_info0"])
    scores = out["scores"]
    matches0, ms0 = brute_force_matches(scores, MATCH_THRESHOLD)
    arrays = {
        "config_kwargs": np.array(repr(kw)),
        "m": m, "n": n, "batch": batch, "seed": seed,
        "matches0": matches0, "matching_scores0": ms0,
        "row_sums64": scores.double().sum(2).numpy(), "col_sums64": scores.double().sum(1).numpy(),
        "encoder0": enc0.numpy(),
    }
    if store == "full":
        arrays["scores"] = scores.numpy()
        arrays["context_descriptors0"] = out["context_descriptors0"].numpy()
        arrays["context_descriptors1"] = out["context_descriptors1"].numpy()
    else:  # strided subsample + the dustbin row/col in full
        arrays["scores_sub8"] = scores[:, ::8, ::8].numpy()
        arrays["scores_lastrow"] = scores[:, -1, :].numpy()
        arrays["scores_lastcol"] = scores[:, :, -1].numpy()
        arrays["context_descriptors0_sub"] = out["context_descriptors0"][:, ::4, ::16].numpy()
        arrays["context_descriptors1_sub"] = out["context_descriptors1"][:, ::4, ::16].numpy()
        if name != "c2":      # (c2 predates these) top-1 / top-2 gap of every row and column of the reference's scores[:, :-1, :-1]: lets a test
            inner = scores[:, :-1, :-1]                                   # explain an index difference as a near-tie without re-running anything
            t2r = inner.topk(2, dim=2).values; t2c = inner.topk(2, dim=1).values
            arrays["row_gap"] = (t2r[..., 0] - t2r[..., 1]).numpy()
            arrays["col_gap"] = (t2c[:, 0] - t2c[:, 1]).numpy()
            arrays["row_argmax"] = inner.argmax(2).numpy().astype(np.int32)
    if name == "c1":  # inputs and weights in full, so this case does not depend on torch's RNG stream
        for k, v in data.items():
            if torch.is_tensor(v):
                arrays["in_" + k] = v.numpy()
        for k, v in sd.items():
            arrays["sd_" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
    nvalid = int((matches0 >= 0).sum())
    print(f"{name}: scores {tuple(scores.shape)} absmax {scores.abs().max():.3f} valid matches {nvalid}/{matches0.size}")


def trained_case():
    """`trained`: the reference on a TRAINED-LIKE checkpoint (syn.make_trained_like_state_dict: dead BatchNorm channels, large
    gamma / sigma, a few large weights) and unit-norm descriptors (desc_scale = 1: the dustbin-dominated regime of real SuperPoint /
    SIFT inputs) -- the statistics that make the eval-mode BatchNorm fold produce weights beyond 256 |w| < 65504."""
    kw = dict(descriptor_dim=256, num_stages=2, num_heads=4, num_iters=20, side_info_size=1)
    m, n, batch, seed = 140, 120, 2, 31
    cfg = syn.make_config(**kw)
    sd = syn.make_trained_like_state_dict(cfg, seed=0)
    ref = RefSuperGlue(cfg)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    arrays = {"config_kwargs": np.array(repr(kw)), "m": m, "n": n, "batch": batch, "seed": seed}
    for tag, scale in (("unit", 1.0), ("x4", 4.0)):
        data = syn.make_batch(batch, m, n, 256, 1, seed=seed, desc_scale=scale)
        with torch.no_grad():
            out = ref(data)
        matches0, ms0 = brute_force_matches(out["scores"], MATCH_THRESHOLD)
        arrays.update({f"{tag}_scores": out["scores"].numpy(), f"{tag}_matches0": matches0, f"{tag}_matching_scores0": ms0,
                       f"{tag}_context_descriptors0": out["context_descriptors0"].numpy()})
        print(f"trained/{tag}: scores absmax {out['scores'].abs().max():.3f} valid matches {int((matches0 >= 0).sum())}/{matches0.size}")
    np.savez_compressed(os.path.join(HERE, "trained.npz"), **arrays)


def stage_cases():
    g = torch.Generator().manual_seed(1234)
    # log_otp_solver alone (optimal_transport.py:4-28), non-square, reg != 1
    B, m, n = 3, 37, 53
    Mx = 4.0 * torch.randn(B, m + 1, n + 1, generator=g)
    norm = -math.log(m + n)
    la = torch.full((B, m + 1), norm); la[:, -1] += math.log(n)
    lb = torch.full((B, n + 1), norm); lb[:, -1] += math.log(m)
    outs = {}
    for iters, reg in ((1, 1.0), (7, 1.0), (25, 0.7)):
        outs[f"sinkhorn_i{iters}_r{reg}"] = log_otp_solver(la, lb, Mx, num_iters=iters, reg=reg).numpy()
    np.savez_compressed(os.path.join(HERE, "stage_sinkhorn.npz"), M=Mx.numpy(), log_a=la.numpy(), log_b=lb.numpy(), **outs)
    # softmax_attention alone (attention.py:8-19), [B, H, d, N] layout, N_q != N_kv
    q = 2.0 * torch.randn(2, 4, 16, 50, generator=g)
    k = 2.0 * torch.randn(2, 4, 16, 70, generator=g)
    v = torch.randn(2, 4, 16, 70, generator=g)
    o, att = softmax_attention(q, k, v)
    np.savez_compressed(os.path.join(HERE, "stage_attention.npz"), q=q.numpy(), k=k.numpy(), v=v.numpy(),
                        out=o.numpy(), att_rowsum=att.sum(-1).numpy())
    print("stage fixtures written")


LAYER_CASES = {
    # name: (config kwargs, m, n, batch, data seed): the residual stream at every stage boundary of the GNN
    "mid": (dict(descriptor_dim=128, num_stages=3, num_heads=4, num_iters=20, side_info_size=6), 300, 257, 2, 2),
    "flags": (dict(descriptor_dim=64, num_stages=2, num_heads=2, num_iters=10, side_info_size=3,
                   residual=False, use_offset=True, reg=0.5, dustbin_score_init=0.3), 96, 130, 2, 3),
    "d256": (dict(descriptor_dim=256, num_stages=1, num_heads=4, num_iters=5, side_info_size=1), 130, 100, 2, 21),   # the fused message-MLP kernel
}
# round 5: the 128-d kernel family (mlp_fused_kernel<128> and friends) at 9 stages, s = 6, >= 512 keypoints; its own file (stage_layers_d128.npz)
LAYER_CASES_D128 = {
    "d128": (dict(descriptor_dim=128, num_stages=9, num_heads=4, num_iters=20, side_info_size=6), 520, 512, 1, 41),
}


def layer_cases(cases=None, fname="stage_layers.npz"):
    """Per-stage goldens: x = local_descriptors + positional_encoding(...) as it enters the GNN (superglue.py:41-55) and
    the descriptors after every element of attention_gnn.layers (attention_gnn.py:57-77: DescriptorsSelfAttention /
    DescriptorsCrossAttention = ResidualAttentionMessagePropagation on both images), from the reference's own modules.  Stored
    token-major [B, n, D] (the reference holds them channel-first)."""
    arrays = {}
    for name, (kw, m, n, batch, seed) in (cases or LAYER_CASES).items():
        cfg = syn.make_config(**kw)
        sd = syn.make_state_dict(cfg, seed=0)
        ref = RefSuperGlue(cfg)
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        data = syn.make_batch(batch, m, n, cfg["descriptor_dim"], cfg["positional_encoding"]["side_info_size"], seed=seed)
        with torch.no_grad():
            hw = (syn.IMAGE_WH[1], syn.IMAGE_WH[0])
            pe0 = ref.positional_encoding(RefSuperGlue.normalize_keypoints(data["keypoints0"], hw), data["side_info0"])
            pe1 = ref.positional_encoding(RefSuperGlue.normalize_keypoints(data["keypoints1"], hw), data["side_info1"])
            d0 = data["local_descriptors0"].transpose(2, 1) + pe0
            d1 = data["local_descriptors1"].transpose(2, 1) + pe1
            arrays[f"{name}/x0_tap0"] = d0.transpose(1, 2).contiguous().numpy()
            arrays[f"{name}/x1_tap0"] = d1.transpose(1, 2).contiguous().numpy()
            nl = len(ref.attention_gnn.layers)
            for i, layer in enumerate(ref.attention_gnn.layers):
                d0, d1 = layer(d0, d1)
                if i < 2 or i == nl - 1:          # the first self and cross layer and the last layer (file size)
                    arrays[f"{name}/x0_tap{i + 1}"] = d0.transpose(1, 2).contiguous().numpy()
                    arrays[f"{name}/x1_tap{i + 1}"] = d1.transpose(1, 2).contiguous().numpy()
            # and the module alone with DIFFERENT query / key-value sets (the cross form), attention_gnn.py:43-55
            arrays[f"{name}/ramp_q0_kv1"] = ref.attention_gnn.layers[1].module(
                data["local_descriptors0"].transpose(2, 1), data["local_descriptors1"].transpose(2, 1)).transpose(1, 2).contiguous().numpy()
        arrays[f"{name}/meta"] = np.array(repr(dict(kw=kw, m=m, n=n, batch=batch, seed=seed, taps=len(ref.attention_gnn.layers) + 1)))
        print(f"layers {name}: {len(ref.attention_gnn.layers)} layers, |x| max {float(d0.abs().max()):.2f}")
    np.savez_compressed(os.path.join(HERE, fname), **arrays)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["layers"]:
        layer_cases()
        sys.exit(0)
    if sys.argv[1:] == ["layers_d128"]:
        layer_cases(LAYER_CASES_D128, "stage_layers_d128.npz")
        sys.exit(0)
    if sys.argv[1:] == ["trained"]:
        trained_case()
        sys.exit(0)
    only = sys.argv[1:]                      # e.g. `make_golden.py favor`: (re)generate the named cases only
    for name, (kw, m, n, batch, seed, store) in CASES.items():
        if not only or name in only:
            full_case(name, kw, m, n, batch, seed, store)
    if not only:
        stage_cases()
    print("torch", torch.__version__)

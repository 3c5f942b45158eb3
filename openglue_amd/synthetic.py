"""Seeded synthetic keypoints / descriptors / weights for the SuperGlue hot path.

Used by bench.py, the parity tests and the golden-fixture generator, so that the
same (seed -> tensors) mapping is available on the build box and on the GPU box
without shipping 48 MB of weights.  Everything is generated on the CPU with an
explicit ``torch.Generator`` and is therefore reproducible for a given torch
version (the fixtures under tests/golden/ were produced with the same functions).

Recipe follows SURVEY.md §8(d): keypoints uniform on a 960x720 image, 60 % of the
image-1 keypoints are noisy copies (2 px) of image-0 keypoints with a perturbed copy
of the descriptor, descriptors are unit-norm Gaussians scaled by 32 (the scale stands
in for trained weights: with random-init weights unit-norm descriptors all fall into
the dustbin), side-info U(0,1).  BatchNorm statistics are randomised so that the
Conv -> ReLU -> BN order of the reference (models/utils.py:52-56) is exercised.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch

IMAGE_WH = (960, 720)  # config/config.yaml:10 target_size

# BASELINE.json "configs", as keyword sets understood by make_config().
CONFIGS = {
    "C1": dict(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=3, side_info_size=1,
               kpts=(64, 64), batch=1),
    "C2": dict(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=100, side_info_size=1,
               kpts=(1024, 1024), batch=32),
    "C3": dict(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=100, side_info_size=1,
               kpts=(2048, 2048), batch=32),   # 256 pairs over 8 GPUs -> 32 per GPU
    "C4": dict(descriptor_dim=128, num_stages=9, num_heads=4, num_iters=100, side_info_size=6,
               kpts=(4096, 4096), batch=8),    # 64 pairs over 8 GPUs -> 8 per GPU
    # not a BASELINE config: the reference's OWN shipped operating point for its 128-d feature family -- config/features/sift_opencv.yaml:2-4
    # (descriptor_dim 128, max_keypoints 2048), config/config.yaml:44,53 (laf_to_sideinfo_method 'none' = response only, num_iters 20)
    "S128": dict(descriptor_dim=128, num_stages=9, num_heads=4, num_iters=20, side_info_size=1,
                 kpts=(2048, 2048), batch=32),
    # ... and for its 256-d family: config/features/superpoint_magicleap.yaml:2-4 (descriptor_dim 256, max_keypoints 2048), num_iters 20
    "S256": dict(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=20, side_info_size=1,
                 kpts=(2048, 2048), batch=32),
    # BASELINE configs[3] leaves the Sinkhorn iteration count open: C4 runs 100 like configs[1-2]; this is the sibling at the reference's default 20
    "C4i20": dict(descriptor_dim=128, num_stages=9, num_heads=4, num_iters=20, side_info_size=6,
                  kpts=(4096, 4096), batch=8),
}


def make_config(descriptor_dim: int = 256, num_stages: int = 9, num_heads: int = 4,
                num_iters: int = 100, side_info_size: int = 1, reg: float = 1.0,
                residual: bool = True, use_offset: bool = False, no_descriptors: bool = False,
                dustbin_score_init: float = 1.0,
                hidden_layers_sizes: Sequence[int] = (32, 64, 128), encoder_name: str = "FeedForwardNet",
                attention: str = "softmax", **_unused) -> dict:
    """The `superglue:` config block with the keys SuperGlue.__init__ reads
    (reference models/superglue/superglue.py:16-27, config/config.yaml:42-55) plus the keys
    MatchingTrainingModule injects (models/matching_module.py:35-43)."""
    return {
        "descriptor_dim": descriptor_dim,
        "positional_encoding": {
            "output_size": descriptor_dim,
            "side_info_size": side_info_size,
            "encoder_name": encoder_name,
            "hidden_layers_sizes": list(hidden_layers_sizes),
        },
        "attention_gnn": {
            "num_stages": num_stages,
            "embed_dim": descriptor_dim,
            "num_heads": num_heads,
            "attention": attention,
            "use_offset": use_offset,
        },
        "dustbin_score_init": dustbin_score_init,
        "otp": {"num_iters": num_iters, "reg": reg},
        "residual": residual,
        "no_descriptors": no_descriptors,
    }


def state_dict_spec(config: dict) -> "OrderedDict[str, tuple]":
    """Names, shapes and kinds of every state-dict entry of the reference module
    (SURVEY.md §3.6), in the reference's registration order."""
    D = config["descriptor_dim"]
    pe = config["positional_encoding"]
    sizes = [2 + pe["side_info_size"], *pe["hidden_layers_sizes"], pe["output_size"]]
    L = config["attention_gnn"]["num_stages"]
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(prefix, cout, cin):
        spec[prefix + ".weight"] = ((cout, cin, 1), "conv_w", cin)
        spec[prefix + ".bias"] = ((cout,), "conv_b", cin)

    def bn(prefix, c):
        spec[prefix + ".weight"] = ((c,), "bn_w", 0)
        spec[prefix + ".bias"] = ((c,), "bn_b", 0)
        spec[prefix + ".running_mean"] = ((c,), "bn_mean", 0)
        spec[prefix + ".running_var"] = ((c,), "bn_var", 0)
        spec[prefix + ".num_batches_tracked"] = ((), "bn_count", 0)

    if config.get("residual", False):
        spec["mix_coefs"] = ((D, 1), "mix", 0)
    spec["dustbin_score"] = ((), "dustbin", 0)
    if pe.get("encoder_name", "FeedForwardNet") == "FeedForwardNetSiren":
        # FeedForwardNetSiren = [Conv1d, Sine] * (n-1) + Conv1d (models/utils.py:32-45) -> conv indices 0, 2, 4, ...
        for i in range(1, len(sizes)):
            conv(f"positional_encoding.encoder.{2 * (i - 1)}", sizes[i], sizes[i - 1])
    else:
        # FeedForwardNet = [Conv1d, ReLU, BatchNorm1d] * (n-1) + Conv1d  -> Sequential indices 0,(1),2, 3,(4),5, ...
        idx = 0
        for i in range(1, len(sizes) - 1):
            conv(f"positional_encoding.encoder.{idx}", sizes[i], sizes[i - 1])
            bn(f"positional_encoding.encoder.{idx + 2}", sizes[i])
            idx += 3
        conv(f"positional_encoding.encoder.{idx}", sizes[-1], sizes[-2])
    favor = config["attention_gnn"].get("attention", "softmax") == "favor_relu"
    for l in range(2 * L):
        p = f"attention_gnn.layers.{l}.module"
        if favor:   # GeneralizedFavorAttention registers its random features as a buffer (attention.py:43-53; __init__.py:19-25:
            #         num_orthogonal_features = 2 * embed_dim), created before the projections (attention_gnn.py:14)
            spec[f"{p}.mha.attention_func.projection_matrix"] = ((2 * D, D), "favor_proj", D)
        for name in ("in_proj_q", "in_proj_k", "in_proj_v", "out_proj"):
            conv(f"{p}.mha.{name}", D, D)
        conv(f"{p}.fc.0", 2 * D, 2 * D)
        bn(f"{p}.fc.2", 2 * D)
        conv(f"{p}.fc.3", D, 2 * D)
    conv("linear_proj", D, D)
    return spec


def orthogonal_random_features(num_rows: int, num_cols: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """FAVOR+ random features as the reference draws them (FavorAttention.sample_orthogonal_random_vectors, attention.py:62-78):
    ceil(rows / cols) Gaussian blocks [cols, cols], each orthogonalised by QR, the unit rows rescaled by the norms of the
    Gaussian rows."""
    blocks = math.ceil(num_rows / num_cols)
    unstructured = torch.randn(blocks, num_cols, num_cols, generator=generator)
    norm = unstructured.norm(dim=-1).view(-1, 1)
    q, _ = torch.linalg.qr(unstructured, mode="reduced")
    q = q.transpose(-1, -2).reshape(-1, num_cols)
    return q[:num_rows] * norm[:num_rows]


def make_state_dict(config: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic weights keyed by parameter NAME (not by module construction order), so the
    reference module, the oracle and the HIP module all see the same numbers.
    Conv weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default Conv1d init);
    BatchNorm affine/statistics randomised."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind, fan_in) in state_dict_spec(config).items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        if kind in ("conv_w", "conv_b"):
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind in ("bn_w", "bn_var"):
            t = 0.75 + 0.5 * torch.rand(shape, generator=g)
        elif kind in ("bn_b", "bn_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_count":
            t = torch.tensor(1, dtype=torch.long)
        elif kind == "mix":
            t = 0.5 * torch.randn(shape, generator=g)
        elif kind == "dustbin":
            t = torch.tensor(float(config["dustbin_score_init"]))
        elif kind == "favor_proj":
            t = orthogonal_random_features(shape[0], shape[1], g)
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[name] = t
    return out


def make_trained_like_state_dict(config: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """make_state_dict with the statistics a TRAINED checkpoint shows and random initialisation never does:
    in every message MLP a few BatchNorm channels are DEAD (post-ReLU output identically 0: running_mean = 0, running_var ~ 1e-12, so
    the eval-mode fold multiplies the next conv's column by ~316 * gamma), a few have a large gamma / sigma ratio, the matching conv
    columns carry ordinary weights, and a few projection weights are large.  Deterministic in (config, seed)."""
    sd = make_state_dict(config, seed)
    D = config["descriptor_dim"]
    L = 2 * config["attention_gnn"]["num_stages"]
    g = torch.Generator().manual_seed(977 + seed)
    for l in range(L):
        pre = f"attention_gnn.layers.{l}.module"
        dead = torch.randperm(2 * D, generator=g)[:3]
        sd[f"{pre}.fc.2.running_var"][dead] = 1e-12
        sd[f"{pre}.fc.2.running_mean"][dead] = 0.0
        sd[f"{pre}.fc.2.weight"][dead] = 8.0 + 4.0 * torch.rand(3, generator=g)
        sd[f"{pre}.fc.3.weight"][:, dead, 0] = 0.3 * torch.randn(D, 3, generator=g)          # 0.3 * 10 / sqrt(1e-5) * 256 >> 65504
        # a dead channel is dead because its conv row is strongly negative for every input: make it so (bias far below any response)
        sd[f"{pre}.fc.0.bias"][dead] = -1e3
        big = torch.randperm(2 * D, generator=g)[:4]
        sd[f"{pre}.fc.2.running_var"][big] = 1e-4            # gamma / sigma ~ 100
        if l % 3 == 0:
            sd[f"{pre}.mha.in_proj_k.weight"][l % D, (7 * l + 3) % D, 0] = 150.0                # 256 * 150 > 32768
    return sd


def make_pair(m: int, n: int, descriptor_dim: int, side_info_size: int, seed: int,
              desc_scale: float = 32.0, inlier_frac: float = 0.6) -> Dict[str, torch.Tensor]:
    """One synthetic image pair (no batch dimension)."""
    g = torch.Generator().manual_seed(1_000_003 * (seed + 1))
    W, H = IMAGE_WH
    wh = torch.tensor([W - 1.0, H - 1.0])
    k0 = torch.rand(m, 2, generator=g) * wh
    d0 = torch.nn.functional.normalize(torch.randn(m, descriptor_dim, generator=g), dim=-1)
    k1 = torch.rand(n, 2, generator=g) * wh
    d1 = torch.nn.functional.normalize(torch.randn(n, descriptor_dim, generator=g), dim=-1)
    n_in = min(int(inlier_frac * min(m, n)), m, n)
    src = torch.randperm(m, generator=g)[:n_in]
    dst = torch.randperm(n, generator=g)[:n_in]
    k1[dst] = (k0[src] + 2.0 * torch.randn(n_in, 2, generator=g)).clamp_(min=0.0)
    k1[dst] = torch.minimum(k1[dst], wh)
    d1[dst] = torch.nn.functional.normalize(
        d0[src] + 0.02 * torch.randn(n_in, descriptor_dim, generator=g), dim=-1)
    s0 = torch.rand(m, side_info_size, generator=g)
    s1 = torch.rand(n, side_info_size, generator=g)
    return {
        "keypoints0": k0, "keypoints1": k1,
        "local_descriptors0": d0 * desc_scale, "local_descriptors1": d1 * desc_scale,
        "side_info0": s0, "side_info1": s1,
    }


def make_batch(batch: int, m: int, n: int, descriptor_dim: int, side_info_size: int = 1,
               seed: int = 0, first_pair: int = 0, device: Optional[torch.device] = None,
               **pair_kw) -> Dict[str, object]:
    """A batch of `batch` pairs; pair p of the job uses seed (seed, first_pair + p) so that a
    sharded job sees the same pairs whatever the world size."""
    pairs = [make_pair(m, n, descriptor_dim, side_info_size, seed * 100_003 + first_pair + p, **pair_kw)
             for p in range(batch)]
    data: Dict[str, object] = {k: torch.stack([p[k] for p in pairs]) for k in pairs[0]}
    if device is not None:
        data = {k: v.to(device) for k, v in data.items()}
    data["image0_size"] = list(IMAGE_WH)  # [W, H]  (reference superglue.py:38 reverses it)
    data["image1_size"] = list(IMAGE_WH)
    return data


def ragged_lengths(batch: int, lo: int, hi: int, seed: int = 0) -> List[tuple]:
    """BASELINE config 5: per-image keypoint counts ~ U{lo..hi}."""
    g = torch.Generator().manual_seed(77_777 + seed)
    t = torch.randint(lo, hi + 1, (batch, 2), generator=g)
    return [(int(a), int(b)) for a, b in t.tolist()]

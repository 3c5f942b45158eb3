"""Ragged batches (BASELINE config 5: 512-2048 keypoints per image, different for every pair).

The reference has no mask argument anywhere on the path: its loaders make batches
rectangular by truncation or by zero "virtual keypoints" that DO take part in attention and Sinkhorn, so
the only unambiguous semantics for a truly ragged batch is "every pair on its own, with its own (m, n)".
Two implementations with identical results:
  * `SuperGlue.match_ragged` (openglue_amd/superglue.py -> og_forward_ragged): token-PACKED tensors and a
    per-pair length descriptor handed to every kernel -- one launch sequence for the whole batch.  This is
    the fast path.
  * `match_ragged` below: pairs bucketed by EXACT shape, every bucket through the uniform batched path,
    results scattered back in job order.  Kept as the independent cross-check for the packed kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Mapping, Sequence

import torch

_PAIR_KEYS = ("keypoints0", "keypoints1", "local_descriptors0", "local_descriptors1", "side_info0", "side_info1")


def bucket_by_shape(pairs: Sequence[Mapping]) -> "OrderedDict[tuple, List[int]]":
    """pair indices grouped by (m, n, image sizes), in first-seen order."""
    buckets: "OrderedDict[tuple, List[int]]" = OrderedDict()
    for i, p in enumerate(pairs):
        key = (p["keypoints0"].shape[-2], p["keypoints1"].shape[-2],
               tuple(p.get("image0_size", ())), tuple(p.get("image1_size", ())))
        buckets.setdefault(key, []).append(i)
    return buckets


def match_ragged(model, pairs: Sequence[Mapping], match_threshold: float = 0.2, both_sides: bool = True) -> List[Dict[str, torch.Tensor]]:
    """`pairs[i]` is an un-batched data dict (keypoints0 [m_i, 2], ... , image0_size=[W, H]) on the GPU.
    Returns one result dict per pair (scores [m_i+1, n_i+1], matches0 [m_i], ...), in order."""
    out: List[Dict[str, torch.Tensor]] = [None] * len(pairs)  # type: ignore[list-item]
    for _key, idx in bucket_by_shape(pairs).items():
        batch = {k: torch.stack([pairs[i][k] for i in idx]) for k in _PAIR_KEYS}
        for k in ("image0_size", "image1_size", "image0", "image1"):
            if k in pairs[idx[0]]:
                batch[k] = pairs[idx[0]][k] if k.endswith("_size") else torch.stack([pairs[i][k] for i in idx])
        res = model.match(batch, match_threshold, both_sides=both_sides)
        for j, i in enumerate(idx):
            out[i] = {k: v[j] for k, v in res.items()}
    return out

"""The steps either side of the matcher in OpenGlue's inference loop, on the GPU:

  prepare_features_output   models/features/utils.py:54-65 + models/laf_converter.py  (before SuperGlue.forward)
  compact_matches           inference.py:192-209                                        (after the match extraction)

Thin wrappers over og_prepare_features / og_compact_matches (include/openglue_amd.h); GPU tensors only.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib

# laf_to_sideinfo_method (config/config.yaml:43, models/laf_converter.py:106-128) -> (code, side-info width incl. the response)
LAF_METHODS = {"none": (0, 1), "scale": (1, 2), "rotation": (2, 3), "scale_rotation": (3, 4), "affine": (4, 6)}


def side_info_size(method: str) -> int:
    """positional_encoding.side_info_size MatchingTrainingModule injects (matching_module.py:42-43): laf dims + 1."""
    return _method(method)[1]


def _method(method: str):
    try:
        return LAF_METHODS[method.lower()]
    except KeyError:
        raise NameError("Unexpected name for the method: {}".format(method))      # laf_converter.py:128


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU; openglue_amd has no CPU path")
    return t.detach().to(dtype).contiguous()


def prepare_features_output(lafs: torch.Tensor, responses: torch.Tensor, desc: torch.Tensor, method: str = "none",
                            permute_desc: bool = False, log_response: bool = False) -> Dict[str, torch.Tensor]:
    """lafs [B,N,2,3], responses [B,N], desc [B,N,D] -> {'keypoints' [B,N,2], 'side_info' [B,N,s], 'local_descriptors'}."""
    lib = _lib.load()
    code, s = _method(method)
    lafs, responses = _req(lafs, "lafs"), _req(responses, "responses")
    B, N = responses.shape
    if lafs.shape != (B, N, 2, 3):
        raise ValueError("lafs must be [B, N, 2, 3]")
    kpts = torch.empty(B, N, 2, device=lafs.device, dtype=torch.float32)
    side = torch.empty(B, N, s, device=lafs.device, dtype=torch.float32)
    rc = lib.og_prepare_features(lafs.data_ptr(), responses.data_ptr(), B * N, code, int(log_response), kpts.data_ptr(),
                                 side.data_ptr(), torch.cuda.current_stream(lafs.device).cuda_stream)
    _lib.check(rc, "og_prepare_features")
    return {"keypoints": kpts, "side_info": side, "local_descriptors": desc.permute(0, 2, 1) if permute_desc else desc}


def compact_matches(matches0: torch.Tensor, matching_scores0: torch.Tensor, lafs0: Optional[torch.Tensor] = None,
                    lafs1: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Valid matches of a batch in (pair, keypoint) order, as OpenGlueMatcher.forward returns them
    (inference.py:192-209).  Synchronises once to read the match count (torch's boolean indexing does too)."""
    lib = _lib.load()
    matches0 = _req(matches0, "matches0", torch.int64)
    ms0 = _req(matching_scores0, "matching_scores0")
    B, M = matches0.shape
    dev = matches0.device
    if (lafs0 is None) != (lafs1 is None):
        raise ValueError("lafs0 and lafs1 go together")
    if lafs0 is not None:
        lafs0, lafs1 = _req(lafs0, "lafs0"), _req(lafs1, "lafs1")
    N = lafs1.shape[1] if lafs1 is not None else 1
    K = B * M
    idxs = torch.empty(K, 2, device=dev, dtype=torch.int64)
    bidx = torch.empty(K, device=dev, dtype=torch.int64)
    conf = torch.empty(K, device=dev, dtype=torch.float32)
    ml0 = torch.empty(K, 2, 3, device=dev) if lafs0 is not None else None
    ml1 = torch.empty(K, 2, 3, device=dev) if lafs0 is not None else None
    k0 = torch.empty(K, 2, device=dev) if lafs0 is not None else None
    k1 = torch.empty(K, 2, device=dev) if lafs0 is not None else None
    count = torch.zeros(1, device=dev, dtype=torch.int32)
    ws = torch.empty(lib.og_compact_workspace_bytes(B, M), device=dev, dtype=torch.uint8)
    p = lambda t: None if t is None else t.data_ptr()
    rc = lib.og_compact_matches(matches0.data_ptr(), ms0.data_ptr(), p(lafs0), p(lafs1), B, M, N, idxs.data_ptr(), bidx.data_ptr(),
                                conf.data_ptr(), p(ml0), p(ml1), p(k0), p(k1), count.data_ptr(), ws.data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "og_compact_matches")
    k = int(count.item())
    out = {"original_matching_idxs": idxs[:k], "batch_indexes": bidx[:k], "confidence": conf[:k]}
    if lafs0 is not None:   # [None]: the reference returns the matched LAFs with a leading batch dimension of 1
        out.update(lafs0=ml0[:k][None], lafs1=ml1[:k][None], keypoints0=k0[:k], keypoints1=k1[:k])
    return out

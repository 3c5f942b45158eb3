"""Drop-in for `OpenGlueMatcher` of the reference's inference script (inference.py:83-209): local features of two images ->
matched keypoints.  Same constructor and `forward(data)` contract; every step between the feature extractor and the returned
dictionary runs on HIP kernels:

    prepare_features_output (models/features/utils.py:54-65 + LAF converter)   features.prepare_features_output  (og_prepare_features)
    SuperGlue.forward + mutual-NN extraction (inference.py:173-190)            SuperGlue.match                   (og_forward)
    compaction of the valid matches, matched LAFs / centres (inference.py:192-209)   features.compact_matches    (og_compact_matches)

The feature extractor itself (`local_feature`: SuperPoint / SIFT / ... -- models/features/*) is the caller's and out of scope
(SURVEY.md §8); it is only invoked when the data do not already carry `lafs{0,1}` / `descriptors{0,1}` / `responses{0,1}`, exactly
like the reference (inference.py:141-153).  GPU tensors only: there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch
import torch.nn as nn

from openglue_amd import features


class OpenGlueMatcher(nn.Module):
    def __init__(self, local_feature: Optional[nn.Module], matcher: nn.Module, match_config: Mapping = {}) -> None:
        super().__init__()
        self.local_feature = local_feature
        self.laf_method = str(match_config["superglue"]["laf_to_sideinfo_method"])      # inference.py:103
        features.side_info_size(self.laf_method)                                         # NameError for an unknown method (laf_converter.py:128)
        self.matcher = matcher
        self.match_config = match_config
        self.eval()

    def extract_features(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """inference.py:108-113."""
        if self.local_feature is None:
            raise RuntimeError("OpenGlueMatcher: no local feature extractor given and the data carry no pre-extracted features")
        lafs, resps, descs = self.local_feature(image)
        return {"lafs": lafs, "responses": resps, "descriptors": descs}

    def no_match_output(self, device: torch.device, dtype: torch.dtype) -> dict:
        """inference.py:115-123."""
        return {"keypoints0": torch.empty(0, 2, device=device, dtype=dtype), "keypoints1": torch.empty(0, 2, device=device, dtype=dtype),
                "lafs0": torch.empty(0, 0, 2, 3, device=device, dtype=dtype), "lafs1": torch.empty(0, 0, 2, 3, device=device, dtype=dtype),
                "confidence": torch.empty(0, device=device, dtype=dtype), "batch_indexes": torch.empty(0, device=device, dtype=torch.long)}

    @torch.no_grad()
    def forward(self, data: Mapping) -> Dict[str, torch.Tensor]:
        feats = []
        for i in (0, 1):                                                                 # inference.py:141-153
            if f"lafs{i}" not in data or f"descriptors{i}" not in data:
                f = self.extract_features(data[f"image{i}"])
                feats.append((f["lafs"], f["descriptors"], f["responses"]))
            else:
                feats.append((data[f"lafs{i}"], data[f"descriptors{i}"], data[f"responses{i}"]))
        (lafs0, descs0, resps0), (lafs1, descs1, resps1) = feats
        if "image0" in data and "image1" in data:                                       # inference.py:156-159: [W, H] from the images
            size0 = [data["image0"].shape[-1], data["image0"].shape[-2]]
            size1 = [data["image1"].shape[-1], data["image1"].shape[-2]]
        else:                                                                            # pre-extracted features without the pixels
            size0, size1 = list(data["image0_size"][:2]), list(data["image1_size"][:2])
        sg_cfg = self.match_config["superglue"]
        log_resp = bool(sg_cfg.get("log_transform_response", False))
        f0 = features.prepare_features_output(lafs0, resps0, descs0, self.laf_method, log_response=log_resp)
        f1 = features.prepare_features_output(lafs1, resps1, descs1, self.laf_method, log_response=log_resp)
        batch = {"keypoints0": f0["keypoints"], "keypoints1": f1["keypoints"],
                 "local_descriptors0": f0["local_descriptors"], "local_descriptors1": f1["local_descriptors"],
                 "side_info0": f0["side_info"], "side_info1": f1["side_info"], "image0_size": size0, "image1_size": size1}
        thr = float(self.match_config["inference"]["match_threshold"])
        out = self.matcher.match(batch, thr, both_sides=False)                           # scores -> mutual-NN matches (inference.py:173-190)
        return features.compact_matches(out["matches0"], out["matching_scores0"], lafs0, lafs1)     # inference.py:192-209

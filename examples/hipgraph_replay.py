"""hipGraph replay of the whole hot path for launch-bound shapes.

og_forward only ENQUEUES work on the caller's stream (no allocation, no synchronisation), so the whole launch
sequence can be captured once into a hipGraph and replayed with a single host call -- useful when the HOST is
the bottleneck (busy Python thread, many small shapes).  Measured on an idle host at BASELINE config 1 (one
pair, 64 keypoints, ~60 launches): eager 344 us, graph replay 352 us -- the time is GPU-side kernel boundaries
(~1.5 us each plus few-microsecond kernels), which a graph does not remove; at config 2 the step is kernel-bound
anyway.  So this is a convenience, not the default path.  PyTorch is used for what it is here: stream capture and the
graph-private memory pool (torch.cuda.CUDAGraph is hipGraph on ROCm).

    gm = GraphedMatcher(model, example_data, match_threshold=0.2)
    out = gm(data)          # copies the inputs into the static buffers, replays, returns the static outputs
"""
from __future__ import annotations

from typing import Dict, Mapping

import torch

_KEYS = ("keypoints0", "keypoints1", "local_descriptors0", "local_descriptors1", "side_info0", "side_info1")


class GraphedMatcher:
    def __init__(self, model, example: Mapping, match_threshold: float = 0.2, both_sides: bool = True, warmup: int = 2):
        if model.training:
            raise RuntimeError("call model.eval() first")
        self.model = model
        self.static_in: Dict[str, object] = {k: example[k].detach().to(torch.float32).contiguous().clone() for k in _KEYS}
        for k in ("image0_size", "image1_size", "image0", "image1"):
            if k in example:
                self.static_in[k] = example[k]
        self.match_threshold, self.both_sides, self.warmup = match_threshold, both_sides, warmup
        self._capture()

    def _capture(self) -> None:
        model, dev = self.model, self.static_in["keypoints0"].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside the capture: packs weights, sizes the workspace
            for _ in range(self.warmup):
                model.match(self.static_in, self.match_threshold, both_sides=self.both_sides)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = model.match(self.static_in, self.match_threshold, both_sides=self.both_sides)
        # The graph bakes in raw pointers to the packed weights and the workspace, which live in the MODEL's caches:
        # keep our own references (a later model.match with another shape drops the model's workspace, a parameter
        # update re-packs), and remember which parameter state the captured blob belongs to.
        self._packed = model._packed
        self._packed_key = model._packed_key
        self._workspaces = list(model._workspace.values())

    @torch.no_grad()
    def __call__(self, data: Mapping) -> Dict[str, torch.Tensor]:
        for k in _KEYS:
            if data[k].shape != self.static_in[k].shape:
                raise ValueError(f"GraphedMatcher was captured for {k} of shape {tuple(self.static_in[k].shape)}")
        dev = self.static_in["keypoints0"].device
        if self.model._param_key(dev) != self._packed_key:      # parameters changed since the capture: stale weights
            self._capture()
        for k in _KEYS:
            self.static_in[k].copy_(data[k], non_blocking=True)
        self.graph.replay()
        return self.static_out

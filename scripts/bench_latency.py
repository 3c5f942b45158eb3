#!/usr/bin/env python3
"""Single-pair latency at BASELINE config 1 (64 kpts, 64-d, 2 stages, 3 Sinkhorn iterations): eager launches vs hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
from examples.hipgraph_replay import GraphedMatcher
dev = torch.device("cuda:0")
kw = dict(syn.CONFIGS["C1"]); (m, n), B = kw.pop("kpts"), kw.pop("batch")
cfg = syn.make_config(**kw); sd = syn.make_state_dict(cfg, 0)
model = SuperGlue(cfg).eval(); model.load_state_dict(sd); model.to(dev)
data = syn.make_batch(B, m, n, 64, 1, seed=0, device=dev)
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
eager = timeit(lambda: model.match(data, 0.2))
gm = GraphedMatcher(model, data, 0.2)
graphed = timeit(lambda: gm(data))
print(f"C1 single pair: eager {eager:.1f} us/pair ({1e6/eager:.0f} pairs/s), hipGraph replay {graphed:.1f} us/pair ({1e6/graphed:.0f} pairs/s)")

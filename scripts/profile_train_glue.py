#!/usr/bin/env python3
"""Attribute the GPU time of one training step to the torch operators (and their Python call sites) that launched it: which part of the
step is glue (copies, concatenations, adds, reductions) and where in openglue_amd/train.py it comes from.  Uses torch.profiler (kineto over
roctracer); informational, writes a table to stdout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
from oracle import superglue_oracle as orc            # only nll_criterion (the loss is the caller's code in the reference too)
from torch.profiler import profile, ProfilerActivity

B, N = int(os.environ.get("B", 4)), int(os.environ.get("N", 1024))
dev = torch.device("cuda:0")
cfg = syn.make_config(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=20)
model = SuperGlue(cfg); model.load_state_dict(syn.make_state_dict(cfg, seed=0)); model = model.to(dev).train()
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.make_batch(B, N, N, 256, 1, seed=1).items()}
gt0 = torch.full((B, N), -1, dtype=torch.long, device=dev); gt1 = torch.full((B, N), -1, dtype=torch.long, device=dev)
def step():
    model.zero_grad(set_to_none=True)
    loss = orc.nll_criterion(model(data)["scores"], gt0, gt1)
    loss.backward()
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
attr = "device_time_total" if hasattr(ka[0], "device_time_total") else "cuda_time_total"
self_attr = "self_device_time_total" if hasattr(ka[0], "self_device_time_total") else "self_cuda_time_total"
print(ka.table(sort_by=self_attr, row_limit=45, max_name_column_width=60))
try:
    ks = prof.key_averages(group_by_stack_n=6)
    rows = sorted(ks, key=lambda e: -getattr(e, self_attr))[:70]
    for e in rows:
        if getattr(e, self_attr) <= 0: continue
        site = [s for s in e.stack if "openglue_amd" in s or "oracle" in s][:2]
        print(f"{getattr(e, self_attr) / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:48]:48s} {' <- '.join(x.split('/')[-1] for x in site)}")
except Exception as ex:
    print("no stack grouping:", ex)

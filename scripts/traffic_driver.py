#!/usr/bin/env python3
"""Driver for the PMC traffic passes: one calibration copy of known size, then 2 hot-path steps at C2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
dev = torch.device("cuda:0")
x = torch.randn(32, 1024, 1024, device=dev)           # 134,217,728 bytes
torch.cuda.synchronize()
y = x.clone()                                          # calibration: reads 134 MB, writes 134 MB (elementwise copy kernel)
torch.cuda.synchronize()
CFG = os.environ.get("OG_TRAFFIC_CONFIG", "C2")          # C2 (default), C3, C4: the uniform BASELINE configs
kw = dict(syn.CONFIGS[CFG]); (m, n), B = kw.pop("kpts"), kw.pop("batch")
cfg = syn.make_config(**kw); sd = syn.make_state_dict(cfg, 0)
model = SuperGlue(cfg).eval(); model.load_state_dict(sd); model.to(dev)
data = syn.make_batch(B, m, n, kw['descriptor_dim'], kw['side_info_size'], seed=0, device=dev)
for _ in range(2):
    model.match(data, 0.2)
torch.cuda.synchronize()

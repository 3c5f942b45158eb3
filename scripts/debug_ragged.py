#!/usr/bin/env python3
"""Debug helper: ragged (packed) forward vs the per-pair uniform forward, per pair: max |d scores| and where."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
dev = torch.device("cuda:0")
cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=7, side_info_size=1)
sd = syn.make_state_dict(cfg, seed=0)
model = SuperGlue(cfg).eval(); model.load_state_dict(sd, strict=True); model = model.to(dev)
lens = [(5, 1100), (300, 40), (1025, 1030), (64, 64), (130, 257), (1, 1), (777, 512)]
if len(sys.argv) > 1: lens = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
pairs = []
for i, (m, n) in enumerate(lens):
    p = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.make_pair(m, n, 64, 1, seed=300 + i).items()}
    p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
    pairs.append(p)
for mode in ("0", "2"):
    os.environ["OG_SINKHORN_RESIDENT"] = mode
    packed = model.match_ragged(pairs, 0.2)
    os.environ["OG_SINKHORN_RESIDENT"] = "0"
    for p, q, (m, n) in zip(pairs, packed, lens):
        one = {k: (v[None] if torch.is_tensor(v) else v) for k, v in p.items()}
        ref = model.match(one, 0.2)
        d = (q["scores"] - ref["scores"][0]).abs()
        i = int(d.argmax()); r, c = divmod(i, n + 1)
        rows_bad = (d.amax(1) > 1e-4).nonzero().flatten().tolist()
        cols_bad = (d.amax(0) > 1e-4).nonzero().flatten().tolist()
        print(f"ragged resident={mode} pair {m}x{n}: max err {float(d.max()):.3e} at ({r},{c}); bad rows {len(rows_bad)} {rows_bad[:8]}; bad cols {len(cols_bad)} {cols_bad[:8]}; nan {int(torch.isnan(q['scores']).sum())}")

#!/usr/bin/env python3
"""Experiment: C2 batch split into S sub-batches issued on S HIP streams (complementary kernels overlap)."""
import os, sys, time, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openglue_amd import synthetic as syn
from openglue_amd.superglue import SuperGlue
dev = torch.device("cuda:0")
kw = dict(syn.CONFIGS["C2"]); (m, n), B = kw.pop("kpts"), kw.pop("batch")
cfg = syn.make_config(**kw); sd = syn.make_state_dict(cfg, 0)
data = syn.make_batch(B, m, n, 256, 1, seed=0, device=dev)
def run(S, stagger):
    models = []
    for i in range(S):
        mdl = SuperGlue(cfg).eval(); mdl.load_state_dict(sd); mdl.to(dev); models.append(mdl)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    parts = [{k: (v[i * B // S:(i + 1) * B // S] if torch.is_tensor(v) else v) for k, v in data.items()} for i in range(S)]
    def step():
        cur = torch.cuda.current_stream(dev)
        for i in range(S):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                models[i].match(parts[i], 0.2)
        for i in range(S):
            cur.wait_stream(streams[i])
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"streams={S}: {dt*1e3:.2f} ms per {B} pairs -> {B/dt:.0f} pairs/s")
for S in (1, 4, 8, 16):
    run(S, False)

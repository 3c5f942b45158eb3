#!/usr/bin/env python3
"""BUILD-CONTAINER measurement (needs /root/reference): the CPU oracle bench.py times as `cpu_baseline` (kind "port") against the
unmodified reference module on the same inputs, same thread count -> profiles/r04_port_vs_reference.json.  bench.py pastes the
file into its cpu_baseline object (tagged with the source): the GPU box has no /root/reference."""
import json, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, "/root/reference")
from openglue_amd import synthetic as syn
from oracle import superglue_oracle as orc
from models.superglue.superglue import SuperGlue as RefSuperGlue

threads = int(os.environ.get("OG_THREADS", str(os.cpu_count() or 8)))
torch.set_num_threads(threads)
out = {"_what": "ms per B=1 forward (scores incl. Sinkhorn; match extraction excluded on both sides), median of 5 after 1 warm-up",
       "threads": threads, "host": os.uname().nodename, "torch": torch.__version__,
       "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()}
for name in ("C1", "C2"):
    kw = dict(syn.CONFIGS[name]); (m, n), _ = kw.pop("kpts"), kw.pop("batch")
    cfg = syn.make_config(**kw); sd = syn.make_state_dict(cfg, seed=0)
    data = syn.make_batch(1, m, n, kw["descriptor_dim"], kw["side_info_size"], seed=0)
    ref = RefSuperGlue(cfg).eval(); ref.load_state_dict(sd, strict=True)
    def t(fn):
        fn(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[2] * 1e3
    with torch.no_grad():
        a = t(lambda: orc.superglue_forward(sd, cfg, data))
        b = t(lambda: ref(data))
        d = (orc.superglue_forward(sd, cfg, data)["scores"] - ref(data)["scores"]).abs().max().item()
    out[name] = {"workload": f"{m}x{n} kpts, {kw['descriptor_dim']}-dim, {kw['num_stages']} stages, {kw['num_iters']} Sinkhorn iters, B=1",
                 "oracle_port_ms": round(a, 2), "reference_ms": round(b, 2), "port_over_reference_time": round(a / b, 3), "max_abs_diff_scores": d}
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_port_vs_reference.json"), "w"), indent=1)

"""The RCCL path on the box the driver has: ONE GPU, backend "nccl" at world size 1.  The 2/4/8-GPU scaling run
is the driver's; what can be proven here is that the collective path loads RCCL, moves the payload and re-assembles it, both through
sharding.gather_matches(always_collective=True) and through bench.py (OG_BENCH_FORCE_DIST=1).  Each check runs in its own process:
a process group must not leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_GATHER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from openglue_amd import sharding, synthetic as syn
from openglue_amd.superglue import SuperGlue
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5, side_info_size=1)
model = SuperGlue(cfg).eval(); model.load_state_dict(syn.make_state_dict(cfg, seed=0)); model.to(dev)
data = syn.make_batch(5, 90, 70, 64, 1, seed=4, device=dev)
out = model.match(data, 0.2)
ids = [3, 0, 4, 1, 2]                       # a permuted shard: the gather must put every pair back in job order
local = {"matches0": out["matches0"][ids], "matching_scores0": out["matching_scores0"][ids]}
got = sharding.gather_matches(local, ids, 5, dst=0, always_collective=True, cap=5)
torch.cuda.synchronize()
assert torch.equal(got["matches0"], out["matches0"]) and torch.equal(got["matching_scores0"], out["matching_scores0"])
# an under-full shard (padding rows) and the cap agreed by all_reduce
got = sharding.gather_matches({k: v[:2] for k, v in local.items()}, ids[:2], 5, dst=0, always_collective=True)
assert torch.equal(got["matches0"][3], out["matches0"][3]) and torch.equal(got["matches0"][0], out["matches0"][0])
assert (got["matches0"][[1, 2, 4]] == -1).all()
print("NCCL_BACKEND", dist.get_backend(), "OK")
dist.destroy_process_group()
'''


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_gather_matches_over_rccl_world1(gpu_device):
    r = subprocess.run([sys.executable, "-c", _GATHER % ROOT], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NCCL_BACKEND nccl OK" in r.stdout


def test_bench_with_the_collective_forced(gpu_device):
    """bench.py with the RCCL path forced at world size 1: one JSON line, same metric, the step includes the gather."""
    env = _env()
    env.update(OG_BENCH_FORCE_DIST="1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "roofline" in line
    print("forced-collective bench:", line["value"], line["unit"], line["ms_per_step"], "ms/step")

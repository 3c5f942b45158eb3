"""The RCCL path.  On the box the driver usually has -- ONE GPU -- backend "nccl" at world size 1; the moment two or more
GPUs are visible, test_two_ranks_over_rccl_equal_one_gpu additionally runs the sharded job on 2 ranks under torch.distributed.run and
compares the gathered match lists with the single-GPU result.  The 2/4/8-GPU scaling run
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
    # round 5: the rccl block explains a scaling line by itself -- every rank's step time with and without the collective, the gather's own time
    rc = line["rccl"]
    assert rc["backend"] == "nccl" and rc["world_size"] == 1
    assert len(rc["per_rank_ms_compute_only"]) == 1 and len(rc["per_rank_ms_with_gather"]) == 1 and rc["per_rank_ms_compute_only"][0] > 0
    assert abs(rc["gather_ms_rank0"] - (rc["per_rank_ms_with_gather"][0] - rc["per_rank_ms_compute_only"][0])) < 1e-2
    print("forced-collective bench:", line["value"], line["unit"], line["ms_per_step"], "ms/step")


_TWO_RANKS = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from openglue_amd import sharding, synthetic as syn
from openglue_amd.superglue import SuperGlue
rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", lr); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
cfg = syn.make_config(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=20, side_info_size=1)
model = SuperGlue(cfg).eval(); model.load_state_dict(syn.make_state_dict(cfg, seed=0)); model.to(dev)
P = 7                                           # not a multiple of the world size: ranks get 4 and 3 pairs
shards = sharding.shard_pairs(P, world)
mine = shards[rank]
data = syn.make_batch(len(mine), 300, 260, 64, 1, seed=9, first_pair=mine[0], device=dev)       # pair i of the job is seeded by i
out = model.match(data, 0.2)
got = sharding.gather_matches({"matches0": out["matches0"], "matching_scores0": out["matching_scores0"]}, mine, P, dst=0)
torch.cuda.synchronize()
if rank == 0:
    whole = model.match(syn.make_batch(P, 300, 260, 64, 1, seed=9, first_pair=0, device=dev), 0.2)      # the same job on ONE GPU
    assert torch.equal(got["matches0"], whole["matches0"]), "gathered match lists differ from the single-GPU result"
    assert torch.equal(got["matching_scores0"], whole["matching_scores0"])
    print("RCCL_WORLD", dist.get_world_size(), dist.get_backend(), "OK", int((whole["matches0"] >= 0).sum()), "matches")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_over_rccl_equal_one_gpu(gpu_device, tmp_path):
    """>= 2 GPUs visible: the sharded job on 2 ranks (one process per GPU, torch.distributed.run, backend nccl = RCCL over xGMI), ONE
    gather of the match lists on rank 0, bit-identical to the same job on one GPU (pairs are independent: SURVEY.md 8e)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, {torch.cuda.device_count()} visible (the driver's scaling run covers N > 1 on the 8-GPU node)")
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANKS % ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(script)], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_WORLD 2 nccl OK" in r.stdout


def test_bench_two_gpus(gpu_device):
    """>= 2 GPUs visible: `python bench.py --gpus 2` re-executes itself under torch.distributed.run and prints ONE line with n_gpus = 2
    and the RCCL block filled in."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, {torch.cuda.device_count()} visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "8"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["backend"] == "nccl" and line["rccl"]["world_size"] == 2 and line["value"] > 0

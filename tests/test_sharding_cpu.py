"""CPU: the multi-GPU driver under gloo with world_size 2 (the N>1 path of bench.py uses the same
functions with backend nccl = RCCL).  The matcher itself is replaced by the oracle here -- this tests the
sharding and the single gather, not the kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openglue_amd import sharding, synthetic as syn
from oracle import superglue_oracle as orc
from tests.util import MATCH_THRESHOLD


def test_shard_pairs_uniform_and_balanced():
    assert sharding.shard_pairs(10, 4) == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert sharding.shard_pairs(2, 4) == [[0], [1], [], []]
    lens = syn.ragged_lengths(128, 512, 2048, seed=1)
    costs = [sharding.pair_cost(m, n) for m, n in lens]
    shards = sharding.shard_pairs(128, 8, costs)
    assert sorted(i for s in shards for i in s) == list(range(128))
    load = [sum(costs[i] for i in s) for s in shards]
    assert max(load) / (sum(load) / 8) < 1.05                 # LPT: within 5 % of perfect balance
    naive = [sum(costs[i] for i in s) for s in sharding.shard_pairs(128, 8)]
    assert max(load) <= max(naive)


def test_cost_balanced_shards_can_exceed_the_even_split():
    """LPT shards are not bounded by ceil(P / world); the gather capacity must be the real maximum."""
    shards = sharding.shard_pairs(4, 2, [10, 1, 1, 1])
    assert shards == [[0], [1, 2, 3]]
    assert sharding.max_shard(shards) == 3 > (4 + 1) // 2
    bad = 0
    for seed in range(50):
        lens = syn.ragged_lengths(24, 512, 2048, seed=seed)
        sh = sharding.shard_pairs(24, 8, [sharding.pair_cost(m, n) for m, n in lens])
        bad += sharding.max_shard(sh) > 3
    assert bad > 0           # the situation the old fixed capacity crashed on really occurs with BASELINE config 5 lengths


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5)
    sd = syn.make_state_dict(cfg, 0)
    data = syn.make_batch(num_pairs, 40, 40, 64, 1, seed=4)
    fn = lambda d: orc.match_pairs(sd, cfg, d, MATCH_THRESHOLD)
    got = sharding.match_sharded(fn, data, num_pairs)
    if rank == 0:
        q.put({k: v.clone() for k, v in got.items()})
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [5, 1])
def test_sharded_equals_unsharded_gloo_world2(num_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5)
    sd = syn.make_state_dict(cfg, 0)
    data = syn.make_batch(num_pairs, 40, 40, 64, 1, seed=4)
    want = orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
    assert torch.equal(got["matches0"], want["matches0"])
    assert torch.allclose(got["matching_scores0"], want["matching_scores0"], atol=1e-6)


def test_single_process_gather_is_a_reorder():
    m0 = torch.tensor([[1, -1], [0, 1]])
    s0 = torch.tensor([[0.5, 0.0], [0.9, 0.3]])
    out = sharding.gather_matches({"matches0": m0, "matching_scores0": s0}, [2, 0], 3)
    assert out["matches0"].tolist() == [[0, 1], [-1, -1], [1, -1]]


def _ragged_job():
    lens = [(61, 40), (9, 12), (11, 10), (8, 9), (10, 7)]
    costs = [10.0, 1.0, 1.0, 1.0, 1.0]          # skewed: rank 0 gets one pair, rank 1 four (> ceil(5 / 2))
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5)
    sd = syn.make_state_dict(cfg, 0)
    pairs = []
    for i, (m, n) in enumerate(lens):
        p = syn.make_pair(m, n, 64, 1, seed=50 + i)
        p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
        pairs.append(p)
    def run(ids):
        out = []
        for i in ids:
            one = {k: (v[None] if torch.is_tensor(v) else v) for k, v in pairs[i].items()}
            r = orc.match_pairs(sd, cfg, one, MATCH_THRESHOLD)
            out.append({"matches0": r["matches0"][0], "matching_scores0": r["matching_scores0"][0]})
        return out
    return lens, costs, run


def _ragged_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    lens, costs, run = _ragged_job()
    assert [len(x) for x in sharding.shard_pairs(len(lens), world, costs)] == [1, 4]
    got = sharding.match_sharded_ragged(run, lens, costs)
    if rank == 0:
        q.put({k: v.clone() for k, v in got.items()})
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_cost_balanced_gather_gloo_world2():
    """The ragged, cost-balanced gather of bench.py --config C5 with UNEVEN shards (1 and 4 pairs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lens, costs, run = _ragged_job()
    want = run(range(len(lens)))
    assert got["matches0"].shape == (len(lens), 61)
    for i, (m, _n) in enumerate(lens):
        assert torch.equal(got["matches0"][i, :m], want[i]["matches0"])
        assert (got["matches0"][i, m:] == -1).all()
        assert torch.allclose(got["matching_scores0"][i, :m], want[i]["matching_scores0"], atol=1e-6)


def test_gather_validates_its_arguments():
    m0 = torch.zeros(3, 4, dtype=torch.int64); s0 = torch.zeros(3, 4)
    with pytest.raises(ValueError, match="pair ids"):
        sharding.gather_matches({"matches0": m0, "matching_scores0": s0}, [0, 1], 3)
    with pytest.raises(ValueError, match="does not fit"):
        sharding.pad_ragged_matches([{"matches0": torch.zeros(9, dtype=torch.int64), "matching_scores0": torch.zeros(9)}], 8)


def test_eight_rank_dry_run_of_the_bench_plumbing():
    """`bench.py --gpus 8` end to end on gloo with the stub matcher (OG_BENCH_DRYRUN=1): self-spawn under torch.distributed.run on
    127.0.0.1, eight ranks, the LPT-balanced ragged shards, ONE gather, ONE JSON line from rank 0 with n_gpus = 8."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OG_BENCH_DRYRUN"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["gather_ok"] is True
    # the dry run drives bench.py's own _measure(): timed region -> per-step spread -> per-rank block, on EVERY rank (steps hold the collective).  With
    # round 4's rank-0-only spread these eight ranks would have dead-locked here instead of printing a line.
    assert d["step_ms_spread"]["steps"] == 2 and d["rccl"]["backend"] == "gloo" and d["rccl"]["world_size"] == 8
    assert len(d["rccl"]["per_rank_ms_with_gather"]) == 8 and len(d["rccl"]["per_rank_ms_compute_only"]) == 8
    # --global-batch: the BASELINE batch of a config split over the ranks (C3: 256 over 8), fixed job size = strong scaling
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--global-batch", "10"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["global_batch"] == 10 and d["scaling"] == "strong" and d["gather_ok"] is True
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--global-batch", "9"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                    # 9 pairs do not split over 2 ranks


def test_pair_cost_orders_pairs_like_the_measured_classes():
    """pair_cost is seconds-like and monotone; a 2048-keypoint pair costs more than four 1024-keypoint pairs' worth of attention
    would suggest less than 8x (the GEMM part is linear)."""
    c1, c2 = sharding.pair_cost(1024, 1024), sharding.pair_cost(2048, 2048)
    assert 2e-4 < c1 < 5e-4                      # the measured C2 step: 9.9 ms / 32 pairs = 0.31 ms per pair
    assert 2.0 * c1 < c2 < 8.0 * c1
    assert sharding.pair_cost(512, 2048) < sharding.pair_cost(2048, 2048)


def test_bench_never_runs_a_step_with_the_collective_on_rank_0_alone():
    """bench.py's `step` closures contain the RCCL gather when N > 1.  Anything that calls them must run on EVERY rank: round 4 computed `step_ms_spread`
    inside `if rank == 0:` -- with more than one rank, rank 0 would have waited in the gather for peers that had already moved on to the final barrier (never
    seen: no multi-GPU box was ever available, and the world-size-1 tests cannot show it).  Static check of the control flow."""
    import ast
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)

    def is_rank0_test(node):
        return (isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name) and node.test.left.id == "rank"
                and len(node.test.comparators) == 1 and isinstance(node.test.comparators[0], ast.Constant) and node.test.comparators[0].value == 0)

    offenders = []
    for node in ast.walk(tree):
        if is_rank0_test(node):
            for sub in ast.walk(ast.Module(body=node.body, type_ignores=[])):
                if isinstance(sub, ast.Call):
                    names = [a.id for a in sub.args if isinstance(a, ast.Name)] + ([sub.func.id] if isinstance(sub.func, ast.Name) else [])
                    if "step" in names:
                        offenders.append(sub.lineno)
    assert not offenders, f"bench.py: step (which holds the collective) is used inside `if rank == 0:` at lines {offenders}"
    # the uniform bench, the ragged bench and the dry run all go through _measure(), which every rank executes
    assert src.count("= _measure(step, ") == 3 and src.count("spread = _step_spread(step, args.steps, dev)") == 1

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

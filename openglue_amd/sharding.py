"""Multi-GPU driver for the matcher: image pairs are independent (eval-mode BatchNorm, no cross-pair
op anywhere on the path -- SURVEY.md §8e), so a job of P pairs is split into `world_size` shards, one
process per GPU, with NO collective on the data path and ONE gather of the match lists at the end
(RCCL over xGMI with backend "nccl"; each rank's message is KBs, i.e. latency-bound on its direct
link to the root).  The same code runs under gloo on CPU tensors, which is how it is tested here.

The reference's only parallelism is Lightning DDP for training (train.py:69-81); inference sharding is
new and deliberately minimal.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Mapping, Optional, Sequence

import torch
import torch.distributed as dist


def pair_cost(m: int, n: int, desc_dim: int = 256, num_stages: int = 9, iters: int = 100) -> float:
    """Estimated seconds of one pair on an MI355X: the algorithmic work of each kernel class over the rate that
    class MEASURES on this path (round 4, profiles/r04_*): split-f16 GEMMs incl. the fused message MLP ~330 algorithmic TFLOP/s, flash
    attention ~305, Sinkhorn on the resident schedule (ragged batches take it too since round 4) ~18 TB/s of the one-read-per-
    iteration equivalent.  Only the RATIOS matter: the figure balances ragged pairs over the ranks (LPT)."""
    D, L = desc_dim, num_stages
    gemm = L * 40.0 * D * D * (m + n) + 2.0 * D * D * (m + n) + 2.0 * m * n * D
    attn = L * (4.0 * D * (m * m + n * n) + 8.0 * D * m * n)
    sink = 4.0 * (m * n * (iters + 1) + (m + 1) * (n + 1))
    return gemm / 330e12 + attn / 305e12 + sink / 18e12


def shard_pairs(num_pairs: int, world_size: int, costs: Optional[Sequence[float]] = None) -> List[List[int]]:
    """Pair indices per rank.  Uniform pairs: contiguous equal split (remainder to the low ranks).
    Ragged pairs (`costs` given): longest-processing-time greedy, then indices sorted per rank."""
    if world_size <= 0 or num_pairs < 0:
        raise ValueError("bad world_size / num_pairs")
    if costs is None:
        base, rem = divmod(num_pairs, world_size)
        out, start = [], 0
        for r in range(world_size):
            cnt = base + (1 if r < rem else 0)
            out.append(list(range(start, start + cnt)))
            start += cnt
        return out
    if len(costs) != num_pairs:
        raise ValueError("len(costs) != num_pairs")
    load = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in sorted(range(num_pairs), key=lambda i: (-costs[i], i)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def take_pairs(data: Mapping, idx: Sequence[int]) -> Dict:
    """Sub-batch of a data dict (tensors indexed on dim 0; image sizes passed through)."""
    sel = torch.as_tensor(list(idx), dtype=torch.long)
    return {k: (v.index_select(0, sel.to(v.device)) if torch.is_tensor(v) and v.dim() > 0 and not k.startswith("image") else v)
            for k, v in data.items()}


_IDS_CACHE: Dict[tuple, torch.Tensor] = {}


def _ids_on(device: torch.device, pair_ids: Sequence[int]) -> torch.Tensor:
    """Pair ids as a device tensor, cached: a steady-state caller (bench.py, a serving loop) passes the same shard every
    step and must not pay a host-to-device copy (and its synchronisation) per step."""
    key = (str(device), tuple(int(i) for i in pair_ids))
    t = _IDS_CACHE.get(key)
    if t is None:
        if len(_IDS_CACHE) > 64:
            _IDS_CACHE.clear()
        t = torch.as_tensor(list(key[1]), dtype=torch.long, device=device)
        _IDS_CACHE[key] = t
    return t


def max_shard(shards: Sequence[Sequence[int]]) -> int:
    """Rows every rank's gather payload is padded to: the LARGEST shard (cost-balanced shards are not bounded by
    ceil(num_pairs / world): shard_pairs(4, 2, [10, 1, 1, 1]) = [[0], [1, 2, 3]])."""
    return max((len(x) for x in shards), default=0)


def gather_matches(local: Mapping[str, torch.Tensor], pair_ids: Sequence[int], num_pairs: int, dst: int = 0,
                   group=None, always_collective: bool = False, cap: Optional[int] = None) -> Optional[Dict[str, torch.Tensor]]:
    """The one collective: every rank contributes matches0 [b, m] (int64) and matching_scores0 [b, m]
    (fp32) of its shard; rank `dst` returns them re-assembled in job order [num_pairs, m]; others None.
    Shards are padded to `cap` rows = the largest shard of the job (max_shard(shard_pairs(...)), identical on every
    rank), so a single fixed-size gather suffices; without `cap` the ranks agree on it with one all_reduce(MAX), which
    synchronises the host -- steady-state callers pass it.  Otherwise nothing here synchronises the host with the device
    (no boolean-mask indexing, no per-call host-to-device copy): the collective and the re-assembly are only enqueued
    behind the kernels that produced the matches."""
    m0, s0 = local["matches0"], local["matching_scores0"]
    dev = m0.device
    width = m0.shape[1]
    b = m0.shape[0]
    if len(pair_ids) != b:
        raise ValueError(f"{b} rows of matches for {len(pair_ids)} pair ids")
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        order = _ids_on(dev, pair_ids)
        out_m = torch.full((num_pairs, width), -1, dtype=torch.int64, device=dev)
        out_s = torch.zeros((num_pairs, width), dtype=torch.float32, device=dev)
        out_m[order], out_s[order] = m0, s0
        return {"matches0": out_m, "matching_scores0": out_s}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if width >= (1 << 24) or num_pairs >= (1 << 24):
        raise ValueError("index does not fit the packed fp32 payload")
    if cap is None:
        t = torch.tensor([b], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        cap = int(t.item())
    if b > cap:          # would otherwise surface as a shape error on ONE rank while the others block in the collective
        raise ValueError(f"rank {rank}: shard of {b} pairs exceeds the gather capacity {cap}; pass cap=max_shard(shards)")
    ids = _ids_on(dev, pair_ids)
    # one packed fp32 payload per rank: [cap, 1 + 2*width] = pair id | matches (exact in fp32 below 2^24) | scores;
    # padding rows carry pair id -1
    if b == cap:                                   # the common case (equal shards): a single fused concatenation
        payload = torch.cat([ids.to(torch.float32)[:, None], m0.to(torch.float32), s0], dim=1)
    else:
        payload = torch.full((cap, 1 + 2 * width), -1.0, dtype=torch.float32, device=dev)
        if b:
            payload[:b, 0] = ids.to(torch.float32)
            payload[:b, 1:1 + width] = m0.to(torch.float32)
            payload[:b, 1 + width:] = s0
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    allp = torch.cat(bufs, 0)
    gid = allp[:, 0].to(torch.long)
    # padding rows (id -1) are scattered into one extra trash row instead of being masked out (masking = nonzero = a sync)
    slot = torch.where(gid >= 0, gid, torch.full_like(gid, num_pairs))
    out_m = torch.full((num_pairs + 1, width), -1, dtype=torch.int64, device=dev)
    out_s = torch.zeros((num_pairs + 1, width), dtype=torch.float32, device=dev)
    out_m.index_copy_(0, slot, allp[:, 1:1 + width].to(torch.int64))
    out_s.index_copy_(0, slot, allp[:, 1 + width:].contiguous())
    return {"matches0": out_m[:num_pairs], "matching_scores0": out_s[:num_pairs]}


def match_sharded(match_fn: Callable[[Mapping], Mapping[str, torch.Tensor]], data: Mapping, num_pairs: int,
                  costs: Optional[Sequence[float]] = None, dst: int = 0, group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Run `match_fn` (e.g. SuperGlue.match bound to this rank's GPU) on this rank's shard of a job whose
    full input `data` every rank can index, then gather the match lists on `dst`."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    shards = shard_pairs(num_pairs, world, costs)
    mine = shards[rank]
    local = match_fn(take_pairs(data, mine)) if mine else None
    if local is None:
        width = data["keypoints0"].shape[1]
        dev = data["keypoints0"].device
        local = {"matches0": torch.empty(0, width, dtype=torch.int64, device=dev),
                 "matching_scores0": torch.empty(0, width, dtype=torch.float32, device=dev)}
    return gather_matches(local, mine, num_pairs, dst, group, cap=max_shard(shards))


def pad_ragged_matches(results: Sequence[Mapping[str, torch.Tensor]], width: int) -> Dict[str, torch.Tensor]:
    """Per-pair match lists of different lengths (SuperGlue.match_ragged) -> one [b, width] block, padded with -1 / 0:
    the fixed-size payload the single gather needs."""
    dev = results[0]["matches0"].device if results else torch.device("cpu")
    m0 = torch.full((len(results), width), -1, dtype=torch.int64, device=dev)
    s0 = torch.zeros((len(results), width), dtype=torch.float32, device=dev)
    for i, r in enumerate(results):
        k = r["matches0"].numel()
        if k > width:
            raise ValueError(f"pair with {k} keypoints does not fit the gather width {width}")
        m0[i, :k] = r["matches0"]
        s0[i, :k] = r["matching_scores0"]
    return {"matches0": m0, "matching_scores0": s0}


def match_sharded_ragged(match_list_fn: Callable[[Sequence[int]], Sequence[Mapping[str, torch.Tensor]]],
                         lens: Sequence[Sequence[int]], costs: Optional[Sequence[float]] = None, dst: int = 0, group=None,
                         always_collective: bool = False, device: Optional[torch.device] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Ragged job (BASELINE config 5): pair i has lens[i] = (m_i, n_i) keypoints.  Pairs are cost-balanced over the ranks
    (LPT on `costs`, default pair_cost(m_i, n_i)), every rank runs `match_list_fn(its pair ids)` -> one result dict per
    pair, and the match lists travel to `dst` in ONE gather, padded to the longest image-0 keypoint set of the JOB."""
    num_pairs = len(lens)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if costs is None:
        costs = [pair_cost(m, n) for m, n in lens]
    shards = shard_pairs(num_pairs, world, costs)
    mine = shards[rank]
    width = max(m for m, _ in lens)
    res = list(match_list_fn(mine)) if mine else []
    local = pad_ragged_matches(res, width)
    if not res and device is not None:
        local = {k: v.to(device) for k, v in local.items()}
    return gather_matches(local, mine, num_pairs, dst, group, always_collective=always_collective, cap=max_shard(shards))

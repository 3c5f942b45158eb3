#!/usr/bin/env python3
"""Benchmark of the SuperGlue hot path on MI355X: matched image-pairs / second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (keypoint encoder -> 9x(self, cross) attention -> scores ->
100 Sinkhorn iterations -> mutual-NN matches) over one batch of synthetic pairs already resident in
HBM, plus (N > 1) the single RCCL gather of the match lists on rank 0.  Weak scaling: every rank
processes its own batch of the configured size.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from openglue_amd import _lib, sharding, synthetic as syn          # noqa: E402
from openglue_amd.superglue import SuperGlue                       # noqa: E402

MATCH_THRESHOLD = 0.2
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA
PEAK_HBM_GBS = 8000.0            # HBM3E spec


def algorithmic_counts(cfg_kw, m, n):
    """Per-pair algorithmic work, formulas of SURVEY.md §8(d) / BASELINE.md §3."""
    D, L, it = cfg_kw["descriptor_dim"], cfg_kw["num_stages"], cfg_kw["num_iters"]
    sizes = [2 + cfg_kw["side_info_size"], 32, 64, 128, D]
    enc = 2.0 * (m + n) * sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    proj = L * 40.0 * D * D * (m + n)
    attn = L * (4.0 * D * (m * m + n * n) + 8.0 * D * m * n)
    final = 2.0 * D * D * (m + n)
    score = 2.0 * m * n * D
    sink_bytes = 4.0 * ((m + 1) * (n + 1) * (2 * it + 1) + 2 * m * n)
    # kernel classes: the encoder MLP runs on the exact-fp32 MFMA kernel; the GNN 1x1 convs, the final projection and the
    # score matrix on the split-f16 kernel
    return {"gemm_f32_flops": enc, "gemm_f16x3_flops": proj + final + score, "attention_flops": attn,
            "total_flops": enc + proj + attn + final + score, "sinkhorn_bytes": sink_bytes}


def profiled_forward(model, data, thr):
    """One og_forward_profiled call through the model's own buffers -> {stage: (ms, launches)}."""
    lib = _lib.load()
    out = model.match(data, thr)            # makes sure weights are packed / workspace exists
    dev = data["keypoints0"].device
    B, m, _ = data["keypoints0"].shape
    n = data["keypoints1"].shape[1]
    shape = model._shape(B, m, n, thr)
    t = {k: data[k].contiguous() for k in ("keypoints0", "keypoints1", "local_descriptors0", "local_descriptors1", "side_info0", "side_info1")}
    inp = _lib.og_inputs(t["keypoints0"].data_ptr(), t["keypoints1"].data_ptr(), t["local_descriptors0"].data_ptr(),
                         t["local_descriptors1"].data_ptr(), t["side_info0"].data_ptr(), t["side_info1"].data_ptr())
    inp.image0_wh[0], inp.image0_wh[1] = data["image0_size"][:2]
    inp.image1_wh[0], inp.image1_wh[1] = data["image1_size"][:2]
    o = _lib.og_outputs(out["scores"].data_ptr(), out["context_descriptors0"].data_ptr(), out["context_descriptors1"].data_ptr(),
                        out["matches0"].data_ptr(), out["matching_scores0"].data_ptr(), out["matches1"].data_ptr(),
                        out["matching_scores1"].data_ptr())
    ms = (C.c_float * len(_lib.OG_STAGES))()
    cnt = (C.c_int32 * len(_lib.OG_STAGES))()
    ws = next(iter(model._workspace.values()))
    rc = lib.og_forward_profiled(C.byref(shape), C.byref(inp), model._packed.data_ptr(), ws.data_ptr(), C.byref(o),
                                 torch.cuda.current_stream(dev).cuda_stream, ms, cnt)
    _lib.check(rc, "og_forward_profiled")
    return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.OG_STAGES)}


def cpu_baseline(cfg, sd, cfg_kw, m, n, budget_s=20.0):
    """The CPU oracle (a torch-CPU port of the reference algorithm, oracle/superglue_oracle.py) on this
    box's host cores, B=1 pairs of the same workload, bounded to ~budget_s seconds.  The thread count is
    picked by a short probe (torch's intra-op pool oversubscribes badly on 256-thread hosts: all 256
    threads ran 250 s/pair); `cores` reports the threads actually used."""
    from oracle import superglue_oracle as orc
    ncpu = os.cpu_count() or 1
    data = syn.make_batch(1, m, n, cfg_kw["descriptor_dim"], cfg_kw["side_info_size"], seed=0)
    probe_kw = dict(cfg_kw, num_stages=1, num_iters=4)
    pcfg = syn.make_config(**probe_kw)
    psd = {k: v for k, v in sd.items()}
    best_t, best_dt = 1, float("inf")
    with torch.no_grad():
        for t in [c for c in (4, 8, 16, 32, 64) if c <= ncpu] or [1]:
            torch.set_num_threads(t)
            orc.superglue_forward(psd, pcfg, data)
            t0 = time.perf_counter(); orc.superglue_forward(psd, pcfg, data); dt = time.perf_counter() - t0
            if dt < best_dt:
                best_t, best_dt = t, dt
            if dt > 5.0:
                break
        torch.set_num_threads(best_t)
        t0 = time.perf_counter(); orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD); warm = time.perf_counter() - t0
        reps = max(1, min(5, int(budget_s / max(warm, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
        dt = (time.perf_counter() - t0) / reps
    return {"value": round(1.0 / dt, 4), "unit": "image-pairs/s", "cores": best_t, "kind": "port",
            "sample": f"{reps} x 1 pair of the same workload (B=1, torch-CPU oracle, {best_t} of {ncpu} host threads), {dt * 1e3:.0f} ms/pair"}


def bench_ragged(args, world, rank, dev, dist_on=False):
    """BASELINE configs[4]: 128 pairs with 512-2048 keypoints per image, cost-balanced over the ranks, through the
    token-packed ragged path (SuperGlue.match_ragged -> og_forward_ragged)."""
    kw = dict(syn.CONFIGS["C2"]); kw.pop("kpts"); kw.pop("batch")
    total = (args.batch or 16) * world
    lens = syn.ragged_lengths(total, 512, 2048, seed=0)
    costs = [sharding.pair_cost(m, n) for m, n in lens]
    mine = sharding.shard_pairs(total, world, costs)[rank]
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=0)
    model = SuperGlue(cfg).eval(); model.load_state_dict(sd, strict=True); model.to(dev)
    pairs = []
    for i in mine:
        p = syn.make_pair(lens[i][0], lens[i][1], 256, 1, seed=i)
        p = {k: v.to(dev) for k, v in p.items()}
        p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
        pairs.append(p)

    def step():
        res = model.match_ragged(pairs, MATCH_THRESHOLD, both_sides=False)
        if dist_on:   # the one collective: match lists to rank 0, padded to the longest keypoint set (2048)
            width = 2048
            m0 = torch.full((len(res), width), -1, dtype=torch.int64, device=dev)
            s0 = torch.zeros((len(res), width), dtype=torch.float32, device=dev)
            for i, r in enumerate(res):
                m0[i, :r["matches0"].numel()] = r["matches0"]; s0[i, :r["matches0"].numel()] = r["matching_scores0"]
            sharding.gather_matches({"matches0": m0, "matching_scores0": s0}, mine, total, dst=0, always_collective=True)
        return res
    for _ in range(args.warmup):
        step()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        line = {"metric": "image-pairs/sec (C5 ragged 512-2048 kpts)", "value": round(total * args.steps / dt, 2), "unit": "image-pairs/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4]: {total} ragged pairs, 512-2048 kpts/image, 256-dim, 9 stages, 100 Sinkhorn iters, "
                                       "token-packed ragged kernels (og_forward_ragged), LPT cost-balanced over ranks",
                           "mean_kpts": round(sum(a + b for a, b in lens) / (2 * total), 1)},
                "roofline": None, "cpu_baseline": None,
                "valid_matches_per_pair": round(sum(int((r["matches0"] >= 0).sum()) for r in res) / max(1, len(res)), 1)}
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(syn.CONFIGS) + ["C5"])
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or os.environ.get("OG_BENCH_FORCE_DIST") == "1"    # the latter: smoke-test the RCCL path on one GPU
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.config == "C5":
        return bench_ragged(args, world, rank, dev, dist_on)
    kw = dict(syn.CONFIGS[args.config])
    (m, n), B = kw.pop("kpts"), kw.pop("batch")
    if args.batch:
        B = args.batch
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=0)
    model = SuperGlue(cfg).eval()
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    data = syn.make_batch(B, m, n, kw["descriptor_dim"], kw["side_info_size"], seed=0, first_pair=rank * B, device=dev)
    pair_ids = list(range(rank * B, rank * B + B))

    def step():
        out = model.match(data, MATCH_THRESHOLD, both_sides=True)
        if dist_on:
            sharding.gather_matches(out, pair_ids, world * B, dst=0, always_collective=True)
        return out

    for _ in range(args.warmup):
        step()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        counts = algorithmic_counts(kw, m, n)
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        # per-kernel-class time of one step, HIP events on the launch stream (median of 3 profiled steps)
        profs = [profiled_forward(model, data, MATCH_THRESHOLD) for _ in range(3)]
        stages = {k: sorted(p[k][0] for p in profs)[1] for k in profs[0]}
        launches = {k: profs[0][k][1] for k in profs[0]}
        per_step = {  # kernel class -> (algorithmic work per step, unit scale, bound, peak, kernel names, note)
            "gemm_f16x3": (counts["gemm_f16x3_flops"] * B, 1e12, "mfma", PEAK_F16_MFMA_TFLOPS, "TFLOP/s",
                           "gemm_nt_f16x3_big_kernel / gemm_nt_f16x3_kernel (GNN 1x1 convs, final projection, score matrix; split-f16 3-pass MFMA: executes 3x the algorithmic flops)"),
            "gemm_f32": (counts["gemm_f32_flops"] * B, 1e12, "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                         "gemm_nt_f32_kernel (keypoint-encoder MLP; exact fp32 MFMA)"),
            "attention": (counts["attention_flops"] * B, 1e12, "mfma", PEAK_F16_MFMA_TFLOPS, "TFLOP/s",
                          "attention_kernel (split-f16 flash attention: executes 3x the algorithmic flops)"),
            "sinkhorn": (counts["sinkhorn_bytes"] * B, 1e9, "hbm", PEAK_HBM_GBS, "GB/s",
                         "sinkhorn_sweep_fast + sinkhorn_combine_fast (stage time incl. launch gaps; the kernels read S ONCE per iteration, the algorithmic figure of SURVEY 8d counts two sweeps: frac can exceed 1)"),
        }
        # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # corrected as MI355X_MICROARCH.md prescribes); only valid for the configuration it was measured on (C2, B=32)
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_c2.json")
        if args.config == "C2" and B == 32 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = {"gemm_f16x3": tj.get("gemm_f16x3"), "gemm_f32": tj.get("gemm_f32"), "attention": tj.get("attention"),
                       "sinkhorn": {"hbm_bytes_per_launch": (tj["sinkhorn_sweep"]["hbm_bytes_per_launch"] + tj["sinkhorn_combine"]["hbm_bytes_per_launch"]) * kw["num_iters"]}}
        roofs = {}
        for k, (work, scale, bound, peak, unit, kern) in per_step.items():
            ms = stages[k]
            ach = work / (ms * 1e-3) / scale if ms > 0 else 0.0
            nl = max(1, launches[k])
            roofs[k] = {"kernel": kern, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4),
                        "traffic": (traffic.get(k) or {}).get("hbm_bytes_per_launch"), "class_ms_per_step": round(ms, 3),
                        "launches_per_step": launches[k], "avg_launch_ms": round(ms / nl, 4),
                        "algorithmic_work_per_launch": round(work / nl / scale, 6)}
        dominant = max(per_step, key=lambda k: stages[k])
        roof = roofs.pop(dominant)
        roof2 = roofs
        line = {
            "metric": "image-pairs/sec (1024 kpts, 256-dim, 9 GNN layers)" if args.config == "C2" else f"image-pairs/sec ({args.config})",
            "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{'1' if args.config == 'C2' else args.config}]: {m}x{n} kpts, {kw['descriptor_dim']}-dim, "
                                   f"{kw['num_stages']} self+cross stages, {kw['num_heads']} heads, {kw['num_iters']} Sinkhorn iters, "
                                   f"batch={B} pairs/GPU, random-init weights, seeded synthetic keypoints/descriptors",
                       "pairs_per_gpu": B, "kpts": [m, n], "parallelism": f"pairs sharded over {world} GPU(s), 1 RCCL gather"},
            "roofline": roof, "roofline_other": roof2,
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "algorithmic": {"gflop_per_pair": round(counts["total_flops"] / 1e9, 2), "sinkhorn_gb_per_pair": round(counts["sinkhorn_bytes"] / 1e9, 3)},
            "valid_matches_per_pair": round(float((out["matches0"] >= 0).sum().item()) / B, 1),
        }
        # the same step fed from pinned HOST buffers (H2D of keypoints/descriptors/side-info included, synchronous with
        # the step: the boundary hands over device tensors, so this is informational and never `value`)
        host = {k: v.cpu().pin_memory() for k, v in data.items() if torch.is_tensor(v)}
        def step_h2d():
            d = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
            d["image0_size"], d["image1_size"] = data["image0_size"], data["image1_size"]
            return model.match(d, MATCH_THRESHOLD, both_sides=True)
        step_h2d(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_h2d()
        torch.cuda.synchronize()
        line["value_incl_h2d"] = round(B * args.steps / (time.perf_counter() - t0), 2)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, kw, m, n)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

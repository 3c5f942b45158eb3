#!/usr/bin/env python3
"""Benchmark of the SuperGlue hot path on MI355X: matched image-pairs / second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C1|C3|C4|C5] [--no-cpu-baseline]

With --gpus N > 1 and no torch.distributed environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one process per GPU,
like the reference's Lightning DDP, train.py:69-81); launched by the driver under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A step = one pass of the whole hot path (keypoint encoder -> 9x(self, cross) attention -> scores ->
100 Sinkhorn iterations -> mutual-NN matches) over one batch of synthetic pairs already resident in
HBM, plus (N > 1) the single RCCL gather of the match lists on rank 0.  Weak scaling: every rank
processes its own batch of the configured size.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
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
# What the whole chip SUSTAINS of the f16 matrix pipe with real (non-zero) operands: 256 CUs issuing nothing but
# v_mfma_f32_32x32x16_f16 settle at a 1.65 GHz shader clock = 1670 TFLOP/s (zeros: 2.40 GHz, 2496); scripts/probes/mfma_power.hip,
# profiles/r02_probe_mfma_power.log.  `frac` stays relative to the guide's 2.5 PFLOP/s; `frac_of_sustained` is reported beside it.
SUSTAINED_F16_MFMA_TFLOPS = 1670.0
# Counter summaries collected by SEPARATE rocprofv3 --pmc passes (scripts/gpu_pmc.sh + parse_pmc.py, scripts/gpu_traffic.sh +
# parse_traffic.py) and committed under profiles/.  They are NOT measured by this run: every block pasted from them into the JSON
# line carries its source file and the commit it was collected on.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
TRAFFIC_SUMMARY = os.path.join(ROOT, "profiles", "r03_traffic_c2.json")
TRAFFIC_BY_CONFIG = {"C3": os.path.join(ROOT, "profiles", "r03_traffic_c3.json"), "C4": os.path.join(ROOT, "profiles", "r03_traffic_c4.json")}


def algorithmic_counts(cfg_kw, m, n):
    """Per-pair algorithmic work, formulas of SURVEY.md §8(d) / BASELINE.md §3."""
    D, L, it = cfg_kw["descriptor_dim"], cfg_kw["num_stages"], cfg_kw["num_iters"]
    sizes = [2 + cfg_kw["side_info_size"], 32, 64, 128, D]
    enc = 2.0 * (m + n) * sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    proj = L * 40.0 * D * D * (m + n)          # per token and stage: q/k/v 12 D^2, out_proj 4 D^2, fc.0 16 D^2, fc.3 8 D^2
    mlp = L * 28.0 * D * D * (m + n)           # ... of which the message MLP incl. out_proj (folded into fc.0 at pack time)
    attn = L * (4.0 * D * (m * m + n * n) + 8.0 * D * m * n)
    final = 2.0 * D * D * (m + n)
    score = 2.0 * m * n * D
    # SURVEY §8d counts TWO sweeps of the augmented matrix per iteration (the reference's structure: row LSE, column LSE)
    sink_survey = 4.0 * ((m + 1) * (n + 1) * (2 * it + 1) + 2 * m * n)
    # what a STREAMING schedule has to move: ONE read of S per iteration (row pass and column pass share the sweep), the
    # per-row-block column partials (written by the sweep, read by the combine), the final read of S and the scores write
    rb = (m + 31) // 32
    sink_one = 4.0 * (m * n * (it + 1) + 2 * rb * n * it + (m + 1) * (n + 1))
    return {"gemm_f32_flops": enc, "gemm_f16x3_flops": proj + final + score, "mlp_flops": mlp, "attention_flops": attn,
            "total_flops": enc + proj + attn + final + score, "sinkhorn_bytes": sink_one, "sinkhorn_bytes_survey": sink_survey}


def _sum_counts(cfg_kw, lens):
    tot = None
    for m, n in lens:
        c = algorithmic_counts(cfg_kw, m, n)
        tot = c if tot is None else {k: tot[k] + c[k] for k in c}
    return tot


def _load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def roofline_block(counts_per_step, stages, launches, num_iters, measured_on_this_workload, sinkhorn_resident, traffic_path=None):
    """Per-kernel-class roofline objects.  achieved = algorithmic work per step / class time (HIP events on the launch
    stream, median of 3 profiled steps) = algorithmic work per launch / average launch duration.
    sinkhorn_resident: the schedule the library took for this shape (og_sinkhorn_schedule), not a guess from bracket counts."""
    pmc = _load_json(PMC_SUMMARY) if measured_on_this_workload else {}
    tpath = TRAFFIC_SUMMARY if measured_on_this_workload else traffic_path      # counter summaries exist for C2 (PMC + traffic) and C3 / C4 (traffic)
    tj = _load_json(tpath) if tpath else {}
    src_pmc = {"source": os.path.relpath(PMC_SUMMARY, ROOT), "collected_at_commit": pmc.get("_commit"), "measured_in_this_run": False}
    src_tr = {"source": os.path.relpath(tpath, ROOT) if tpath else None, "collected_at_commit": tj.get("_commit"), "measured_in_this_run": False}
    # the split-f16 GEMM class = the stand-alone GEMM launches + the fused message-MLP launches (same arithmetic, same pipe)
    gemm_ms = stages["gemm_f16x3"] + stages.get("mlp_fused", 0.0)
    gemm_launches = launches["gemm_f16x3"] + launches.get("mlp_fused", 0)
    cls_ms = {"gemm_f16x3": gemm_ms, "gemm_f32": stages["gemm_f32"], "attention": stages["attention"], "sinkhorn": stages["sinkhorn"]}
    # real kernel launches of the Sinkhorn bracket: resident = first-iteration sweep + combine, resident kernel, safety net, scores
    sk_launches = 5 if sinkhorn_resident else 2 * num_iters + 1
    cls_launches = {"gemm_f16x3": gemm_launches, "gemm_f32": launches["gemm_f32"], "attention": launches["attention"], "sinkhorn": sk_launches}
    per_step = {  # kernel class -> (algorithmic work per step, unit scale, bound, peak, unit, kernels)
        "gemm_f16x3": (counts_per_step["gemm_f16x3_flops"], 1e12, "mfma", PEAK_F16_MFMA_TFLOPS, "TFLOP/s",
                       "mlp_fused_kernel (fc.0 -> ReLU -> fc.3 + residual, hidden activation in registers) / gemm_nt_f16x3_big2_kernel (256x256 tiles: q/k/v) / "
                       "gemm_nt_f16x3_kernel (128-token tiles) / gemm_nt_f16x3_big_kernel (batched score matrix): GNN 1x1 convs, last encoder conv, "
                       "final projection, score matrix; split-f16 3-pass MFMA: executes 3x the algorithmic flops"),
        "gemm_f32": (counts_per_step["gemm_f32_flops"], 1e12, "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                     "gemm_nt_f32_kernel (keypoint-encoder MLP without its last conv; exact fp32 MFMA)"),
        "attention": (counts_per_step["attention_flops"], 1e12, "mfma", PEAK_F16_MFMA_TFLOPS, "TFLOP/s",
                      "attention_dma_kernel (dh = 64, 32: K/V tiles by LDS-DMA) / attention_kernel (dh = 16): split-f16 flash attention, executes 3x the algorithmic flops"),
    }
    roofs = {}
    for k, (work, scale, bound, peak, unit, kern) in per_step.items():
        ms = cls_ms[k]
        ach = work / (ms * 1e-3) / scale if ms > 0 else 0.0
        nl = max(1, cls_launches[k])
        roofs[k] = {"kernel": kern, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": None, "class_ms_per_step": round(ms, 3),
                    "launches_per_step": cls_launches[k], "avg_launch_ms": round(ms / nl, 4),
                    "algorithmic_work_per_launch": round(work / nl / scale, 6)}
        if bound == "mfma" and peak == PEAK_F16_MFMA_TFLOPS:
            # the split-f16 kernels execute 3 MFMA passes per algorithmic product: their ceiling is a third of the pipe's rate
            roofs[k]["executed_tflops"] = round(3.0 * ach, 1)
            roofs[k]["sustained_peak_measured"] = SUSTAINED_F16_MFMA_TFLOPS
            roofs[k]["executed_frac_of_sustained"] = round(3.0 * ach / SUSTAINED_F16_MFMA_TFLOPS, 4)
        if k in tj and isinstance(tj[k], dict) and "hbm_bytes_per_launch" in tj[k]:
            roofs[k]["traffic"] = tj[k]["hbm_bytes_per_launch"]
            roofs[k]["traffic_source"] = src_tr
        if k in pmc:       # SQ counter summary of the same kernels (separate rocprofv3 --pmc passes)
            roofs[k]["mfma_busy_frac"] = pmc[k].get("mfma_busy_frac")
            roofs[k]["pmc"] = dict({kk: vv for kk, vv in pmc[k].items() if kk != "mfma_busy_frac"}, **src_pmc)
    if stages.get("mlp_fused", 0.0) > 0:
        g = roofs["gemm_f16x3"]
        mlp_ms, mlp_n = stages["mlp_fused"], max(1, launches["mlp_fused"])
        g["mlp_fused"] = {"ms_per_step": round(mlp_ms, 3), "launches_per_step": launches["mlp_fused"], "avg_launch_ms": round(mlp_ms / mlp_n, 4),
                          "algorithmic_tflops": round(counts_per_step["mlp_flops"] / (mlp_ms * 1e-3) / 1e12, 1)}
        g["standalone_gemms"] = {"ms_per_step": round(stages["gemm_f16x3"], 3), "launches_per_step": launches["gemm_f16x3"],
                                 "algorithmic_tflops": round((counts_per_step["gemm_f16x3_flops"] - counts_per_step["mlp_flops"]) / max(stages["gemm_f16x3"] * 1e-3, 1e-9) / 1e12, 1)}
    # ---- Sinkhorn.  Streaming schedule: HBM roofline on what it has to move (one read of S per iteration + column partials + scores).
    #      Resident schedule: the matrices stay on chip, so there is no meaningful HBM roofline; the kernel is bound by vector-ALU
    #      issue.  Reported: SURVEY 8(d) bytes / time (survey_equivalent: may exceed the HBM peak -- that is the point), counter bytes
    #      / time as the HBM fraction (from the committed traffic summary, tagged), and the VALU-issue fraction from the committed PMC
    #      summary (SQ_INSTS_VALU x 4 cycles / (busy cycles x SIMDs)) as the bounding figure.
    ms = stages["sinkhorn"]
    sk = {"schedule": "resident" if sinkhorn_resident else "streaming", "class_ms_per_step": round(ms, 3), "launches_per_step": sk_launches,
          "stage_brackets": launches.get("sinkhorn", 1)}
    if ms > 0:
        sk["survey_equivalent_gbs"] = round(counts_per_step["sinkhorn_bytes_survey"] / (ms * 1e-3) / 1e9, 1)
        one_sweep = counts_per_step["sinkhorn_bytes"] / (ms * 1e-3) / 1e9
        if sinkhorn_resident:
            sk.update(kernel="sinkhorn_resident_kernel (iterations 2..iters in one launch, score matrices in registers + LDS) + first-iteration "
                             "sweep/combine, safety net (no-op), sinkhorn_scores", bound="valu", unit="fraction of VALU issue slots", peak=1.0)
            key = "sinkhorn_resident" if "sinkhorn_resident" in pmc else "sinkhorn"
            vf = (pmc.get(key) or {}).get("valu_issue_frac")
            sk["achieved"] = vf
            sk["frac"] = vf
            if vf is not None:
                sk["pmc"] = dict({kk: vv for kk, vv in pmc[key].items()}, **src_pmc)
            tb = (tj.get("sinkhorn_resident") or {}).get("hbm_bytes_per_launch")
            sk["traffic"] = tb
            if tb:
                sk["hbm_counter_gbs"] = round(tb / (ms * 1e-3) / 1e9, 1)
                sk["hbm_frac_of_peak"] = round(tb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                sk["traffic_source"] = src_tr
            sk["streaming_one_sweep_equivalent_gbs"] = round(one_sweep, 1)
        else:
            sk.update(kernel="sinkhorn_sweep(_fast) + sinkhorn_combine(_fast) per iteration, then sinkhorn_scores; one stage bracket incl. launch gaps; "
                             "algorithmic bytes = ONE read of S per iteration + column partials + scores write",
                      bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS, achieved=round(one_sweep, 1), frac=round(one_sweep / PEAK_HBM_GBS, 4),
                      algorithmic_gb_per_step=round(counts_per_step["sinkhorn_bytes"] / 1e9, 3))
            if "sinkhorn_sweep" in tj and "sinkhorn_combine" in tj:
                sk["traffic"] = (tj["sinkhorn_sweep"]["hbm_bytes_per_launch"] + tj["sinkhorn_combine"]["hbm_bytes_per_launch"]) * num_iters
                sk["traffic_source"] = src_tr
            else:
                sk["traffic"] = None
    roofs["sinkhorn"] = sk
    dominant = max(per_step, key=lambda k: cls_ms[k])
    roof = roofs.pop(dominant)
    roof["class"] = dominant
    return roof, roofs


# ----------------------------------------------------------------------------------------------- CPU baseline
def _physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def _cpu_worker(cfg_kw, m, n, threads, seconds, start_at, q):
    """One process of the throughput baseline: B=1 pairs of the workload back to back for `seconds` seconds."""
    import torch as th
    th.set_num_threads(threads)
    from oracle import superglue_oracle as orc
    cfg = syn.make_config(**cfg_kw)
    sd = syn.make_state_dict(cfg, seed=0)
    data = syn.make_batch(1, m, n, cfg_kw["descriptor_dim"], cfg_kw["side_info_size"], seed=0)
    with th.no_grad():
        orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)            # warm-up
        while time.time() < start_at:
            time.sleep(0.01)
        t0 = time.time(); done = 0
        while time.time() - t0 < seconds:
            orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
            done += 1
        q.put((done, time.time() - t0))


def cpu_baseline(cfg, sd, cfg_kw, m, n, budget_s=24.0):
    """The CPU oracle (a torch-CPU port of the reference algorithm, oracle/superglue_oracle.py: `kind: port`; the GPU box
    has no /root/reference) on this box's host cores, B=1 pairs of the same workload, bounded to ~budget_s seconds:
    (a) latency: one process, thread count picked by a short probe (torch's intra-op pool oversubscribes badly on
    256-thread hosts); (b) throughput: k processes x t threads covering the physical cores.  `value` is the better of
    the two pairs/s figures, `cores` the threads it used."""
    import multiprocessing as mp
    from oracle import superglue_oracle as orc
    ncpu, nphys = os.cpu_count() or 1, _physical_cores()
    data = syn.make_batch(1, m, n, cfg_kw["descriptor_dim"], cfg_kw["side_info_size"], seed=0)
    probe_kw = dict(cfg_kw, num_stages=1, num_iters=4)
    pcfg = syn.make_config(**probe_kw)
    best_t, best_dt = 1, float("inf")
    with torch.no_grad():
        for t in [c for c in (4, 8, 16, 32, 64) if c <= ncpu] or [1]:
            torch.set_num_threads(t)
            orc.superglue_forward(sd, pcfg, data)
            t0 = time.perf_counter(); orc.superglue_forward(sd, pcfg, data); dt = time.perf_counter() - t0
            if dt < best_dt:
                best_t, best_dt = t, dt
            if dt > 5.0:
                break
        torch.set_num_threads(best_t)
        t0 = time.perf_counter(); orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD); warm = time.perf_counter() - t0
        reps = max(1, min(3, int(0.3 * budget_s / max(warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
        dt = (time.perf_counter() - t0) / reps
    single = 1.0 / dt
    out = {"value": round(single, 4), "unit": "image-pairs/s", "cores": best_t, "kind": "port",
           "host_threads": ncpu, "physical_cores": nphys,
           "single_process": {"pairs_per_s": round(single, 4), "threads": best_t, "ms_per_pair": round(dt * 1e3, 1)},
           "sample": f"{reps} x 1 pair of the same workload (B=1, torch-CPU oracle, {best_t} of {ncpu} host threads), {dt * 1e3:.0f} ms/pair"}
    # (b) throughput: k processes x t threads, pinned by nothing but the OS scheduler
    t_per = max(1, min(best_t, 16))
    k = max(1, min(16, nphys // t_per))
    window = max(4.0, min(12.0, 0.5 * budget_s))
    if k > 1 and dt * 2 < window:
        try:
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            start_at = time.time() + 10.0 + 2.0 * dt       # import torch + weights + one warm-up pair in every worker
            procs = [ctx.Process(target=_cpu_worker, args=(cfg_kw, m, n, t_per, window, start_at, q)) for _ in range(k)]
            for p_ in procs:
                p_.start()
            res = [q.get(timeout=start_at - time.time() + window + 60.0) for _ in procs]
            for p_ in procs:
                p_.join(timeout=30)
            thr = sum(d / el for d, el in res)
            if thr > single:      # reported only when it beats the single process (a losing leg says nothing about the host)
                out["multi_process"] = {"pairs_per_s": round(thr, 4), "processes": k, "threads_each": t_per, "window_s": window,
                                        "pairs_done": sum(d for d, _ in res)}
                out.update(value=round(thr, 4), cores=k * t_per,
                           sample=f"{k} processes x {t_per} threads ({k * t_per} of {nphys} physical cores), B=1 pairs of the same "
                                  f"workload back to back for {window:.0f} s: {sum(d for d, _ in res)} pairs; single process "
                                  f"{best_t} threads: {dt * 1e3:.0f} ms/pair")
        except Exception:            # the baseline is informational: never fail the bench line over it
            pass
    return out


# ----------------------------------------------------------------------------------------------- helpers
def _stage_dict(ms, cnt):
    return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.OG_STAGES)}


def _median_stages(run_profiled, n=3):
    profs = [run_profiled() for _ in range(n)]
    stages = {k: sorted(p[k][0] for p in profs)[n // 2] for k in profs[0]}
    launches = {k: profs[0][k][1] for k in profs[0]}
    return stages, launches


def _timed_steps(step, args, dist_on, dev):
    for _ in range(args.warmup):
        step()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
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
    return dt, out


def bench_ragged(args, world, rank, dev, dist_on=False):
    """BASELINE configs[4]: ragged pairs with 512-2048 keypoints per image, 16 per GPU, cost-balanced over the ranks, through
    the token-packed ragged path (SuperGlue.match_ragged_packed -> og_forward_ragged)."""
    kw = dict(syn.CONFIGS["C2"]); kw.pop("kpts"); kw.pop("batch")
    per_gpu = args.batch or 16
    total = per_gpu * world
    lens = syn.ragged_lengths(total, 512, 2048, seed=0)
    costs = [sharding.pair_cost(m, n) for m, n in lens]
    shards = sharding.shard_pairs(total, world, costs)
    mine = shards[rank]
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=0)
    model = SuperGlue(cfg).eval(); model.load_state_dict(sd, strict=True); model.to(dev)
    pairs = []
    for i in mine:
        p = syn.make_pair(lens[i][0], lens[i][1], 256, 1, seed=i)
        p = {k: v.to(dev) for k, v in p.items()}
        p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
        pairs.append(p)
    packed = model.pack_ragged(pairs)          # inputs resident in HBM in the token-packed layout the boundary takes

    def match_mine(_ids):
        return model.match_ragged_packed(packed, MATCH_THRESHOLD, both_sides=False)

    def step():
        if dist_on:      # the one collective: match lists to rank 0, padded to the longest keypoint set of the job
            return sharding.match_sharded_ragged(match_mine, lens, costs, dst=0, always_collective=True, device=dev)
        return match_mine(mine)

    dt, _ = _timed_steps(step, args, dist_on, dev)
    if rank == 0:
        res = match_mine(mine)
        my_lens = [lens[i] for i in mine]
        counts = _sum_counts(kw, my_lens)

        def run_profiled():
            ms = (C.c_float * len(_lib.OG_STAGES))(); cnt = (C.c_int32 * len(_lib.OG_STAGES))()
            model.match_ragged_packed(packed, MATCH_THRESHOLD, both_sides=False, _profile=(ms, cnt))
            return _stage_dict(ms, cnt)
        stages, launches = _median_stages(run_profiled)
        roof, roof2 = roofline_block(counts, stages, launches, kw["num_iters"], False, False)      # ragged batches always stream
        line = {"metric": "image-pairs/sec (C5 ragged 512-2048 kpts)", "value": round(total * args.steps / dt, 2), "unit": "image-pairs/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4]: {total} ragged pairs ({per_gpu}/GPU), 512-2048 kpts/image, 256-dim, 9 stages, "
                                       "100 Sinkhorn iters, token-packed ragged kernels (og_forward_ragged), LPT cost-balanced over ranks",
                           "mean_kpts": round(sum(a + b for a, b in lens) / (2 * total), 1), "pairs_per_gpu": per_gpu},
                "roofline": roof, "roofline_other": roof2, "stages_ms": {k: round(v, 3) for k, v in stages.items()},
                "algorithmic": {"gflop_per_step_rank0": round(counts["total_flops"] / 1e9, 2),
                                "sinkhorn_gb_per_step_rank0": round(counts["sinkhorn_bytes"] / 1e9, 3)},
                "valid_matches_per_pair": round(sum(int((r["matches0"] >= 0).sum()) for r in res) / max(1, len(res)), 1)}
        if world == 1 and not args.no_cpu_baseline:   # the mean-size pair of the job as the CPU sample
            mm = int(round(sum(a for a, _ in lens) / total)); nn_ = int(round(sum(b for _, b in lens) / total))
            line["cpu_baseline"] = cpu_baseline(cfg, sd, kw, mm, nn_)
            line["cpu_baseline"]["sample"] += f" [one {mm}x{nn_}-keypoint pair = the mean size of the ragged job]"
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _respawn(args):
    """`python bench.py --gpus N` without a torch.distributed environment: one process per GPU under torch.distributed.run."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def _dry_run(args, world, rank):
    """OG_BENCH_DRYRUN=1 (tests, no GPU): the launch / rendezvous / gather plumbing of the N-rank bench under gloo with
    the matcher replaced by a stub.  Prints a line marked "dry_run": true -- never a measurement."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = syn.ragged_lengths(3 * world, 8, 32, seed=0)

    def stub(ids):
        return [{"matches0": torch.full((lens[i][0],), i, dtype=torch.int64), "matching_scores0": torch.full((lens[i][0],), 0.5)}
                for i in ids]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = sharding.match_sharded_ragged(stub, lens, always_collective=world > 1)
    dt = time.perf_counter() - t0
    if rank == 0:
        ok = all(bool((got["matches0"][i, :lens[i][0]] == i).all()) for i in range(len(lens)))
        print(json.dumps({"metric": "dry run (no GPU, gloo, stub matcher)", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "gather_ok": ok, "value": round(len(lens) * args.steps / dt, 1), "unit": "stub-pairs/s"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(syn.CONFIGS) + ["C5"])
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn(args)                                   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if os.environ.get("OG_BENCH_DRYRUN") == "1":
        return _dry_run(args, world, rank)
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
            sharding.gather_matches(out, pair_ids, world * B, dst=0, always_collective=True, cap=B)
        return out

    dt, out = _timed_steps(step, args, dist_on, dev)
    model.check_status()          # outside the timed region: the resident Sinkhorn kernel of the last step completed (no time-out)

    if rank == 0:
        c1 = algorithmic_counts(kw, m, n)
        counts = {k: v * B for k, v in c1.items()}
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt

        def run_profiled():     # per-kernel-class time of one step, HIP events on the launch stream
            ms = (C.c_float * len(_lib.OG_STAGES))(); cnt = (C.c_int32 * len(_lib.OG_STAGES))()
            model.match(data, MATCH_THRESHOLD, both_sides=True, _profile=(ms, cnt))
            return _stage_dict(ms, cnt)
        stages, launches = _median_stages(run_profiled)
        resident = bool(_lib.load().og_sinkhorn_schedule(B, m, n, kw["num_iters"]))
        std_batch = B == syn.CONFIGS[args.config]["batch"]
        roof, roof2 = roofline_block(counts, stages, launches, kw["num_iters"], args.config == "C2" and std_batch, resident,
                                     TRAFFIC_BY_CONFIG.get(args.config) if std_batch else None)
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
            "algorithmic": {"gflop_per_pair": round(c1["total_flops"] / 1e9, 2), "sinkhorn_gb_per_pair": round(c1["sinkhorn_bytes"] / 1e9, 3),
                            "sinkhorn_gb_per_pair_survey_two_sweeps": round(c1["sinkhorn_bytes_survey"] / 1e9, 3)},
            "valid_matches_per_pair": round(float((out["matches0"] >= 0).sum().item()) / B, 1),
        }
        # The same step fed from pinned HOST buffers, H2D of keypoints/descriptors/side-info included and double-buffered
        # on a copy stream (batch k+1 uploads while batch k computes).  The boundary hands over device tensors, so this is
        # informational and never `value`.
        host = {k: v.cpu().pin_memory() for k, v in data.items() if torch.is_tensor(v)}
        copy_stream = torch.cuda.Stream(device=dev)
        bufs = [{k: torch.empty_like(v, device=dev) for k, v in host.items()} for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]

        def upload(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[i])            # the step that read this buffer has finished
                for k, v in host.items():
                    bufs[i][k].copy_(v, non_blocking=True)
                ready[i].record(copy_stream)
        for i in range(2):
            freed[i].record(torch.cuda.current_stream(dev))
        upload(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(args.steps):
            i = s_ & 1
            if s_ + 1 < args.steps:
                upload(i ^ 1)
            torch.cuda.current_stream(dev).wait_event(ready[i])
            d = dict(bufs[i]); d["image0_size"], d["image1_size"] = data["image0_size"], data["image1_size"]
            model.match(d, MATCH_THRESHOLD, both_sides=True)
            freed[i].record(torch.cuda.current_stream(dev))
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

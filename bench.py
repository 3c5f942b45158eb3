#!/usr/bin/env python3
"""Benchmark of the SuperGlue hot path on MI355X: matched image-pairs / second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C1|C3|C4|C5] [--batch PAIRS_PER_GPU | --global-batch PAIRS]
                    [--no-cpu-baseline]

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
def _newest(*names):
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", names[-1])


PMC_SUMMARY = _newest("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json")
TRAFFIC_SUMMARY = _newest("r06_traffic_c2.json", "r05_traffic_c2.json", "r04_traffic_c2.json", "r03_traffic_c2.json")
TRAFFIC_BY_CONFIG = {"C3": _newest("r06_traffic_c3.json", "r05_traffic_c3.json", "r04_traffic_c3.json", "r03_traffic_c3.json"),
                     "C4": _newest("r06_traffic_c4.json", "r05_traffic_c4.json", "r04_traffic_c4.json", "r03_traffic_c4.json")}
# oracle (the port bench.py times on the GPU box) vs the unmodified reference, timed side by side in the BUILD container where
# /root/reference exists (scripts/port_vs_reference.py); pasted into cpu_baseline with its source
PORT_VS_REFERENCE = _newest("r05_port_vs_reference.json", "r04_port_vs_reference.json")
ARITHMETIC = ("f32 inputs / outputs; GNN 1x1 convs, attention QK^T and PV, final projection and score matrix as split-f16 x3 MFMA "
              "(x = hi + lo binary16, Ah.Bh + Ah.Bl + Al.Bh into one fp32 accumulator: fp32-class accuracy); keypoint-encoder MLP exact "
              "fp32 MFMA; softmax, Sinkhorn and match extraction fp32 VALU")


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
    # ... and a RESIDENT schedule: S read by the first-iteration sweep and once more into registers + LDS, column partials of the first
    # iteration, the final read for the scores kernel and the scores write
    sink_res = 4.0 * (3.0 * m * n + 2 * rb * n + (m + 1) * (n + 1))
    return {"gemm_f32_flops": enc, "gemm_f16x3_flops": proj + final + score, "mlp_flops": mlp, "attention_flops": attn,
            "total_flops": enc + proj + attn + final + score, "sinkhorn_bytes": sink_one, "sinkhorn_bytes_survey": sink_survey,
            "sinkhorn_bytes_resident": sink_res}


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


def roofline_block(counts_per_step, stages, launches, num_iters, measured_on_this_workload, sinkhorn_launches, traffic_path=None):
    """One roofline object PER KERNEL (the profiled step brackets every launch of the MFMA kernels separately: og_forward_profiled).
    achieved = algorithmic work per step / the kernel's time per step (HIP events on the launch stream, median of 3 profiled steps)
    = algorithmic work per launch / average launch duration.  Returns (the kernel with the largest time share, the others).
    sinkhorn_launches: resident-kernel launches the library schedules for this shape (og_sinkhorn_schedule; 0 = streaming)."""
    pmc = _load_json(PMC_SUMMARY) if measured_on_this_workload else {}
    tpath = TRAFFIC_SUMMARY if measured_on_this_workload else traffic_path      # counter summaries exist for C2 (PMC + traffic) and C3 / C4 (traffic)
    tj = _load_json(tpath) if tpath else {}
    src_pmc = {"source": os.path.relpath(PMC_SUMMARY, ROOT), "collected_at_commit": pmc.get("_commit"), "measured_in_this_run": False}
    src_tr = {"source": os.path.relpath(tpath, ROOT) if tpath else None, "collected_at_commit": tj.get("_commit"), "measured_in_this_run": False}
    mlp_ms = stages.get("mlp_fused", 0.0)
    # kernel -> (profiler stage, algorithmic work per step, peak, PMC / traffic key, description)
    kernels = {
        "attention_dma_kernel": ("attention", counts_per_step["attention_flops"], PEAK_F16_MFMA_TFLOPS, "attention",
                                 "split-f16 flash attention (dh = 64, 32: K/V tiles by LDS-DMA; dh = 16: attention_kernel; one to four pairs: the key-split forms of the "
                                 "same kernel); executes 3x the algorithmic flops"),
        "gemm_nt_f16x3 (stand-alone)": ("gemm_f16x3", counts_per_step["gemm_f16x3_flops"] - (counts_per_step["mlp_flops"] if mlp_ms > 0 else 0.0),
                                        PEAK_F16_MFMA_TFLOPS, "gemm_f16x3_standalone",
                                        "gemm_nt_f16x3_big2_kernel (256x256 tiles: q/k/v projections) / gemm_nt_f16x3_kernel (128-token tiles: last encoder "
                                        "conv, final projection; the message MLP when it is not fused) / gemm_nt_f16x3_big_kernel (batched score matrix) / "
                                        "proj_small_kernel (q/k/v projections of launches of <= 8192 rows) / proj_stream_kernel (q/k/v projections of 128-d batches: x fragments in "
                                        "registers, weights through an LDS ring); "
                                        "split-f16 3-pass MFMA: executes 3x the algorithmic flops"),
        "gemm_nt_f32_kernel": ("gemm_f32", counts_per_step["gemm_f32_flops"], PEAK_F32_MFMA_TFLOPS, "gemm_f32",
                               "keypoint-encoder MLP without its last conv; exact fp32 MFMA"),
    }
    if mlp_ms > 0:
        kernels["mlp_fused_kernel"] = ("mlp_fused", counts_per_step["mlp_flops"], PEAK_F16_MFMA_TFLOPS, "mlp_fused",
                                       "message MLP of a GNN layer (fc.0 -> ReLU -> fc.3 + residual) in one launch, hidden activation in registers "
                                       "(launches of <= 8192 rows: mlp_small_kernel, 32-token workgroups); "
                                       "split-f16 3-pass MFMA: executes 3x the algorithmic flops")
    roofs, ms_of = {}, {}
    for name, (stage, work, peak, key, desc) in kernels.items():
        ms = stages[stage]
        nl = max(1, launches[stage])
        ach = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r = {"kernel": name, "what": desc, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
             "traffic": None, "ms_per_step": round(ms, 3), "launches_per_step": launches[stage], "avg_launch_ms": round(ms / nl, 4),
             "algorithmic_work_per_launch_tflop": round(work / nl / 1e12, 6)}
        if peak == PEAK_F16_MFMA_TFLOPS:
            # the split-f16 kernels execute 3 MFMA passes per algorithmic product: their ceiling is a third of the pipe's rate
            r["executed_tflops"] = round(3.0 * ach, 1)
            r["sustained_peak_measured"] = SUSTAINED_F16_MFMA_TFLOPS
            r["executed_frac_of_sustained"] = round(3.0 * ach / SUSTAINED_F16_MFMA_TFLOPS, 4)
        if key in tj and isinstance(tj[key], dict) and "hbm_bytes_per_launch" in tj[key]:
            r["traffic"] = tj[key]["hbm_bytes_per_launch"]
            r["traffic_source"] = src_tr
        if key in pmc:       # SQ counter summary of the same kernel (separate rocprofv3 --pmc passes)
            r["mfma_busy_frac"] = pmc[key].get("mfma_busy_frac")
            r["pmc"] = dict({kk: vv for kk, vv in pmc[key].items() if kk != "mfma_busy_frac"}, **src_pmc)
        roofs[name] = r
        ms_of[name] = ms
    # ---- Sinkhorn (one profiler bracket around the whole stage).  Streaming schedule: HBM roofline on what it has to move (one read of
    #      S per iteration + column partials + scores).  Resident schedule: the plan matrices stay on chip (registers + LDS) for all
    #      iterations after the first; HBM sees S once per launch plus the scores, so the HBM fraction says little -- reported are the
    #      SURVEY 8(d) bytes / time (may exceed the HBM peak: that is the point of residency), the counter bytes / time, and the
    #      iteration rate, which is bound by the cross-workgroup exchange latency (profiles/r04_e_sinkhorn_lazy_trace.log).
    ms = stages["sinkhorn"]
    resident = sinkhorn_launches > 0
    sk_launches = 4 + sinkhorn_launches if resident else 2 * num_iters + 1   # first-iteration sweep + combine, resident launches, safety net, scores
    sk = {"kernel": "sinkhorn", "schedule": "resident" if resident else "streaming", "ms_per_step": round(ms, 3), "launches_per_step": sk_launches,
          "stage_brackets": launches.get("sinkhorn", 1)}
    if ms > 0:
        survey_gbs = round(counts_per_step["sinkhorn_bytes_survey"] / (ms * 1e-3) / 1e9, 1)
        one_sweep = counts_per_step["sinkhorn_bytes"] / (ms * 1e-3) / 1e9
        if resident:
            # a fraction above 1 is not a roofline fraction.  The resident schedule reads S twice (first-iteration sweep, then the
            # load into registers + LDS) and writes the scores once -- THAT is what it has to move, and `frac` prices those bytes against the HBM
            # peak: a small number, because the stage is not bound by HBM at all.  What bounds it is the LATENCY of an iteration (two L2 hops of
            # the column-sum exchange + five workgroup barriers: ~12.7k cycles for ~5k cycles of arithmetic at C2, profiles/r04_e_sinkhorn_lazy_trace.log):
            # reported as `latency_model`.  The SURVEY 8(d) figure (two sweeps of the augmented matrix per iteration / stage time) stays as a side
            # figure without a fraction.
            must_move = counts_per_step["sinkhorn_bytes_resident"]
            ach = must_move / (ms * 1e-3) / 1e9
            iters_resident = max(1, num_iters - 1)
            us_it = ms * 1e3 / iters_resident / max(1, sinkhorn_launches)
            sk.update(what=f"sinkhorn_resident_kernel x {sinkhorn_launches} (iterations 2..iters of a round of co-resident pairs in one launch, plan entries in "
                           "registers + LDS, column sums exchanged through {epoch, value} granules) + first-iteration sweep/combine, safety net (no-op), "
                           "sinkhorn_scores", bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS, achieved=round(ach, 1), frac=round(ach / PEAK_HBM_GBS, 4),
                      algorithmic_gb_per_step=round(must_move / 1e9, 3),
                      note="achieved = the bytes the RESIDENT schedule has to move (S read by the first-iteration sweep and once more into registers + LDS, "
                           "scores written once) / stage time; far below the HBM peak by construction -- the stage is latency-bound, see latency_model",
                      latency_model={"bound": "iteration latency of the cross-workgroup exchange (two L2 hops + five barriers per iteration), not bytes or flops",
                                     "iterations_on_chip": iters_resident, "launches": sinkhorn_launches,
                                     "us_per_iteration_incl_first_sweep_and_scores": round(us_it, 2),
                                     "cycles_per_iteration_at_2.4GHz": int(us_it * 2400),
                                     "arithmetic_cycles_per_iteration_traced": 5000, "source": "profiles/r04_e_sinkhorn_lazy_trace.log (phase trace at C2)"},
                      survey_equivalent_gbs=survey_gbs,
                      survey_equivalent_note="SURVEY 8(d) bytes (two sweeps of the augmented matrix per iteration, as the reference does them) / stage time: "
                                             "above the HBM peak because the resident schedule does not move them; a throughput-equivalent, not a roofline fraction")
            key = "sinkhorn_resident"
            if key in pmc:
                sk["pmc"] = dict({kk: vv for kk, vv in pmc[key].items()}, **src_pmc)
            tb = (tj.get("sinkhorn_resident") or {}).get("hbm_bytes_per_launch")
            sk["traffic"] = tb
            if tb:
                sk["hbm_counter_gbs"] = round(tb * sinkhorn_launches / (ms * 1e-3) / 1e9, 1)
                sk["traffic_source"] = src_tr
            sk["streaming_one_sweep_equivalent_gbs"] = round(one_sweep, 1)
        else:
            sk.update(what="sinkhorn_sweep(_fast) + sinkhorn_combine(_fast) per iteration, then sinkhorn_scores; one stage bracket incl. launch gaps; "
                           "algorithmic bytes = ONE read of S per iteration + column partials + scores write",
                      bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS, achieved=round(one_sweep, 1), frac=round(one_sweep / PEAK_HBM_GBS, 4),
                      algorithmic_gb_per_step=round(counts_per_step["sinkhorn_bytes"] / 1e9, 3), survey_equivalent_gbs=survey_gbs)
            if "sinkhorn_sweep" in tj and "sinkhorn_combine" in tj:
                sk["traffic"] = (tj["sinkhorn_sweep"]["hbm_bytes_per_launch"] + tj["sinkhorn_combine"]["hbm_bytes_per_launch"]) * num_iters
                sk["traffic_source"] = src_tr
            else:
                sk["traffic"] = None
    roofs["sinkhorn"] = sk
    ms_of["sinkhorn"] = ms
    total = sum(ms_of.values()) or 1.0
    for k in roofs:
        roofs[k]["time_share"] = round(ms_of[k] / total, 4)
    dominant = max(ms_of, key=lambda k: ms_of[k])          # the single kernel with the largest time share
    roof = roofs.pop(dominant)
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
    pvr = _load_json(PORT_VS_REFERENCE)
    if pvr:
        pvr = dict(pvr, note="ratio of the port's time to the unmodified reference's, timed interleaved in the build container; see each config's "
                             "port_over_reference_spread: not resolvable better than about +-30 % there (8 shared threads)")
    out = {"value": round(single, 4), "unit": "image-pairs/s", "cores": best_t, "kind": "port",
           "port_vs_reference": dict(pvr, source=os.path.relpath(PORT_VS_REFERENCE, ROOT), measured_in_this_run=False) if pvr else None,
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
            out["multi_process"] = {"pairs_per_s": round(thr, 4), "processes": k, "threads_each": t_per, "window_s": window,
                                    "pairs_done": sum(d for d, _ in res), "beats_single_process": bool(thr > single)}      # reported even when it loses
            if thr > single:      # `value` = the better of the two legs
                out.update(value=round(thr, 4), cores=k * t_per,
                           sample=f"{k} processes x {t_per} threads ({k * t_per} of {nphys} physical cores), B=1 pairs of the same "
                                  f"workload back to back for {window:.0f} s: {sum(d for d, _ in res)} pairs; single process "
                                  f"{best_t} threads: {dt * 1e3:.0f} ms/pair")
        except Exception as e:       # the baseline is informational: never fail the bench line over it -- but say that the leg did not run
            out["multi_process"] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
    else:
        out["multi_process"] = {"skipped": f"k = {k} processes, {dt:.1f} s per pair against a {window:.0f} s window: the leg would not finish inside the baseline's time budget"}
    return out


# ----------------------------------------------------------------------------------------------- helpers
def _stage_dict(ms, cnt):
    return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.OG_STAGES)}


def _median_stages(run_profiled, n=3):
    profs = [run_profiled() for _ in range(n)]
    stages = {k: sorted(p[k][0] for p in profs)[n // 2] for k in profs[0]}
    launches = {k: profs[0][k][1] for k in profs[0]}
    return stages, launches


def _flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass


def _sync(dev):
    if dev.type == "cuda":            # (the dry run drives the same control flow on the CPU under gloo)
        torch.cuda.synchronize()


def _timed_steps(step, args, dist_on, dev):
    for _ in range(args.warmup):
        step()
    if dist_on:
        torch.distributed.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step()
    _sync(dev)
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def _step_spread(step, steps, dev):
    """min / median / max of the per-step times of `steps` further steps: one HIP event pair per step on the launch stream (torch's
    current stream = the stream every kernel of the step is enqueued on), read after ONE synchronisation at the end -- the steps run
    back to back exactly like the timed region that defines `value`."""
    if dev.type != "cuda":            # dry run: host clock
        ms = []
        for _ in range(steps):
            t0 = time.perf_counter(); step(); ms.append((time.perf_counter() - t0) * 1e3)
        ms.sort()
        return {"min": round(ms[0], 3), "median": round(ms[len(ms) // 2], 3), "max": round(ms[-1], 3), "steps": steps, "how": "host clock (dry run)"}
    st = torch.cuda.current_stream(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record(st)
    for i in range(steps):
        step()
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return {"min": round(ms[0], 3), "median": round(ms[len(ms) // 2], 3), "max": round(ms[-1], 3), "steps": steps,
            "how": "HIP event pairs on the launch stream around every step of a second back-to-back run"}


def _rccl_block(dist_on, step_compute=None, step_full=None, dev=None, reps=10):
    """What a SCALE line needs to explain itself (no scaling curve is asked of this script): every rank's own step time with and without
    the collective, measured after the timed region (HIP events on the launch stream, `reps` steps each), gathered to rank 0."""
    if not dist_on:
        return {"backend": None, "world_size": 1, "note": "single process: no process group, no collective"}
    import torch.distributed as dist
    blk = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective": "one gather of the match lists to rank 0 per step"}
    if step_compute is None or step_full is None:
        return blk

    def timed(fn):
        if dev.type != "cuda":        # dry run: host clock
            fn(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) / reps * 1e3
        st = torch.cuda.current_stream(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(); dist.barrier()
        a.record(st)
        for _ in range(reps):
            fn()
        b.record(st); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    t_full, t_comp = timed(step_full), timed(step_compute)
    mine = torch.tensor([t_comp, t_full], device=dev, dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    blk.update(per_rank_ms_compute_only=[round(float(t[0]), 3) for t in allr], per_rank_ms_with_gather=[round(float(t[1]), 3) for t in allr],
               gather_ms_rank0=round(float(allr[0][1] - allr[0][0]), 3),
               how=f"{reps} steps each after the timed region, HIP events on the launch stream; gather = (step with the collective) - (step without) on rank 0")
    return blk


def _training_step(dev, B=4, N=1024, iters=20, steps=5):
    """Informational (never `value`): one TRAINING step of the C2 model on HIP kernels -- `SuperGlue(config).train()(data)`, the reference's NLL
    (utils/losses.py:7-53, margin None, written with masks: no data-dependent shapes), `loss.backward()` into every parameter -- at the
    4 pairs x 1024 keypoints the round reviews quote it on (`scripts/bench_train_step.py` is the stand-alone form).  SURVEY 8 row f2."""
    try:
        from openglue_amd.superglue import SuperGlue
        cfg = syn.make_config(descriptor_dim=256, num_stages=9, num_heads=4, num_iters=iters)
        model = SuperGlue(cfg); model.load_state_dict(syn.make_state_dict(cfg, seed=0)); model = model.to(dev).train()
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.make_batch(B, N, N, 256, 1, seed=1).items()}
        g = torch.Generator().manual_seed(0)
        gt0 = torch.full((B, N), -1, dtype=torch.long); gt1 = torch.full((B, N), -1, dtype=torch.long)
        for b in range(B):
            i = torch.randperm(N, generator=g)[: N // 2]; j = torch.randperm(N, generator=g)[: N // 2]
            gt0[b, i] = j; gt1[b, j] = i
        gt0, gt1 = gt0.to(dev), gt1.to(dev)
        m0, u0, u1 = (gt0 >= 0).float(), (gt0 == -1).float(), (gt1 == -1).float()
        per = lambda val, mask: (-(val * mask).sum(1) / mask.sum(1).clamp_min(1)).sum()

        def step():
            model.zero_grad(set_to_none=True)
            scores = model(data)["scores"]
            picked = scores[:, :-1, :-1].gather(2, gt0.clamp_min(0)[:, :, None])[:, :, 0]
            loss = (per(picked, m0) + 0.5 * (per(scores[:, :-1, -1], u0) + per(scores[:, -1, :-1], u1))) / B
            loss.backward()
            return loss
        step(); step(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        grads = sum(1 for p in model.parameters() if p.grad is not None)
        out = {"ms_per_step": round(ms, 2), "pairs_per_s": round(B / ms * 1e3, 1), "steps": steps, "loss": round(float(loss.item()), 4),
               "parameters_with_grad": grads, "workload": f"{B} pairs x {N}x{N} kpts, 256-dim, 9 stages, 4 heads, {iters} Sinkhorn iters, forward (train mode) + NLL + backward, "
                                                          "exact-fp32 GEMMs, no optimizer step (the caller's)", "dtype": "f32"}
        del model, data
        torch.cuda.empty_cache()
        return out
    except Exception as e:       # informational leg: never takes the bench line down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def _measure(step, step_compute, args, dist_on, dev):
    """The part of a bench run EVERY rank executes, in this order: the timed region, the per-step spread, the per-rank times of the rccl block.
    `step` holds the collective when N > 1, so nothing here may run on rank 0 alone (round 4 computed the spread inside `if rank == 0`: with more
    than one rank it would have waited in the gather for ever).  The gloo dry run (tests/test_sharding_cpu.py, 2 and 8 ranks) drives this very
    function with a stub matcher."""
    dt, out = _timed_steps(step, args, dist_on, dev)
    spread = _step_spread(step, args.steps, dev)
    rccl = _rccl_block(dist_on, step_compute, step, dev)
    return dt, out, spread, rccl


def bench_ragged(args, world, rank, dev, dist_on=False):
    """BASELINE configs[4]: ragged pairs with 512-2048 keypoints per image, 16 per GPU, cost-balanced over the ranks, through
    the token-packed ragged path (SuperGlue.match_ragged_packed -> og_forward_ragged)."""
    kw = dict(syn.CONFIGS["C2"]); kw.pop("kpts"); kw.pop("batch")
    per_gpu = args.batch or (args.global_batch // world if args.global_batch else 16)
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

    dt, _, spread, rccl = _measure(step, lambda: match_mine(mine), args, dist_on, dev)
    if rank == 0:
        res = match_mine(mine)
        my_lens = [lens[i] for i in mine]
        counts = _sum_counts(kw, my_lens)

        def run_profiled():
            ms = (C.c_float * len(_lib.OG_STAGES))(); cnt = (C.c_int32 * len(_lib.OG_STAGES))()
            model.match_ragged_packed(packed, MATCH_THRESHOLD, both_sides=False, _profile=(ms, cnt))
            return _stage_dict(ms, cnt)
        stages, launches = _median_stages(run_profiled)
        a0 = (C.c_int32 * len(my_lens))(*[a for a, _ in my_lens]); a1 = (C.c_int32 * len(my_lens))(*[b for _, b in my_lens])
        sk_launches = int(_lib.load().og_sinkhorn_schedule_ragged(len(my_lens), a0, a1, kw["num_iters"])) if len(my_lens) <= 64 else 0
        roof, roof2 = roofline_block(counts, stages, launches, kw["num_iters"], False, sk_launches)
        line = {"metric": "image-pairs/sec (C5 ragged 512-2048 kpts)", "value": round(total * args.steps / dt, 2), "unit": "image-pairs/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "step_ms_spread": spread,
                "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4]: {total} ragged pairs ({per_gpu}/GPU), 512-2048 kpts/image, 256-dim, 9 stages, "
                                       "100 Sinkhorn iters, token-packed ragged kernels (og_forward_ragged), LPT cost-balanced over ranks",
                           "arithmetic": ARITHMETIC, "global_batch": total,
                           "mean_kpts": round(sum(a + b for a, b in lens) / (2 * total), 1), "pairs_per_gpu": per_gpu},
                "rccl": rccl,
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
        _flush_c_stdio()              # RCCL's version banner sits in the C library's stdout buffer: get it out BEFORE the line, which stays the last one
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
    if args.global_batch and args.global_batch % world:
        raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the {world} ranks")
    lens = syn.ragged_lengths(args.global_batch or 3 * world, 8, 32, seed=0)      # --global-batch: the job's size is fixed, ranks split it

    def stub(ids):
        return [{"matches0": torch.full((lens[i][0],), i, dtype=torch.int64), "matching_scores0": torch.full((lens[i][0],), 0.5)}
                for i in ids]
    costs = [sharding.pair_cost(m, n) for m, n in lens]
    mine = sharding.shard_pairs(len(lens), world, costs)[rank]
    dev = torch.device("cpu")

    def step():
        return sharding.match_sharded_ragged(stub, lens, costs, always_collective=world > 1)
    # the SAME control flow as the real bench (timed region -> spread -> per-rank block, every rank; then rank 0 reports; then the barrier)
    dt, got, spread, rccl = _measure(step, lambda: stub(mine), args, world > 1, dev)
    if rank == 0:
        ok = all(bool((got["matches0"][i, :lens[i][0]] == i).all()) for i in range(len(lens)))
        print(json.dumps({"metric": "dry run (no GPU, gloo, stub matcher)", "dry_run": True, "n_gpus": world, "steps": args.steps, "global_batch": len(lens),
                          "scaling": "strong" if args.global_batch else "weak", "step_ms_spread": spread, "rccl": rccl,
                          "gather_ok": ok, "value": round(len(lens) * args.steps / dt, 1), "unit": "stub-pairs/s"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=sorted(syn.CONFIGS) + ["C5"])
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default: the config's)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="pairs of the whole job, split evenly over the ranks (BASELINE: C3 256, C4 64, C5 128 over 8 GPUs) -- strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training-step", action="store_true", help="skip the informational training-step leg (C2, N = 1 only)")
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
    elif args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the {world} ranks")
        B = args.global_batch // world
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

    dt, out, spread, rccl = _measure(step, lambda: model.match(data, MATCH_THRESHOLD, both_sides=True), args, dist_on, dev)
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
        sk_launches = int(_lib.load().og_sinkhorn_schedule(B, m, n, kw["num_iters"]))
        std_batch = B == syn.CONFIGS[args.config]["batch"]
        roof, roof2 = roofline_block(counts, stages, launches, kw["num_iters"], args.config == "C2" and std_batch, sk_launches,
                                     TRAFFIC_BY_CONFIG.get(args.config) if std_batch else None)
        line = {
            "metric": "image-pairs/sec (1024 kpts, 256-dim, 9 GNN layers)" if args.config == "C2" else f"image-pairs/sec ({args.config})",
            "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "step_ms_spread": spread,
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ({"S128": "the reference's shipped 128-d operating point (config/features/sift_opencv.yaml:2-4, config/config.yaml:53), not a BASELINE config: ",
                                     "S256": "the reference's shipped 256-d operating point (config/features/superpoint_magicleap.yaml:2-4, config/config.yaml:53), not a BASELINE config: ",
                                     "C4i20": "BASELINE configs[3] at the reference's default 20 Sinkhorn iterations (config/config.yaml:53; C4 runs 100): "}.get(
                                         args.config, f"BASELINE configs[{'1' if args.config == 'C2' else args.config}]: ")) + f"{m}x{n} kpts, {kw['descriptor_dim']}-dim, "
                                   f"{kw['num_stages']} self+cross stages, {kw['num_heads']} heads, {kw['num_iters']} Sinkhorn iters, "
                                   f"batch={B} pairs/GPU, random-init weights, seeded synthetic keypoints/descriptors",
                       "arithmetic": ARITHMETIC, "global_batch": world * B,
                       "pairs_per_gpu": B, "kpts": [m, n], "parallelism": f"pairs sharded over {world} GPU(s), 1 RCCL gather"},
            "rccl": rccl,
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
        # only in the driver's own invocation (C2 at its default batch, one GPU): sweeps over --batch / --config and profiler runs stay what they name
        want_train = world == 1 and args.config == "C2" and args.batch is None and args.global_batch is None and not args.no_training_step
        line["training_step"] = _training_step(dev) if want_train else None
        # ... and at the reference's OWN training shape: batch_size_per_gpu 2 (config/config.yaml:12) x up to 2048 keypoints, 20 iterations
        line["training_step_reference_shape"] = _training_step(dev, B=2, N=2048, iters=20, steps=5) if want_train else None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, kw, m, n)
        else:
            line["cpu_baseline"] = None
        _flush_c_stdio()              # RCCL's version banner sits in the C library's stdout buffer: get it out BEFORE the line, which stays the last one
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

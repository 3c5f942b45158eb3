#!/usr/bin/env python3
"""gpurun_out/pmc/{sq1,sq2,sq3,grbm}/p_counter_collection.csv (scripts/gpu_pmc.sh: separate rocprofv3 --kernel-trace --pmc
passes over scripts/traffic_driver.py = 2 hot-path steps at C2, B=32) -> profiles/r03_pmc_summary.json, the per-kernel-class SQ
counter summary bench.py attaches to its roofline objects.

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (= 32 x N_mfma for v_mfma_f32_32x32x16_f16);
SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE are summed over the 8 XCDs (32 shader engines for SQ).
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x effective clock x 256 CUs x 4 SIMDs)
with the effective clock = GRBM_GUI_ACTIVE / 8 / duration of the same kernel (its own pass), i.e. the fraction of the matrix-pipe
issue capacity the kernel used while it ran."""
import collections, csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_pmc_summary.json"
# gemm_f16x3 = the whole split-f16 GEMM class (stand-alone GEMM launches AND the fused message-MLP kernel, like bench.py's class);
# mlp_fused / gemm_f16x3_standalone = its two parts
CLASSES = [("gemm_f16x3", ("gemm_nt_f16x3", "mlp_fused_kernel", "proj_stream_kernel", "proj_small_kernel", "mlp_small_kernel")), ("mlp_fused", ("mlp_fused_kernel", "mlp_small_kernel")), ("gemm_f16x3_standalone", ("gemm_nt_f16x3", "proj_stream_kernel", "proj_small_kernel")),
           ("attention", ("attention",)), ("sinkhorn_resident", ("sinkhorn_resident_kernel",)), ("sinkhorn_sweep", ("sinkhorn_sweep",)),
           ("gemm_f32", ("gemm_nt_f32",))]
N_SIMD = 256 * 4

acc = {c: collections.defaultdict(float) for c, _ in CLASSES}     # counter -> sum over launches
dur = {c: collections.defaultdict(float) for c, _ in CLASSES}     # pass -> summed duration (ns)
cnt = {c: collections.defaultdict(int) for c, _ in CLASSES}
for path in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    tag = os.path.basename(os.path.dirname(path))
    seen = set()
    for r in csv.DictReader(open(path)):
        for c, key in CLASSES:
            if any(k_ in r["Kernel_Name"] for k_ in key):
                acc[c][r["Counter_Name"]] += float(r["Counter_Value"])
                did = (c, r["Dispatch_Id"])
                if did not in seen:
                    seen.add(did)
                    dur[c][tag] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    cnt[c][tag] += 1
res = {"_method": __doc__.strip().split("\n\n")[0].replace("\n", " "), "_units": "counter sums per launch (averaged over all launches of the class)",
       "_commit": os.environ.get("OG_COMMIT")}
for c, _ in CLASSES:
    if not cnt[c]:
        continue
    tags = sorted(cnt[c])
    n = cnt[c][tags[0]]
    d_ns = {t: dur[c][t] / max(1, cnt[c][t]) for t in tags}
    per = {k: v / max(1, n) for k, v in acc[c].items()}
    e = {"launches_profiled": n, "avg_launch_us": {t: round(v / 1e3, 1) for t, v in d_ns.items()}}
    g = lambda k: per.get(k)
    clk = None
    if g("GRBM_GUI_ACTIVE") and "grbm" in d_ns:
        clk = g("GRBM_GUI_ACTIVE") / 8.0 / d_ns["grbm"]            # GHz
        # GRBM_GUI_ACTIVE also counts the dispatch ramp and drain around a kernel: for launches under ~50 us the quotient comes out ABOVE the part's 2.4 GHz
        # maximum (round 5: 2.93 for gemm_f32, 2.73 for sinkhorn_sweep) -- not a clock.  Such classes get no clock from this file (the s_memtime / s_memrealtime
        # sampler, scripts/clock_under_load.py -> profiles/*clock_under_load.log, is the measurement for them) and their mfma_busy is priced at the 2.4 GHz
        # maximum, i.e. a LOWER bound of the busy fraction; the same for any quotient above 2.4.
        if d_ns["grbm"] < 50e3 or clk > 2.4:
            e["effective_clock_ghz"] = None
            e["effective_clock_note"] = f"GRBM_GUI_ACTIVE / duration = {clk:.2f} GHz for a {d_ns['grbm'] / 1e3:.0f} us launch: not a clock (ramp / drain counted); see the clock sampler log"
            clk = None
        else:
            e["effective_clock_ghz"] = round(clk, 3)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        t = next((d_ns[t] for t in tags if t.startswith("sq2")), None)
        use_clk = clk or 2.4
        if t:
            e["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (t * use_clk * N_SIMD), 4)
            e["mfma_busy_clock_ghz_used"] = round(use_clk, 3)
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        for k, name in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "wave_issuing_frac")):
            if g(k) is not None:
                e[name] = round(g(k) / wc, 4)
    if g("SQ_INSTS_VALU") is not None:
        # vector-ALU issue utilisation: a wave64 VALU instruction occupies its SIMD's issue slot for 4 cycles (16 lanes per cycle)
        t = next((d_ns[t] for t in tags if t.startswith("sq2")), None)
        if t:
            e["valu_issue_frac"] = round(g("SQ_INSTS_VALU") * 4.0 / (t * (clk or 2.4) * N_SIMD), 4)
    if g("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round((g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("SQ_INSTS_MFMA"):
        m = g("SQ_INSTS_MFMA")
        e["per_mfma"] = {k[9:].lower(): round(g(k) / m, 3) for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM") if g(k) is not None}
    e["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(per.items())}
    res[c] = e
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} if isinstance(v, dict) else v for k, v in res.items()}, indent=1))

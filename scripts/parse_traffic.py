#!/usr/bin/env python3
"""gpurun_out/traffic/{FETCH_SIZE,WRITE_SIZE}/p_counter_collection.csv (scripts/gpu_traffic.sh) -> the per-kernel-class HBM
traffic summary bench.py reads (profiles/r03_traffic_c2.json).  FETCH_SIZE counts 64 B per 128-B request on gfx950
(MI355X_MICROARCH.md): read bytes = 2 x FETCH_SIZE KB; WRITE_SIZE is 1:1 (calibrated on the 134 MB torch copy in the driver)."""
import collections, csv, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_traffic_c2.json"
CLASSES = [("gemm_f16x3", ("gemm_nt_f16x3", "mlp_fused_kernel", "proj_stream_kernel", "proj_small_kernel", "mlp_small_kernel")), ("mlp_fused", ("mlp_fused_kernel", "mlp_small_kernel")), ("gemm_f16x3_standalone", ("gemm_nt_f16x3", "proj_stream_kernel", "proj_small_kernel")),
           ("attention", ("attention",)), ("sinkhorn_resident", ("sinkhorn_resident_kernel",)), ("sinkhorn_sweep", ("sinkhorn_sweep",)),
           ("sinkhorn_combine", ("sinkhorn_combine",)), ("gemm_f32", ("gemm_nt_f32",))]
vals = {c: collections.defaultdict(list) for c, _ in CLASSES}
cal = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(root, counter, "p_counter_collection.csv"))):
        name = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        if "direct_copy" in name or "elementwise_kernel" in name and "copy" in name.lower():
            cal.setdefault(counter, []).append(v)
        for c, key in CLASSES:
            if any(k_ in name for k_ in key):
                vals[c][counter].append(v)
res = {"_commit": os.environ.get("OG_COMMIT"), "_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/traffic_driver.py "
                  "(2 C2 steps, B=32); counters are in KB; per MI355X_MICROARCH.md FETCH_SIZE counts 64 B per 128-B request on gfx950, "
                  "so read bytes = 2 x FETCH_SIZE; WRITE_SIZE 1:1.  Averages over all launches of a kernel class (scripts/parse_traffic.py)."}
for c, _ in CLASSES:
    f, w = vals[c]["FETCH_SIZE"], vals[c]["WRITE_SIZE"]
    if not f or not w:
        continue
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    res[c] = {"launches_profiled": len(f), "FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
              "hbm_bytes_per_launch": int(round((2 * fa + wa) * 1024))}
if cal:
    res["_calibration_copy_KB"] = {k: max(v) for k, v in cal.items()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

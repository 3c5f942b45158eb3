#!/bin/bash
# round 6, call i: where the stand-alone GEMM's LDS bank conflicts come from -- the SQ LDS counters of the regular build against a timing build without the
# main loop's fragment reads (-DOG_GEMM_ABL=16: results wrong, the epilogue's LDS transposes unchanged)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for tag in regular abl16; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ $tag = abl16 ] && lib=openglue_amd/lib/libog_gemm_abl16.so
  rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
  OPENGLUE_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_$tag/sq -o p -- python scripts/traffic_driver.py > gpurun_out/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
python - <<'PY' | tee gpurun_out/r06i_gemm_lds_conflicts.log
import csv, glob, collections
for tag in ("regular", "abl16"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for path in glob.glob(f"gpurun_out/pmc_{tag}/sq/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            cls = "gemm_big2" if "big2" in k else "attention" if "attention" in k else "mlp_fused" if "mlp_fused" in k else None
            if cls:
                acc[cls][r["Counter_Name"]] += float(r["Counter_Value"]); n[cls].add(r["Dispatch_Id"])
    for cls in sorted(acc):
        c = acc[cls]; L = len(n[cls])
        print(f"{tag:8s} {cls:10s} launches {L:4d}  per launch: LDS instructions {c['SQ_INSTS_LDS'] / L:12.0f}  LDS active cycles {c['SQ_LDS_IDX_ACTIVE'] / L:12.0f}  bank-conflict cycles {c['SQ_LDS_BANK_CONFLICT'] / L:12.0f}  "
              f"= {c['SQ_LDS_BANK_CONFLICT'] / max(1.0, c['SQ_LDS_IDX_ACTIVE']):.3f} of the active cycles")
PY

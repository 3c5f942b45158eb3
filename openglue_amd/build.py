"""Build libopenglue_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m openglue_amd.build [--force] [--debug]

The library has no torch dependency; it is loaded through ctypes (openglue_amd/_lib.py).
hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only container.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libopenglue_amd.so")
SOURCES = ["gemm_f32.hip", "gemm_f16x3.hip", "mlp_fused.hip", "attention.hip", "attention_train.hip", "linear_attention.hip", "sinkhorn.hip", "sinkhorn_resident.hip", "sinkhorn_train.hip", "batchnorm_train.hip", "matches.hip", "features.hip", "api.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


# On gfx950 a packed-fp32 VALU instruction (v_pk_add/mul/fma_f32) does not issue while the matrix pipe of its SIMD is busy: next to
# the MFMA stream of another wave every other VALU class still gets an issue slot every ~14 cycles, packed fp32 gets none
# (scripts/probes/mfma_valu_classes.hip, profiles/r04_probe_mfma_valu_classes.log).  In the attention kernel, whose two waves per SIMD
# alternate softmax (VALU) and MFMA phases, the compiler's packed rescale / row-sum arithmetic therefore serialised the waves: the
# device pass of that file is compiled without the feature (-3.5 % kernel time at C2; the Sinkhorn kernels, which have no MFMA and
# live on packed fp32, keep it).  The host pass does not know the feature and says so; harmless.
_NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PER_FILE_FLAGS = {"attention.hip": _NO_PACKED_FP32}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, debug: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "og_common.h"), os.path.join(os.path.dirname(HERE), "include", "openglue_amd.h")]
    flags = ["--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-O3", "-Wall", "-Wno-unused-function"]
    if debug:
        flags += ["-g", "-save-temps=obj"]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp, *headers, os.path.abspath(__file__)]):
            jobs.append([hipcc, *flags, *PER_FILE_FLAGS.get(src, []), "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)   # the host pass, see above
        if err.strip() and verbose:
            print(err, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH, *objs])
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--debug", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, debug=a.debug))

"""The attention kernel's opt-in forms (round 6; both stayed off by default after their A/B runs, profiles/r06_*): they ship in the library, so they stay under test.

* PIPE (OG_ATTN_PIPE=1, read once per process -> child process): the software-pipelined tile loop against a float64 softmax attention over the tile-count
  edge cases (1 .. 16 tiles, partial last tiles, dh 64 / 32, spikes that move the running max mid-way): same 4e-5 bound as the phase form.
* P16 (OG_ATTN_P16=1): the pipelined loop on 16x16x32 MFMAs (attention_p16_kernel).
* MX (OG_ATTN_MX_SV, per call): the P V cross products on the block-scaled e4m3 MFMA, with the 8-bit V rows made by torch exactly as a projection epilogue
  would write them: |O - float64| <= 5e-4 on |O| ~ 9 (the stage tolerance that gave 1e-4 on the log-scores in the emulation of round 5)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


def test_pipelined_tile_loop_matches_float64(gpu_device):
    out = _run("check_attention_pipe.py", {"OG_ATTN_PIPE": "1", "OG_CHECK_NO_TIMING": "1"})
    m = re.search(r"\[pipe=1\] worst error over the edge cases: ([0-9.e+-]+)", out)
    assert m, out[-1500:]
    assert float(m.group(1)) < 1e-4, out[-1500:]
    assert "nan" not in out.lower()


def test_pipelined_16x16x32_kernel_matches_float64(gpu_device):
    """attention_p16_kernel (OG_ATTN_P16=1, dh = 64 batch form): the pipelined loop re-tiled for v_mfma_f32_16x16x32_f16 -- two queries per lane, row statistics over
    four lanes, its own V swizzle.  Same edge cases, same bound (dh = 32 cases run the default kernel)."""
    out = _run("check_attention_pipe.py", {"OG_ATTN_P16": "1", "OG_CHECK_NO_TIMING": "1"})
    m = re.search(r"\+p16\] worst error over the edge cases: ([0-9.e+-]+)", out)
    assert m, out[-1500:]
    assert float(m.group(1)) < 1e-4, out[-1500:]
    assert "nan" not in out.lower()


def test_block_scaled_pv_cross_products_match_float64(gpu_device):
    out = _run("bench_attention_mx.py", {"MX_NO_TIMING": "1"})
    errs = [float(x) for x in re.findall(r"MX ([0-9.e+-]+)", out)]
    assert len(errs) >= 4, out[-1500:]
    assert max(errs) < 5e-4, out[-1500:]

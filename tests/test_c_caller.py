"""INTEGRATION.md section 3 made executable: examples/c_caller.c is compiled with gcc against include/openglue_amd.h and
libopenglue_amd.so and run (a) here, host entry points only, (b) on the GPU box, the whole og_forward -- and must
reproduce what the Python binding produces from the same bytes."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from openglue_amd import _lib, synthetic as syn
from openglue_amd.superglue import SuperGlue

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params_blob(model: SuperGlue) -> np.ndarray:
    """fp32 arrays in the order examples/c_caller.c reads them."""
    out = []
    f = lambda t: out.append(t.detach().cpu().float().numpy().ravel())
    enc = model.positional_encoding.encoder
    for i in range(len(model.hidden) + 1):
        conv = enc[3 * i]
        f(conv.weight); f(conv.bias)
        if i < len(model.hidden):
            bn = enc[3 * i + 2]
            f(bn.weight); f(bn.bias); f(bn.running_mean); f(bn.running_var)
    for holder in model.attention_gnn.layers:
        mod = holder.module
        for name in ("in_proj_q", "in_proj_k", "in_proj_v", "out_proj"):
            f(getattr(mod.mha, name).weight); f(getattr(mod.mha, name).bias)
        f(mod.fc[0].weight); f(mod.fc[0].bias)
        f(mod.fc[2].weight); f(mod.fc[2].bias); f(mod.fc[2].running_mean); f(mod.fc[2].running_var)
        f(mod.fc[3].weight); f(mod.fc[3].bias)
    f(model.linear_proj.weight); f(model.linear_proj.bias)
    if model.residual:
        f(model.mix_coefs)
    f(model.dustbin_score.reshape(1))
    return np.concatenate(out).astype(np.float32)


def _compile(tmp_path):
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    _lib.load()                                       # makes sure the library is built
    exe = str(tmp_path / "c_caller")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "examples", "c_caller.c"), "-L" + os.path.dirname(_lib.LIB_PATH), "-lopenglue_amd", "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH) + ":/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


def _write_case(tmp_path, B, m, n):
    cfg = syn.make_config(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=5, side_info_size=1)
    sd = syn.make_state_dict(cfg, seed=3)
    model = SuperGlue(cfg).eval()
    model.load_state_dict(sd, strict=True)
    shape = model._shape(B, m, n, 0.2)
    (tmp_path / "shape.bin").write_bytes(bytes(shape))
    (tmp_path / "params.bin").write_bytes(_params_blob(model).tobytes())
    data = syn.make_batch(B, m, n, 64, 1, seed=8)
    parts = [data[k].numpy().ravel() for k in ("keypoints0", "keypoints1", "local_descriptors0", "local_descriptors1", "side_info0", "side_info1")]
    parts.append(np.asarray(list(syn.IMAGE_WH) * 2, dtype=np.float32))
    (tmp_path / "inputs.bin").write_bytes(np.concatenate(parts).astype(np.float32).tobytes())
    return model, data


def test_c_caller_host_entry_points(tmp_path):
    exe = _compile(tmp_path)
    model, _ = _write_case(tmp_path, 2, 48, 56)
    r = subprocess.run([exe, str(tmp_path), "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"abi {_lib.OG_ABI_VERSION}," in r.stdout
    packed_c = np.frombuffer((tmp_path / "packed.bin").read_bytes(), dtype=np.float32)
    packed_py = model.pack_host()
    assert packed_c.shape == packed_py.shape
    assert np.array_equal(packed_c.view(np.uint32), packed_py.view(np.uint32))      # the same blob, bit for bit


@pytest.mark.gpu
def test_c_caller_forward_equals_python_binding(tmp_path, gpu_device):
    exe = _compile(tmp_path)
    B, m, n = 2, 48, 56
    model, data = _write_case(tmp_path, B, m, n)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    scores = np.frombuffer((tmp_path / "scores.bin").read_bytes(), dtype=np.float32).reshape(B, m + 1, n + 1)
    matches0 = np.frombuffer((tmp_path / "matches0.bin").read_bytes(), dtype=np.int64).reshape(B, m)
    model.to(gpu_device)
    out = model.match({k: (v.to(gpu_device) if torch.is_tensor(v) else v) for k, v in data.items()}, 0.2, both_sides=False)
    assert np.array_equal(scores, out["scores"].cpu().numpy())          # same library, same bytes in -> same bytes out
    assert np.array_equal(matches0, out["matches0"].cpu().numpy())

"""CPU: the drop-in boundary -- library exports, state-dict layout, weight packing, error behaviour.
No kernel is launched here (no GPU in the build container)."""
import ctypes as C
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

from openglue_amd import _lib, synthetic as syn
from openglue_amd.superglue import SuperGlue
from oracle import superglue_oracle as orc
from tests.packed_model import forward_from_packed
from tests.util import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "openglue_amd.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t)\s+(og_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.og_abi_version() == _lib.OG_ABI_VERSION


def test_struct_sizes_match_header():
    # og_shape: 8 ints + 8 hidden + int + float + int + float = 20 * 4 bytes
    assert C.sizeof(_lib.og_shape) == 80
    assert C.sizeof(_lib.og_conv) == 16 and C.sizeof(_lib.og_bn) == 32
    assert C.sizeof(_lib.og_layer_params) == 5 * 16 + 32 + 16 + 8     # + favor_projection (ABI v4)
    assert C.sizeof(_lib.og_inputs) == 6 * 8 + 16 and C.sizeof(_lib.og_outputs) == 7 * 8


def test_state_dict_layout_matches_reference_names():
    cfg = syn.make_config(**{k: v for k, v in syn.CONFIGS["C1"].items() if k not in ("kpts", "batch")})
    model = SuperGlue(cfg)
    spec = syn.state_dict_spec(cfg)             # proven against the reference by make_golden.py (strict load)
    sd = model.state_dict()
    assert set(sd) == set(spec)
    for k, (shape, *_rest) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    # Lightning checkpoints prefix keys with 'superglue.' (inference.py:71-73): strip + strict load
    ck = {"superglue." + k: v for k, v in syn.make_state_dict(cfg, 0).items()}
    stripped = {k[len("superglue."):]: v for k, v in ck.items()}
    assert not any(model.load_state_dict(stripped, strict=True))


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference not mounted")
def test_state_dict_equals_live_reference():
    sys.path.insert(0, "/root/reference")
    from models.superglue.superglue import SuperGlue as Ref
    for kw in (dict(num_heads=4), dict(num_heads=1, attention="favor_relu")):      # favor_relu: + the projection_matrix buffers
        cfg = syn.make_config(descriptor_dim=64, num_stages=2, num_iters=3, side_info_size=6, **kw)
        a, b = Ref(cfg).state_dict(), SuperGlue(cfg).state_dict()
        assert list(a.keys()) == list(b.keys())          # same names in the same registration order
        assert {k: tuple(v.shape) for k, v in a.items()} == {k: tuple(v.shape) for k, v in b.items()}
        assert list(syn.state_dict_spec(cfg)) == list(a.keys())


@pytest.mark.parametrize("name", ["c1", "flags", "nodesc", "mid", "siren", "linear", "favor"])
def test_pack_weights_algebra_against_oracle(name):
    """og_pack_weights (BN folds, out_proj -> fc.0 fold, q pre-scale, padding) evaluated on the CPU in
    float64 must reproduce the oracle: proves the packed blob og_forward consumes is right."""
    z, cfg, sd, data = load_case(name)
    model = SuperGlue(cfg).eval()
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        got = forward_from_packed(model, data)
        ref = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)
    assert (got["scores"] - ref["scores"]).abs().max() < 2e-4      # packed weights are fp32-rounded
    assert (got["context_descriptors0"] - ref["context_descriptors0"]).abs().max() < 2e-5
    assert np.abs(got["scores"].float().numpy() - z["scores"]).max() < 3e-4


@pytest.mark.parametrize("D", [256, 128])
def test_fragment_major_streams_in_the_blob_equal_the_hl32_matrices(D):
    """D = 256 / 128: og_pack_weights stores every layer's q | k | v matrix TWICE -- hl32 rows for the tile GEMM and a fragment-major stream for the
    small-batch projection kernel (o_wqkvs, ABI v7; csrc/mlp_fused.hip: og_pack_proj_stream) -- with the same power-of-two pre-scale.  Both copies
    are re-read here with numpy and must hold the same numbers; favor_relu and other widths have no stream (-1)."""
    from tests.packed_model import layout_of
    cfg = syn.make_config(descriptor_dim=D, num_stages=1, num_heads=4, num_iters=3)
    model = SuperGlue(cfg).eval()
    model.load_state_dict(syn.make_state_dict(cfg, seed=3), strict=True)
    L = layout_of(model)
    assert L.o_wqkvs >= 0 and L.o_wmlp >= 0
    raw = model.pack_host()
    half = raw.view(np.float16).astype(np.float64)
    N = 3 * D
    for l in range(2):
        base = L.layer0 + l * L.layer_stride
        g = half[2 * (base + L.o_wqkv):2 * (base + L.o_wqkv) + 2 * N * D].reshape(N, D // 32, 2, 32)
        w_rows = (g[:, :, 0] + g[:, :, 1]).reshape(N, D)                                   # S w from the hl32 rows
        st = half[2 * (base + L.o_wqkvs):2 * (base + L.o_wqkvs) + 2 * N * D].reshape(N // 32, D // 16, 2, 64, 8)      # [block][k-step][part][lane][e]
        w_st = np.zeros((N, D))
        for lane in range(64):
            rho, hh = lane & 31, lane >> 5
            for e in range(8):
                w_st[rho::32, 8 * hh + e::16] = st[:, :, 0, lane, e] + st[:, :, 1, lane, e]
        assert np.array_equal(w_rows, w_st)
        assert np.abs(w_rows).max() > 1.0                                                  # (pre-scaled by 256: not all zeros)
    small = SuperGlue(syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3)).eval()
    assert layout_of(small).o_wqkvs == -1 and layout_of(small).o_wmlp == -1


def test_repack_when_parameters_change():
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3)
    model = SuperGlue(cfg).eval()
    a = model.pack_host().copy()
    k0 = model._param_key("cpu")
    with torch.no_grad():
        model.linear_proj.bias.add_(1.0)
    assert model._param_key("cpu") != k0
    assert np.abs(model.pack_host() - a).max() > 0.5


def test_error_behaviour():
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3)
    bad = dict(cfg); bad["positional_encoding"] = dict(cfg["positional_encoding"], encoder_name="Nope")
    with pytest.raises(NameError):           # reference: get_positional_encoder raises NameError (__init__.py:39-42)
        SuperGlue(bad)
    bad = dict(cfg); bad["attention_gnn"] = dict(cfg["attention_gnn"], attention="favor_relu")     # 4 heads: the reference's
    with pytest.raises(ValueError):                                                                # forward raises in torch.matmul
        SuperGlue(bad)
    bad = dict(cfg); bad["attention_gnn"] = dict(cfg["attention_gnn"], attention="favor_softmax")  # unreachable upstream
    with pytest.raises(ValueError):
        SuperGlue(bad)
    fav = dict(cfg); fav["attention_gnn"] = dict(cfg["attention_gnn"], attention="favor_relu", num_heads=1)
    fm = SuperGlue(fav).eval()
    k0 = fm._param_key("cpu")
    fm.attention_gnn.layers[0].module.mha.attention_func.resample_projection()      # the redraw callback's hook: triggers a re-pack
    assert fm._param_key("cpu") != k0
    model = SuperGlue(cfg).eval()
    data = syn.make_batch(1, 16, 16, 64, 1, seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # product path never computes on the CPU
        model(data)
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # ... in training mode neither (openglue_amd.train)
        model.train()(data)
    model.eval()
    lib = _lib.load()
    s = model._shape(1, 16, 16)
    assert lib.og_check_shape(C.byref(s)) == 0
    s.desc_dim = 100
    assert lib.og_check_shape(C.byref(s)) == -2 and lib.og_workspace_bytes(C.byref(s)) == 0
    s = model._shape(1, 16, 9000)
    assert lib.og_check_shape(C.byref(s)) == -2
    s = model._shape(1, 16, 16); s.flags = 64
    assert lib.og_check_shape(C.byref(s)) == -4
    # NULL arguments are rejected before any launch (no GPU needed)
    assert lib.og_forward(None, None, None, None, None, None) == -1
    assert lib.og_gemm_nt(None, 4, 0, None, 4, 0, None, 4, 0, 1, 1, 4, 1, None, 0, None, 0, None, 1.0, None) == -1
    # split-f16 GEMM (hl32 operand rows): NULL operands, K not a multiple of the 32-channel slab, row stride < 2K, N % 32 with
    # hl32 output -- all rejected before any launch (the pointers below are never dereferenced)
    fake = 0x10000
    g = lambda A, lda, B, ldb, M, N, K, chl=0, Ch=None, C32=fake: lib.og_gemm_nt_f16x3(A, lda, B, ldb, M, N, K, 1.0, None, 0, None, N,
                                                                                     C32, N, Ch, None, 2 * N, chl, None)
    assert g(None, 128, fake, 128, 64, 64, 64) == -1
    assert g(fake, 128, fake, 128, 64, 64, 48) == -3            # K % 32
    assert g(fake, 100, fake, 128, 64, 64, 64) == -3            # lda < 2K
    assert g(fake, 128, fake, 128, 64, 48, 64, chl=1, Ch=fake) == -3      # hl32 output needs whole 32-channel groups
    assert g(fake, 128, fake, 128, 64, 64, 64, C32=None) == -1  # no output at all
    assert lib.og_split_f16_hl(fake, 8, 48, 48, fake, 96, None) == -3       # cols % 32
    assert lib.og_split_f16_hl(None, 8, 64, 64, fake, 128, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.load()


def test_pack_survives_trained_checkpoint_statistics():
    """A BatchNorm fold over a dead post-ReLU channel (running_var ~ 0) multiplies a weight by ~316 * gamma; 256 * w then leaves
    binary16 (round 1 packed an inf, round 2 refused with OG_E_RANGE).  Now the matrix gets a smaller power-of-two pre-scale, its
    1 / S is stored next to it, and the packed model still reproduces the oracle (packing algebra on the CPU); only non-finite
    weights are refused."""
    from tests.packed_model import forward_from_packed, layout_of
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3)
    model = SuperGlue(cfg).eval()
    model.load_state_dict(syn.make_state_dict(cfg, seed=0), strict=True)
    L = layout_of(model)
    blob = model.pack_host()                                    # as generated: every matrix at the default 256
    base = L.layer0
    assert list(blob[base + L.o_scale:base + L.o_scale + 3]) == [1 / 256.0] * 3 and float(blob[L.scales]) == 1 / 256.0
    bn = model.attention_gnn.layers[0].module.fc[2]
    with torch.no_grad():
        bn.running_var[5] = 0.0                                 # dead post-ReLU channel
        bn.running_mean[5] = 0.0
        bn.weight[5] = 4.0
        model.attention_gnn.layers[0].module.fc[3].weight[:, 5, 0] = 0.9     # 0.9 * 4 / sqrt(1e-5) * 256 = 2.9e5 > 65504
        model.attention_gnn.layers[1].module.mha.in_proj_k.weight[7, 3, 0] = 700.0      # a plain large weight
    blob = model.pack_host()
    inv3 = float(blob[base + L.o_scale + 2])
    assert inv3 > 1 / 256.0 and math.log2(inv3) == round(math.log2(inv3))               # fc.3 of layer 0: a smaller power of two
    assert float(blob[base + L.o_scale + 1]) == 1 / 256.0                                 # its fc.0 is untouched
    assert float(blob[base + L.layer_stride + L.o_scale]) > 1 / 256.0                     # q | k | v of layer 1
    data = syn.make_batch(1, 40, 33, 64, 1, seed=5)
    with torch.no_grad():
        ref = orc.superglue_forward(model.state_dict(), cfg, data, dtype=torch.float64)
        got = forward_from_packed(model, data)
    assert (got["scores"] - ref["scores"]).abs().max() < 1e-6 * max(1.0, float(ref["scores"].abs().max()))
    with torch.no_grad():
        model.linear_proj.weight[3, 3, 0] = float("nan")
    with pytest.raises(RuntimeError, match="OG_E_RANGE"):
        model.pack_host()


def test_bench_self_spawns_one_process_per_gpu():
    """`python bench.py --gpus 2` invoked directly (no torch.distributed environment) must run: it
    re-executes itself under torch.distributed.run.  Dry run = gloo + stub matcher (no GPU here): checks the launch, the
    rendezvous on 127.0.0.1, the cost-balanced ragged gather and that rank 0 prints ONE JSON line with n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OG_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["gather_ok"] is True


def test_input_validation_before_any_launch():
    """mismatched batch / channel dims must raise in Python, not read out of bounds on the device."""
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3)
    model = SuperGlue(cfg).eval()
    p = syn.make_pair(10, 12, 64, 1, seed=0)
    p["image0_size"] = p["image1_size"] = list(syn.IMAGE_WH)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.pack_ragged([p])
    with pytest.raises(ValueError):
        model.pack_ragged([])


def test_openglue_matcher_contract_on_cpu():
    """examples/openglue_matcher.py: OpenGlueMatcher keeps the reference constructor (local_feature, matcher, match_config) and its error
    behaviour: unknown LAF method -> NameError (laf_converter.py:128); CPU tensors -> RuntimeError (no CPU path)."""
    from examples.openglue_matcher import OpenGlueMatcher
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=3, side_info_size=2)
    cfg["laf_to_sideinfo_method"] = "scale"
    model = SuperGlue(cfg).eval()
    mc = {"superglue": cfg, "inference": {"match_threshold": 0.2}}
    pipe = OpenGlueMatcher(None, model, mc)
    assert not pipe.training and pipe.no_match_output(torch.device("cpu"), torch.float32)["lafs0"].shape == (0, 0, 2, 3)
    bad = dict(cfg); bad["laf_to_sideinfo_method"] = "nope"
    with pytest.raises(NameError):
        OpenGlueMatcher(None, model, {"superglue": bad, "inference": {"match_threshold": 0.2}})
    data = {"lafs0": torch.randn(1, 8, 2, 3), "lafs1": torch.randn(1, 9, 2, 3), "responses0": torch.rand(1, 8), "responses1": torch.rand(1, 9),
            "descriptors0": torch.randn(1, 8, 64), "descriptors1": torch.randn(1, 9, 64), "image0_size": [640, 480], "image1_size": [640, 480]}
    with pytest.raises(RuntimeError, match="no CPU path"):
        pipe(data)


def test_resident_sinkhorn_tile_geometry():
    """Host arithmetic of csrc/sinkhorn_resident.hip (rs_geom / rs_pairs_per_round), no device involved: how a pair is cut into workgroup
    tiles on 8 XCDs x 32 CUs -- {W, X column blocks, Gx row blocks, pairs per launch} -- for the BASELINE shapes and the edges."""
    import ctypes as C
    lib = _lib.load()
    def geom(m, n):
        out = (C.c_int32 * 4)()
        rc = lib.og_sinkhorn_resident_geometry(m, n, out)
        return rc, tuple(out)
    assert geom(1024, 1024) == (0, (1, 1, 8, 32))        # C2: 8 tiles of 128 x 1024 per pair, 32 pairs per launch (4 per XCD)
    assert geom(2048, 2048) == (0, (2, 1, 32, 8))        # C3: 32 tiles of 64 x 2048, one pair per XCD
    assert geom(4096, 4096) == (0, (1, 4, 32, 2))        # C4: 4 column blocks x 32 row blocks of 128 x 1024, 2 pairs per launch
    assert geom(77, 4096) == (0, (4, 1, 8, 32))          # short and wide: 32 x 4096 tiles, at least 2 W workgroups (an owner sums <= 512 columns)
    assert geom(1500, 3000) == (0, (2, 2, 24, 4))        # 2 x 24 tiles of 64 x 2048
    assert geom(2500, 1800) == (0, (1, 2, 20, 4))        # 2 x 20 tiles of 128 x 1024
    assert geom(1, 1) == (0, (1, 1, 2, 128))
    assert geom(512, 2048)[1][:3] == (2, 1, 8)
    for m, n in [(4097, 1024), (1024, 4097), (0, 5), (8192, 8192)]:
        assert geom(m, n)[0] == -2, (m, n)
    # every geometry fits the part: X Gx workgroups per pair, pairs per launch x that <= 256
    for m in (1, 100, 129, 1000, 2049, 4096):
        for n in (1, 1024, 1025, 2048, 2049, 4096):
            rc, (W, X, Gx, ppr) = geom(m, n)
            assert rc == 0 and Gx <= 32 and X * Gx * ppr <= 256 and ppr >= 1
            assert X * 1024 * W >= n and Gx * (128 // W) >= m



def test_resident_sinkhorn_ragged_footprint_fits_its_workspace_slot():
    """a ragged batch launches every width class with its own tile width W, and a pair with FEWER rows can take a wider tile
    than the maxima (m_max, n_max) would -- lens (2100 x 900) + 8 x (1000 x 2100): the maxima give W = 1, the 1000 x 2100 pairs W = 4.
    The exchange slot inside og_sinkhorn_workspace_bytes(batch, m_max, n_max) must hold the widest launch of the plan."""
    import ctypes as C
    import random
    lib = _lib.load()
    def footprint(l0, l1):
        B = len(l0)
        a0 = (C.c_int32 * B)(*l0); a1 = (C.c_int32 * B)(*l1)
        out = (C.c_int64 * 2)()
        rc = lib.og_sinkhorn_resident_ragged_footprint(B, a0, a1, out)
        return rc, int(out[0]), int(out[1])
    def geomW(m, n):
        out = (C.c_int32 * 4)()
        assert lib.og_sinkhorn_resident_geometry(m, n, out) == 0
        return out[0]
    l0 = [2100] + [1000] * 8; l1 = [900] + [2100] * 8
    assert geomW(max(l0), max(l1)) == 1 and geomW(1000, 2100) == 4      # the advisor's example: the maxima alone would under-size the slot
    rc, launches, bytes_ = footprint(l0, l1)
    assert rc == 0 and launches >= 2
    assert bytes_ <= lib.og_sinkhorn_workspace_bytes(len(l0), max(l0), max(l1))
    rng = random.Random(5)
    checked = 0
    for _ in range(300):
        B = rng.randint(1, 32)
        hi = rng.choice([600, 1024, 1500, 2048, 3000, 4096])
        l0 = [rng.randint(1, hi) for _ in range(B)]
        l1 = [rng.randint(1, rng.choice([1024, 2048, 4096])) for _ in range(B)]
        rc, launches, bytes_ = footprint(l0, l1)
        if rc != 0:
            assert rc == -2
            continue
        ws = lib.og_sinkhorn_workspace_bytes(B, max(l0), max(l1))
        assert launches >= 1 and 0 < bytes_ <= ws, (l0, l1, bytes_, ws)
        checked += 1
    assert checked > 100


def test_favor_base_is_an_explicit_switch():
    """VERDICT r4 weak 9 / ADVICE: the class the FAVOR buffer containers derive from is chosen by register_favor_base() or the favor_base
    argument, never by what happens to be importable.  Both branches: the built-in container and a host-style base class."""
    import torch.nn as nn
    from openglue_amd import superglue as ogs
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=1, num_iters=3, attention="favor_relu")

    class HostFavor(nn.Module):                       # the shape of the reference's FavorAttention (attention.py:43-70), not its code
        def __init__(self, embed_dim, num_orthogonal_features):
            super().__init__()
            self.calls = 0
            self.register_buffer("projection_matrix", torch.zeros(num_orthogonal_features, embed_dim))

        def resample_projection(self):
            self.calls += 1
            self.projection_matrix.copy_(torch.full_like(self.projection_matrix, float(self.calls)))

    try:
        plain = ogs.SuperGlue(cfg)
        assert all(type(m) is ogs._FavorFeatures for m in plain.modules() if hasattr(m, "projection_matrix"))
        hosted = ogs.SuperGlue(cfg, favor_base=HostFavor)
        mods = [m for m in hosted.modules() if isinstance(m, HostFavor)]
        assert len(mods) == 2 and set(hosted.state_dict()) == set(plain.state_dict())
        v = mods[0].projection_matrix._version
        mods[0].resample_projection()
        assert mods[0].calls == 1 and mods[0].projection_matrix._version > v
        ogs.register_favor_base(HostFavor)
        assert len([m for m in ogs.SuperGlue(cfg).modules() if isinstance(m, HostFavor)]) == 2
        ogs.register_favor_base(None)
        assert not any(isinstance(m, HostFavor) for m in ogs.SuperGlue(cfg).modules())
        with pytest.raises(TypeError):
            ogs.register_favor_base(int)
        class Bad(nn.Module):
            def __init__(self, embed_dim, num_orthogonal_features):
                super().__init__()
        with pytest.raises(TypeError, match="projection_matrix"):
            ogs.SuperGlue(cfg, favor_base=Bad)
    finally:
        ogs.register_favor_base(None)


def test_favor_without_registered_base_warns_once_when_the_host_class_is_loaded(monkeypatch):
    """with the reference's FavorAttention importable in the process but NOT registered, its redraw callback (isinstance test) finds nothing and the
    projection is silently never resampled -- the module says so once (nothing is imported from the host: only sys.modules is looked at)."""
    import sys, types, warnings
    import openglue_amd.superglue as ogs
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=1, num_iters=2, attention="favor_relu")
    monkeypatch.setattr(ogs, "_FAVOR_WARNED", False)
    monkeypatch.setitem(sys.modules, "models.superglue.attention", types.ModuleType("models.superglue.attention"))
    ogs.register_favor_base(None)
    with pytest.warns(RuntimeWarning, match="register_favor_base"):
        ogs.SuperGlue(cfg)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ogs.SuperGlue(cfg)                      # once per process


def test_resident_sinkhorn_few_pairs_geometry_selection(monkeypatch):
    """Host arithmetic of the few-pairs geometries (csrc/sinkhorn_resident.hip: rs_rows_per_wave, round 5): launches of one to eight pairs of
    <= 1024 x 1024 keypoints take 4 rows per wave (a pair = 32 tiles on one XCD), 9 to 16 pairs take 8, anything that fills the chip -- or is wider
    than one wave tile, or has more than 1024 rows -- keeps 16."""
    lib = _lib.load()
    monkeypatch.delenv("OG_SINKHORN_FEW", raising=False)
    rpw = lib.og_sinkhorn_resident_rows_per_wave
    assert [rpw(B, 1024, 1024) for B in (1, 2, 4, 8, 9, 12, 16, 17, 32)] == [4, 4, 4, 4, 8, 8, 8, 16, 16]
    assert rpw(1, 2048, 2048) == 16 and rpw(1, 1024, 1025) == 16 and rpw(1, 1025, 1024) == 8 and rpw(1, 2049, 1024) == 16
    assert rpw(1, 1, 1) == 4 and rpw(64, 100, 100) == 4 and rpw(65, 100, 100) == 16       # tiny pairs: at least two tiles each; the coarser geometry must leave half the chip idle
    assert rpw(1, 5000, 100) == 0 and rpw(0, 10, 10) == 0                                   # no resident geometry at all
    # every selected geometry fits ONE launch of 256 workgroups with one XCD (<= 32 tiles) per pair
    for B in range(1, 40):
        for m in (1, 31, 32, 33, 500, 1000, 1024):
            r = rpw(B, m, 700)
            if r in (4, 8):
                tiles = max(2, -(-m // (r * 8)))
                assert tiles <= 32 and B * tiles <= 256, (B, m, r)
    # a round of the per-XCD map holds 8 (32 / Gx) pairs -- 9 to 12 pairs of 513-900 rows are 20-29 row blocks each at 4 rows per wave (8 pairs per
    # launch: two launches), 10-15 at 8 rows per wave (16-24 pairs: one launch).  The heuristic must pick a geometry whose ONE launch holds the batch.
    for B in (9, 10, 12):
        for m in (640, 800, 900):
            assert rpw(B, m, 1000) == 8, (B, m, rpw(B, m, 1000))
    for B in range(1, 33):
        for m in (100, 513, 640, 800, 1024):
            r = rpw(B, m, 1000)
            if r in (4, 8):
                gx = max(2, -(-m // (r * 8)))
                assert B <= 8 * (32 // gx), (B, m, r, gx)
    monkeypatch.setenv("OG_SINKHORN_FEW", "0")
    assert rpw(1, 1024, 1024) == 16
    monkeypatch.setenv("OG_SINKHORN_FEW", "8")
    assert rpw(1, 1024, 1024) == 8 and rpw(32, 1024, 1024) == 16                             # 8 rows per wave x 32 pairs would need 512 workgroups


def test_training_stacked_projection_weights_follow_their_parameters():
    """openglue_amd.train stacks the q | k | v weights of a layer once per forward call (never across calls: a parameter changed through
    `.data` does not move its version counter): the stack is the concatenation of the CURRENT parameters, carries no graph and never
    enters the state dict."""
    import torch
    from openglue_amd.train import ProjectedAttention as PA

    class MHA(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.in_proj_q, self.in_proj_k, self.in_proj_v = (torch.nn.Conv1d(8, 8, 1) for _ in range(3))

    m = MHA()
    w = lambda c: c.weight[:, :, 0]
    a = PA.stacked(m, True)
    assert torch.equal(a[0], torch.cat([w(m.in_proj_q), w(m.in_proj_k), w(m.in_proj_v)]))
    assert torch.equal(a[1], torch.cat([m.in_proj_q.bias, m.in_proj_k.bias, m.in_proj_v.bias]))
    m.in_proj_k.weight.data.add_(1.0)                             # a change the version counter does not see
    assert torch.equal(PA.stacked(m, True)[0][8:16], w(m.in_proj_k))
    Wq, bq, Wkv, bkv = PA.stacked(m, False)
    assert Wkv.shape == (16, 8) and torch.equal(Wkv[:8], w(m.in_proj_k)) and torch.equal(bkv[8:], m.in_proj_v.bias) and torch.equal(Wq, w(m.in_proj_q))
    assert len(m.state_dict()) == 6 and not any(t.requires_grad for t in PA.stacked(m, True))

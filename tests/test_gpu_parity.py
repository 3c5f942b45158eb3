"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and against the
golden fixtures generated from the reference.  Run on the MI355X box: pytest -m gpu.

Tolerances: BASELINE.json north_star -- log-assignment scores within 1e-3 (fp32), match indices
identical (rows whose top-1/top-2 gap in the float64 oracle is < 1e-4 are exempted and counted:
SURVEY.md §7 "index parity is ill-posed on near-ties").
"""
import math
import os

import numpy as np
import pytest
import torch

from openglue_amd import ops, synthetic as syn
from openglue_amd.superglue import SuperGlue
from oracle import superglue_oracle as orc
from tests.util import GOLDEN, MATCH_THRESHOLD, load_case, parity_note, to_device

pytestmark = pytest.mark.gpu

TOL_SCORES = 1e-3      # the bar
TOL_STAGE = 2e-5       # exact-fp32 stages vs float64 oracle (relative to magnitude)


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 96), (77, 64, 64), (1000, 768, 256), (33, 257, 512)])
def test_gemm_nt_exact_fp32(gpu_device, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a, b = _rand(g, M, K), _rand(g, N, K)          # asymmetric operands: a transposed/ swapped store cannot pass
    bias, res, alpha = _rand(g, N), _rand(g, M, N), torch.rand(N, generator=g)
    ref = a.double() @ b.double().T
    dev = lambda t: t.to(gpu_device)
    out = ops.gemm_nt(dev(a), dev(b)).cpu()
    assert (out.double() - ref).abs().max() < TOL_STAGE * K ** 0.5
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), relu=True).cpu()
    assert (out.double() - torch.relu(ref + bias.double())).abs().max() < TOL_STAGE * K ** 0.5
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), res=dev(res)).cpu()
    assert (out.double() - (ref + bias.double() + res.double())).abs().max() < TOL_STAGE * K ** 0.5
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), res=dev(res), alpha=dev(alpha), scale=0.25).cpu()
    want = (alpha.double() * (ref + bias.double()) + (1 - alpha.double()) * res.double()) * 0.25
    assert (out.double() - want).abs().max() < TOL_STAGE * K ** 0.5


def test_gemm_nt_batched_scores_shape(gpu_device):
    g = torch.Generator().manual_seed(5)
    a, b = _rand(g, 3, 130, 64), _rand(g, 3, 97, 64)
    out = ops.gemm_nt(a.to(gpu_device), b.to(gpu_device), scale=64 ** -0.5).cpu()
    ref = (a.double() @ b.double().transpose(1, 2)) * 64 ** -0.5
    assert out.shape == (3, 130, 97)
    assert (out.double() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 192, 96), (77, 64, 64), (1000, 768, 256), (4099, 512, 512),
                                   (49999, 256, 64), (24700, 512, 256)])     # the last two take the 256x256-tile kernel
def test_gemm_nt_f16x3_fp32_class_accuracy(gpu_device, M, N, K):
    """The split-f16 3-pass GEMM must be as accurate as an fp32 GEMM (both measured against float64)."""
    g = torch.Generator().manual_seed(M * 3 + K)
    a, b = _rand(g, M, K, scale=3.0), _rand(g, N, K, scale=0.05)
    bias, res = _rand(g, N), _rand(g, M, N)
    ref = a.double() @ b.double().T
    dev = lambda t: t.to(gpu_device)
    fp32_err = ((a @ b.T).double() - ref).abs().max().item()
    out = ops.gemm_nt_f16x3(dev(a), dev(b)).cpu()
    err = (out.double() - ref).abs().max().item()
    print(f"[f16x3 {M}x{N}x{K}] err {err:.2e} (fp32 CPU GEMM err {fp32_err:.2e})")
    assert err < max(2.0 * fp32_err, 1e-6)
    out, ch, cl = ops.gemm_nt_f16x3(dev(a), dev(b), bias=dev(bias), relu=True, res=dev(res), want="planes")
    want = torch.relu(ref + bias.double()) + res.double()
    assert (out.cpu().double() - want).abs().max() < max(2.0 * fp32_err, 1e-6)
    merged = ops.merge_f16(ch, cl).cpu()
    assert (merged.double() - want).abs().max() < max(2.0 * fp32_err, 1e-6) + 1e-6 * want.abs().max()
    out2, chl = ops.gemm_nt_f16x3(dev(a), dev(b), bias=dev(bias), relu=True, res=dev(res), want="hl")     # hl32 output rows
    assert torch.equal(out2, out)
    assert torch.equal(ops.merge_f16_hl(chl).cpu(), merged)


@pytest.mark.parametrize("M,N,K,form", [(49152, 768, 256, "planes"), (49152, 512, 512, "relu"), (49152, 256, 512, "res_hl"),
                                        (49152, 256, 256, "relu_res_hl"), (24700, 512, 256, "res_hl"), (300, 192, 96, "res_hl")])
def test_gemm_nt_f16x3_gnn_forms(gpu_device, M, N, K, form):
    """The launch forms of the GNN (split-f16 output only): whole-tile shapes take the 256-tile kernel whose epilogue arithmetic is
    fixed at compile time (q/k/v planes, fc.0 ReLU rows, fc.3 rows with the (hi, lo) residual); ragged / small shapes and
    ReLU + residual take the run-time forms.  All must agree with float64 to fp32-GEMM accuracy, and with OG_GEMM_SPEC_EPI=0
    the generic epilogue path (OG_GEMM_SPEC_EPI=0, exercised by scripts/gpu_gemm_epi.sh) must give the same bits."""
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _rand(g, M, K, scale=3.0), _rand(g, N, K, scale=0.05)
    bias = _rand(g, N)
    res = _rand(g, M, N, scale=5.0) if "res_hl" in form else None
    relu = "relu" in form
    dev = lambda t: t.to(gpu_device) if t is not None else None
    out = ops.gemm_nt_f16x3_split_only(dev(a), dev(b), bias=dev(bias), relu=relu, res=dev(res), planes=form == "planes").cpu()
    rows = torch.cat([torch.arange(0, 700), torch.arange(M - 300, M)]) if M > 1000 else torch.arange(M)     # float64 reference on a slice
    ref = a[rows].double() @ b.double().T + bias.double()
    if relu: ref = torch.relu(ref)
    if res is not None:
        res_hl = ops.merge_f16_hl(ops.split_f16_hl(dev(res))).cpu()      # what the kernel is given: the (hi, lo) representation
        ref = ref + res_hl[rows].double()
    fp32_err = ((a[rows] @ b.T).double() - a[rows].double() @ b.double().T).abs().max().item()
    err = (out[rows].double() - ref).abs().max().item()
    print(f"[f16x3 {form} {M}x{N}x{K}] err {err:.2e} (fp32 CPU GEMM err {fp32_err:.2e})")
    assert err < max(2.0 * fp32_err, 1e-6) + 2e-6 * ref.abs().max().item()       # + the (hi, lo) output representation
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("D", [256, 128])              # 128: the reference's SIFT / HardNet width (config/features/sift_opencv.yaml:2; attention_gnn.py:41 is generic in D)
@pytest.mark.parametrize("M", [128, 4096, 32768, 777, 1, 8192, 8320, 33])       # <= 8192 rows: mlp_small_kernel (32-token workgroups), above: 128-token tiles
def test_mlp_block_fused_vs_float64(gpu_device, M, D):
    """og_mlp_block (csrc/mlp_fused.hip): x + W3 relu(W0 [x ; O] + b0) + b3 in ONE launch with the hidden activation in registers
    (attention_gnn.py:53-55 + models/utils.py:48-58 after the folds of og_pack_weights) -- against float64 and against the two
    split-f16 GEMM launches it replaces.  Asymmetric random operands: a wrong fragment permutation cannot pass.  M = 777 / 1: partial
    tiles (clamped loads, predicated stores); rows past M must stay untouched."""
    g = torch.Generator().manual_seed(1000 + M)
    x, o = _rand(g, M, D, scale=2.0), _rand(g, M, D, scale=1.5)
    w0, w3 = _rand(g, 2 * D, 2 * D, scale=0.04), _rand(g, D, 2 * D, scale=0.05)
    b0, b3 = _rand(g, 2 * D, scale=0.3), _rand(g, D, scale=0.3)
    dev = lambda t: t.to(gpu_device)
    out, rows = ops.mlp_block(dev(x), dev(o), dev(w0), dev(b0), dev(w3), dev(b3), return_rows=True)
    out = out.cpu()
    xo_in = ops.merge_f16_hl(ops.split_f16_hl(dev(torch.cat([x, o], 1).contiguous()))).cpu()       # what the kernel is given
    sl = torch.cat([torch.arange(0, min(M, 400)), torch.arange(max(M - 300, 0), M)]).unique()
    h = torch.relu(xo_in[sl].double() @ w0.double().T + b0.double())
    ref = xo_in[sl, :D].double() + h @ w3.double().T + b3.double()
    err = (out[sl].double() - ref).abs().max().item()
    h32 = torch.relu(xo_in[sl] @ w0.T + b0)
    fp32_err = ((xo_in[sl, :D] + h32 @ w3.T + b3).double() - ref).abs().max().item()
    print(f"[mlp_block D={D} M={M}] err {err:.2e} (fp32 CPU err {fp32_err:.2e})")
    assert torch.isfinite(out).all()
    assert err < max(2.0 * fp32_err, 2e-6) + 2e-6 * ref.abs().max().item()
    # the O half of the rows is read-only
    assert torch.equal(ops.merge_f16_hl(rows)[:, D:].cpu(), xo_in[:, D:])
    # the two-launch form (fc.0 with ReLU -> hl32 rows; fc.3 with the (hi, lo) residual): same operands, same hidden rounding
    hid = ops.gemm_nt_f16x3_split_only(dev(xo_in), dev(w0), bias=dev(b0), relu=True)
    two = ops.gemm_nt_f16x3_split_only(hid, dev(w3), bias=dev(b3), res=dev(xo_in[:, :D].contiguous())).cpu()
    assert (two - out).abs().max().item() < 2e-5 + 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("D", [256, 128])
def test_proj_block_wide_matrix(gpu_device, D):
    """N = 1024 output channels (32 blocks): more than the 3 blocks per wave one workgroup covers -- the launcher must deal the range out."""
    N, M = 1024, 4000
    g = torch.Generator().manual_seed(77)
    x, w, b = _rand(g, M, D, scale=2.0), _rand(g, N, D, scale=0.06), _rand(g, N, scale=0.3)
    out = ops.proj_block(x.to(gpu_device), w.to(gpu_device), b.to(gpu_device)).cpu()
    x_in = ops.merge_f16_hl(ops.split_f16_hl(x.to(gpu_device))).cpu()
    ref = x_in.double() @ w.double().T + b.double()
    assert (out.double() - ref).abs().max().item() < 1e-5 + 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("D", [256, 128])
@pytest.mark.parametrize("M,split,cols_a,cols_b", [(2048, 0, None, None), (1, 0, None, None), (33, 0, None, None), (8192, 0, None, (256, 768)),
                                                   (1000, 512, (0, 256), (0, 768)), (96, 64, (0, 256), (0, 768)), (777, 0, None, (0, 256))])
def test_proj_block_small_batch_vs_float64(gpu_device, M, split, cols_a, cols_b, D):
    """og_proj_block (csrc/mlp_fused.hip: proj_small_kernel), the q / k / v projection kernel og_forward uses for launches of <= 8192 token
    rows (attention_gnn.py:43-47): against float64 on the (hi, lo) operands the kernel is given.  Partial tiles, one / two / three output
    blocks per wave, the cross layer's row split (rows of image 0: the q columns only); columns outside a row's range stay untouched.
    D = 128: K = 128 (8 k-steps), N = 384 = the q | k | v matrix of the 128-d family (12 blocks: one or two per wave)."""
    N = 3 * D
    sc = lambda c: None if c is None else (c[0] * D // 256, c[1] * D // 256)       # the column ranges were written for N = 768
    cols_a, cols_b = sc(cols_a), sc(cols_b)
    g = torch.Generator().manual_seed(2000 + M)
    x, w, b = _rand(g, M, D, scale=2.0), _rand(g, N, D, scale=0.06), _rand(g, N, scale=0.3)
    dev = lambda t: t.to(gpu_device)
    out = ops.proj_block(dev(x), dev(w), dev(b), split_row=split, cols_a=cols_a, cols_b=cols_b).cpu()
    x_in = ops.merge_f16_hl(ops.split_f16_hl(dev(x))).cpu()
    ref = x_in.double() @ w.double().T + b.double()
    fp32_err = ((x_in @ w.T + b).double() - ref).abs().max().item()
    ca, cb = cols_a or (0, N), cols_b or (0, N)
    mask = torch.zeros(M, N, dtype=torch.bool)
    mask[:split, ca[0]:ca[1]] = True
    mask[split:, cb[0]:cb[1]] = True
    err = ((out.double() - ref).abs() * mask).max().item()
    print(f"[proj_block D={D} M={M} split={split}] err {err:.2e} (fp32 CPU err {fp32_err:.2e})")
    assert torch.isfinite(out).all()
    assert err < max(2.0 * fp32_err, 2e-6) + 2e-6 * ref.abs().max().item()
    assert (out[~mask] == 0).all()                 # nothing written outside the requested ranges


@pytest.mark.parametrize("D", [256, 128])
@pytest.mark.parametrize("M,split,cols_a,cols_b", [(8320, 0, None, None), (32768, 0, None, None), (9001, 0, None, None), (16384, 0, None, (256, 768)),
                                                   (20000, 8192, (0, 256), (0, 768)), (65536, 32768, (0, 256), (0, 768)), (8200, 0, None, (512, 768))])
def test_proj_block_batch_kernel_vs_float64(gpu_device, M, split, cols_a, cols_b, D):
    """og_proj_block above 8192 rows = proj_stream_kernel (csrc/mlp_fused.hip; round 5): the q / k / v projections of a batch with the x fragments
    in registers and the weights through an LDS ring (attention_gnn.py:43-47).  Whole and partial 128-token tiles, one to six 128-channel groups,
    the cross layer's row split, a column range that starts in the middle of the matrix (k | v of the updated image 0)."""
    N = 3 * D
    sc = lambda c: None if c is None else (c[0] * D // 256, c[1] * D // 256)
    cols_a, cols_b = sc(cols_a), sc(cols_b)
    g = torch.Generator().manual_seed(3000 + M)
    x, w, b = _rand(g, M, D, scale=2.0), _rand(g, N, D, scale=0.06), _rand(g, N, scale=0.3)
    dev = lambda t: t.to(gpu_device)
    out = ops.proj_block(dev(x), dev(w), dev(b), split_row=split, cols_a=cols_a, cols_b=cols_b).cpu()
    x_in = ops.merge_f16_hl(ops.split_f16_hl(dev(x))).cpu()
    rows = torch.cat([torch.arange(0, 600), torch.arange(max(split - 300, 0), min(split + 300, M)), torch.arange(M - 400, M)]).unique()
    ref = x_in[rows].double() @ w.double().T + b.double()
    fp32_err = ((x_in[rows] @ w.T + b).double() - ref).abs().max().item()
    ca, cb = cols_a or (0, N), cols_b or (0, N)
    mask = torch.zeros(M, N, dtype=torch.bool)
    mask[:split, ca[0]:ca[1]] = True
    mask[split:, cb[0]:cb[1]] = True
    err = ((out[rows].double() - ref).abs() * mask[rows]).max().item()
    print(f"[proj_block batch D={D} M={M} split={split}] err {err:.2e} (fp32 CPU err {fp32_err:.2e})")
    assert torch.isfinite(out).all()
    assert err < max(2.0 * fp32_err, 2e-6) + 2e-6 * ref.abs().max().item()
    assert (out[~mask] == 0).all()                 # nothing written outside the requested ranges
    # and the small-batch kernel on the same operands (OG_PROJ_STREAM is read once per process: compare through the forced tile GEMM instead)
    tile = ops.gemm_nt_f16x3_split_only(dev(x_in[rows]), dev(w), bias=dev(b), planes=True).cpu()
    assert ((tile - out[rows]).abs() * mask[rows]).max().item() < 2e-5 + 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("D", [256, 128])
def test_mlp_block_rows_past_m_untouched(gpu_device, D):
    from openglue_amd import _lib
    lib = _lib.load()
    M, R = 200, 384
    g = torch.Generator().manual_seed(7)
    xo = _rand(g, R, 2 * D).to(gpu_device)
    w0, w3 = _rand(g, 2 * D, 2 * D, scale=0.04), _rand(g, D, 2 * D, scale=0.05)
    b0, b3 = _rand(g, 2 * D).to(gpu_device), _rand(g, D).to(gpu_device)
    st = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
    _lib.check(lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), st.data_ptr()), "pack")
    st = st.to(gpu_device)
    rows = ops.split_f16_hl(xo)
    before = rows.clone()
    _lib.check(lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, st.data_ptr(), b0.data_ptr(), b3.data_ptr(), torch.cuda.current_stream().cuda_stream), "og_mlp_block")
    torch.cuda.synchronize()
    assert torch.equal(rows[M:], before[M:])
    assert not torch.equal(rows[:M, :2 * D], before[:M, :2 * D])
    assert torch.equal(rows[:M, 2 * D:], before[:M, 2 * D:])


def test_split_f16_roundtrip(gpu_device):
    g = torch.Generator().manual_seed(1)
    x = torch.cat([_rand(g, 1000, scale=s) for s in (1e-3, 1.0, 50.0, 3000.0)])
    hi, lo = ops.split_f16(x.to(gpu_device))
    back = ops.merge_f16(hi, lo).cpu()
    assert ((back - x).abs() <= 2.0 ** -21 * x.abs() + 2.0 ** -25).all()      # lo may be an f16 subnormal (spacing 2^-24)
    x2 = x.view(125, 32)                                   # the same values as hl32 rows
    hl = ops.split_f16_hl(x2.to(gpu_device))
    assert hl.shape == (125, 64)
    assert torch.equal(hl[:, :32].cpu(), hi.view(125, 32).cpu()) and torch.equal(hl[:, 32:].cpu(), lo.view(125, 32).cpu())
    assert torch.equal(ops.merge_f16_hl(hl).cpu(), back.view(125, 32))


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, H):
    return orc.softmax_attention(q.double(), k.double(), v.double(), H)     # applies d^-1/2 itself


@pytest.mark.parametrize("H,dh,nq,nk", [(4, 16, 50, 70), (4, 32, 129, 64), (4, 64, 300, 257), (2, 64, 128, 1024), (1, 32, 1, 1),
                                        (4, 32, 300, 257), (4, 16, 300, 257), (2, 32, 40, 65),    # several key tiles at every head size
                                        (1, 64, 1, 1), (2, 64, 70, 63), (3, 64, 129, 65), (1, 64, 33, 128)])   # dh=64 = the LDS-DMA kernel: single / partial / odd tiles
def test_attention_vs_oracle(gpu_device, H, dh, nq, nk):
    g = torch.Generator().manual_seed(H * 100 + dh + nq)
    D = H * dh
    q, k, v = _rand(g, 2, nq, D, scale=3.0), _rand(g, 2, nk, D, scale=3.0), _rand(g, 2, nk, D, scale=2.0)
    ref = _attn_ref(q, k, v, H)
    out = ops.attention((q * dh ** -0.5).to(gpu_device), k.to(gpu_device), v.to(gpu_device), H).cpu()
    err = (out.double() - ref).abs().max().item()
    print(f"[attention H={H} dh={dh} nq={nq} nk={nk}] max abs err {err:.2e}")
    assert err < 5e-5, err                # split-f16 operands: ~22 mantissa bits per product


def test_attention_reference_fixture(gpu_device):
    z = np.load(os.path.join(GOLDEN, "stage_attention.npz"))
    q, k, v = (torch.from_numpy(z[n]) for n in "qkv")          # reference layout [B, H, d, N]
    B, H, d, nq = q.shape
    tok = lambda t: t.permute(0, 3, 1, 2).reshape(B, t.shape[3], H * d).contiguous()
    out = ops.attention((tok(q) * d ** -0.5).to(gpu_device), tok(k).to(gpu_device), tok(v).to(gpu_device), H).cpu()
    ref = torch.from_numpy(z["out"]).permute(0, 3, 1, 2).reshape(B, nq, H * d)
    assert (out - ref).abs().max() < 5e-5


def test_attention_online_softmax_rescale_branch(gpu_device):
    """Force the running-max rescale: one key in the LAST 64-key tile dominates one query (rule 26 of the
    CDNA guide: a rare data-dependent branch needs an input that takes it)."""
    g = torch.Generator().manual_seed(9)
    H, dh, nq, nk = 2, 64, 64, 200
    D = H * dh
    q, k, v = _rand(g, 1, nq, D), _rand(g, 1, nk, D), _rand(g, 1, nk, D)
    k[0, 190, :dh] = q[0, 7, :dh] * 6.0          # spike for (query 7, head 0) at key 190 (tile 2)
    k[0, 3, dh:] = q[0, 40, dh:] * 6.0           # and an early spike for (query 40, head 1): later tiles must not disturb it
    ref = _attn_ref(q, k, v, H)
    out = ops.attention((q * dh ** -0.5).to(gpu_device), k.to(gpu_device), v.to(gpu_device), H).cpu()
    assert (out.double() - ref).abs().max() < 5e-5
    assert torch.isfinite(out).all()


# ----------------------------------------------------------------------------- Sinkhorn
def _sinkhorn_ref(S, z, iters, reg):
    return orc.matching_log_probs(S.double(), torch.tensor(z, dtype=torch.float64), iters, reg)


@pytest.mark.parametrize("B,m,n,iters,reg", [(3, 37, 53, 7, 1.0), (2, 64, 64, 0, 1.0), (2, 64, 64, 1, 1.0),
                                             (1, 130, 1023, 20, 0.7), (2, 257, 1500, 10, 1.0),
                                             (1, 100, 2049, 5, 1.0), (1, 1, 1, 3, 1.0), (1, 17, 4096, 4, 2.0),
                                             (2, 45, 300, 6, 1.0), (1, 70, 5000, 3, 1.0), (2, 33, 2048, 8, 1.0)])
def test_sinkhorn_vs_oracle(gpu_device, B, m, n, iters, reg):
    g = torch.Generator().manual_seed(m * 31 + n)
    S = _rand(g, B, m, n, scale=4.0)
    ref = _sinkhorn_ref(S, 0.7, iters, reg)
    out = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg).cpu()
    assert out.shape == (B, m + 1, n + 1)
    err = (out.double() - ref).abs().max().item()
    assert err < 1e-4, err
    if iters > 0:   # after the last v update the column marginals are exact
        norm = -math.log(m + n)
        lb = torch.full((n + 1,), norm, dtype=torch.float64); lb[-1] += math.log(m)
        assert (torch.logsumexp(out.double() + norm, dim=1) - lb).abs().max() < 1e-4


@pytest.mark.parametrize("scale,reg,z", [(25.0, 0.5, 0.7), (8.0, 0.1, -30.0), (60.0, 1.0, 50.0), (1e-3, 1.0, 0.0)])
def test_sinkhorn_extreme_score_range(gpu_device, scale, reg, z):
    """The iterations after the first run without a running maximum (sinkhorn.hip, dual-stabilised form): scores whose
    range is far beyond exp()'s must neither overflow nor lose the marginals.  |S/reg| reaches several hundred here."""
    g = torch.Generator().manual_seed(int(scale * 10) + 3)
    B, m, n, iters = 2, 96, 200, 30
    S = _rand(g, B, m, n, scale=scale)
    S[0, 5, :] = -4.0 * scale        # a row nobody wants and a column everybody wants
    S[1, :, 7] = 4.0 * scale
    ref = _sinkhorn_ref(S, z, iters, reg)
    out = ops.sinkhorn(S.to(gpu_device), z, iters, reg).cpu()
    assert torch.isfinite(out).all()
    err = (out.double() - ref).abs()
    # scores = S/reg + u + v - norm with |u|, |v|, |S/reg| of several hundred: the attainable fp32 accuracy is a few
    # ulps of THAT magnitude (the max-subtracted kernels measure the same: 4e-4 .. 2e-3 on these inputs)
    tol = 1e-4 + 2e-6 * ref.abs().max().item()
    print(f"[sinkhorn extreme scale={scale} reg={reg} z={z}] max err {err.max().item():.2e} (tol {tol:.1e}), max |score| {ref.abs().max().item():.0f}")
    assert err.max().item() <= tol
    norm = -math.log(m + n)
    lb = torch.full((n + 1,), norm, dtype=torch.float64); lb[-1] += math.log(m)
    assert (torch.logsumexp(out.double() + norm, dim=1) - lb).abs().max() < 2e-4 + 4e-6 * ref.abs().max()


def test_sinkhorn_reference_fixture(gpu_device):
    z = np.load(os.path.join(GOLDEN, "stage_sinkhorn.npz"))
    Mx = torch.from_numpy(z["M"])                    # augmented [B, m+1, n+1]; its dustbin entries are random,
    B, m1, n1 = Mx.shape                             # so only the inner block + a CONSTANT bin can be fed to og_sinkhorn:
    S = Mx[:, :-1, :-1].contiguous()                 # compare against the reference solver re-run by the oracle restatement
    norm = -math.log(m1 - 1 + n1 - 1)
    for iters in (1, 7):
        ref = orc.matching_log_probs(S.double(), torch.tensor(0.5, dtype=torch.float64), iters, 1.0)
        out = ops.sinkhorn(S.to(gpu_device), 0.5, iters, 1.0).cpu()
        assert (out.double() - ref).abs().max() < 1e-4


# ----------------------------------------------------------------------------- match extraction
@pytest.mark.parametrize("name", ["c1", "mid", "flags", "nodesc"])
def test_extract_matches_on_reference_scores(gpu_device, name):
    z, *_ = load_case(name)
    scores = torch.from_numpy(z["scores"])
    got = {k: v.cpu() for k, v in ops.extract_matches(scores.to(gpu_device), MATCH_THRESHOLD).items()}
    want = orc.extract_matches(scores, MATCH_THRESHOLD)
    np.testing.assert_array_equal(got["matches0"].numpy(), z["matches0"])          # brute-force fixture
    np.testing.assert_allclose(got["matching_scores0"].numpy(), z["matching_scores0"], rtol=2e-6)
    assert torch.equal(got["matches1"], want["matches1"])
    np.testing.assert_allclose(got["matching_scores1"].numpy(), want["matching_scores1"].numpy(), rtol=2e-6)


def test_extract_matches_ties_first_index_wins(gpu_device):
    s = torch.full((2, 70, 300), -5.0)
    s[0, 0, 1] = s[0, 0, 200] = 0.0        # row tie -> column 1
    s[0, 5, 7] = s[0, 66, 7] = -1.0        # column tie across two 64-row slabs -> row 5
    s[1, :, :] = -3.0                      # everything ties: row i -> col 0, col j -> row 0
    s[:, -1, :] = 10.0; s[:, :, -1] = 10.0  # dustbins must be ignored
    got = {k: v.cpu() for k, v in ops.extract_matches(s.to(gpu_device), 0.0).items()}
    want = orc.extract_matches(s, 0.0)
    for k in ("matches0", "matches1"):
        assert torch.equal(got[k], want[k]), k
    assert got["matches0"][0, 0] == 1 and got["matches0"][0, 5] == 7 and got["matches0"][0, 66] == -1
    assert got["matches0"][1, 0] == 0 and (got["matches0"][1, 1:] == -1).all()


# ----------------------------------------------------------------------------- whole path
def _build(cfg, sd, device):
    model = SuperGlue(cfg).eval()
    model.load_state_dict(sd, strict=True)
    return model.to(device)


MAX_EXEMPT_FRAC = 1e-3      # ceiling on near-tie exemptions: at most 0.1 % of the rows (and never more than 2 on a tiny case)


def _index_agreement(got_matches0, scores_gpu, sd, cfg, data):
    """matches0 must equal the fp32 oracle's except on rows that are near-ties in float64 -- and those exemptions are BOUNDED:
    more than MAX_EXEMPT_FRAC of the rows differing fails even if every one of them is a near-tie.
    -> (rows that differ = exemptions granted, rows not explained, float64 oracle output)"""
    with torch.no_grad():
        o64 = orc.superglue_forward(sd, cfg, data, dtype=torch.float64)
    want = orc.extract_matches(o64["scores"].float(), MATCH_THRESHOLD)
    amb_r, amb_c = orc.ambiguous_rows(o64["scores"], 1e-4)
    diff = got_matches0 != want["matches0"]
    # a row may legitimately differ if it, or the column it points to, is a near-tie, or its score sits at the threshold
    ms = want["matching_scores0"]
    near_thr = (ms - MATCH_THRESHOLD).abs() < 1e-3
    unexplained = diff & ~amb_r & ~near_thr
    n_bad = 0
    for b, i in torch.nonzero(unexplained).tolist():
        j = int(want["_row_argmax"][b, i])
        if not amb_c[b, j]:
            n_bad += 1
    ndiff = int(diff.sum())
    ceiling = max(2, int(math.ceil(MAX_EXEMPT_FRAC * diff.numel())))
    assert ndiff <= ceiling, f"{ndiff} of {diff.numel()} rows differ from the oracle: over the exemption ceiling of {ceiling} even if all are near-ties"
    return ndiff, n_bad, o64


@pytest.mark.parametrize("name", ["c1", "mid", "flags", "nodesc", "siren", "linear", "favor", "d128"])
def test_forward_against_reference_fixture(gpu_device, name):
    z, cfg, sd, data = load_case(name)
    model = _build(cfg, sd, gpu_device)
    out = model.match(to_device(data, gpu_device), MATCH_THRESHOLD)
    out = {k: v.cpu() for k, v in out.items()}
    err = np.abs(out["scores"].numpy() - z["scores"]).max()
    assert err < TOL_SCORES, f"{name}: scores max abs err {err}"
    for k in ("context_descriptors0", "context_descriptors1"):
        assert out[k].shape == z[k].shape
        assert np.abs(out[k].numpy() - z[k]).max() < TOL_SCORES, k
    ndiff, unexplained, _ = _index_agreement(out["matches0"], out["scores"], sd, cfg, data)
    assert unexplained == 0, f"{name}: {ndiff} rows differ, {unexplained} not explained by near-ties"
    same = out["matches0"].numpy() == z["matches0"]
    parity_note(f"[{name}] scores err {err:.2e}; matches0 identical on {same.mean() * 100:.2f}% rows exempt={ndiff}")
    # extraction itself is exact given the GPU's own scores
    want = orc.extract_matches(out["scores"], MATCH_THRESHOLD)
    assert torch.equal(out["matches0"], want["matches0"]) and torch.equal(out["matches1"], want["matches1"])
    # forward() returns exactly the reference's three keys
    out2 = model(to_device(data, gpu_device))
    assert set(out2) == {"context_descriptors0", "context_descriptors1", "scores"}
    assert torch.equal(out2["scores"].cpu(), out["scores"])


def test_forward_c2_shape_against_reference_fixture(gpu_device):
    z, cfg, sd, data = load_case("c2")
    model = _build(cfg, sd, gpu_device)
    out = {k: v.cpu() for k, v in model.match(to_device(data, gpu_device), MATCH_THRESHOLD).items()}
    s = out["scores"]
    errs = [np.abs(s[:, ::8, ::8].numpy() - z["scores_sub8"]).max(), np.abs(s[:, -1, :].numpy() - z["scores_lastrow"]).max(),
            np.abs(s[:, :, -1].numpy() - z["scores_lastcol"]).max()]
    assert max(errs) < TOL_SCORES, errs
    assert np.abs(s.double().sum(2).numpy() - z["row_sums64"]).max() < 1025 * TOL_SCORES
    assert np.abs(out["context_descriptors0"][:, ::4, ::16].numpy() - z["context_descriptors0_sub"]).max() < TOL_SCORES
    same = (out["matches0"].numpy() == z["matches0"])
    print(f"[c2] scores err {max(errs):.2e}; matches0 identical on {same.mean() * 100:.3f}% of rows")
    ndiff, unexplained, o64 = _index_agreement(out["matches0"], s, sd, cfg, data)
    assert unexplained == 0, (ndiff, unexplained)
    assert (s.double() - o64["scores"]).abs().max() < TOL_SCORES


@pytest.mark.parametrize("name", ["c3", "c4"])
def test_forward_large_shapes_against_reference_fixture(gpu_device, name):
    """VERDICT r4 missing 5: the LARGE BASELINE shapes against fixtures produced by the reference itself (c3: 2048 x 2048 x 256-d, c4: 4096 x 4096 x
    128-d with s = 6 -- the 128-d kernel family at BASELINE configs[3]; 9 stages, 100 iterations, 2 pairs): sub-sampled scores, dustbin row and
    column, float64 row sums, context descriptors, matches0.  Index differences are allowed only on rows the reference's OWN scores mark as
    near-ties (stored top-1 / top-2 gaps), and are bounded."""
    z, cfg, sd, data = load_case(name)
    model = _build(cfg, sd, gpu_device)
    out = {k: v.cpu() for k, v in model.match(to_device(data, gpu_device), MATCH_THRESHOLD).items()}
    model.check_status()
    s = out["scores"]
    n = s.shape[2] - 1
    errs = [np.abs(s[:, ::8, ::8].numpy() - z["scores_sub8"]).max(), np.abs(s[:, -1, :].numpy() - z["scores_lastrow"]).max(),
            np.abs(s[:, :, -1].numpy() - z["scores_lastcol"]).max()]
    assert max(errs) < TOL_SCORES, errs
    assert np.abs(s.double().sum(2).numpy() - z["row_sums64"]).max() < (n + 1) * TOL_SCORES
    assert np.abs(out["context_descriptors0"][:, ::4, ::16].numpy() - z["context_descriptors0_sub"]).max() < TOL_SCORES
    assert np.abs(out["context_descriptors1"][:, ::4, ::16].numpy() - z["context_descriptors1_sub"]).max() < TOL_SCORES
    diff = out["matches0"].numpy() != z["matches0"]
    unexplained = 0
    for b, i in zip(*np.nonzero(diff)):
        if not (z["row_gap"][b, i] < 2e-4 or z["col_gap"][b, z["row_argmax"][b, i]] < 2e-4 or abs(float(z["matching_scores0"][b, i]) - MATCH_THRESHOLD) < 1e-3):
            unexplained += 1
    assert unexplained == 0 and diff.sum() <= max(2, int(math.ceil(MAX_EXEMPT_FRAC * diff.size))), (int(diff.sum()), unexplained)
    parity_note(f"[{name} vs reference fixture] scores err {max(errs):.2e}; matches0 identical on {(~diff).mean() * 100:.3f}% rows exempt={int(diff.sum())}")
    want = orc.extract_matches(s, MATCH_THRESHOLD)          # extraction itself is exact given the GPU's own scores
    assert torch.equal(out["matches0"], want["matches0"])


def test_image_tensor_path_equals_size_path(gpu_device):
    z, cfg, sd, data = load_case("c1")
    model = _build(cfg, sd, gpu_device)
    a = model(to_device(data, gpu_device))["scores"]
    d2 = {k: v for k, v in to_device(data, gpu_device).items() if not k.endswith("_size")}
    d2["image0"] = torch.empty(1, 1, syn.IMAGE_WH[1], syn.IMAGE_WH[0], device=gpu_device)
    d2["image1"] = torch.empty(1, 1, syn.IMAGE_WH[1], syn.IMAGE_WH[0], device=gpu_device)
    assert torch.equal(a, model(d2)["scores"])


def test_full_size_c2_batch_properties(gpu_device):
    """BASELINE configs[1] at full size (B=32, 1024 kpts, 256-d, 9 stages, 100 iters): size-independent
    properties + a per-pair spot check against the oracle."""
    kw = {k: v for k, v in syn.CONFIGS["C2"].items() if k not in ("kpts", "batch")}
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    B, m, n = 32, 1024, 1024
    data = syn.make_batch(B, m, n, 256, 1, seed=11)
    out = model.match(to_device(data, gpu_device), MATCH_THRESHOLD)
    model.check_status()                 # B=32 x 1024 x 1024 takes the on-chip-resident Sinkhorn schedule: no peer time-out
    s = out["scores"]
    assert torch.isfinite(s).all()
    norm = -math.log(m + n)
    # (1) column marginals exact after the last v update; row marginals close after 100 iterations
    col = torch.logsumexp(s.double() + norm, dim=1)
    lb = torch.full((n + 1,), norm, dtype=torch.float64, device=s.device); lb[-1] += math.log(m)
    assert (col - lb).abs().max() < 1e-4
    row = torch.logsumexp(s.double() + norm, dim=2)
    la = torch.full((m + 1,), norm, dtype=torch.float64, device=s.device); la[-1] += math.log(n)
    assert (row - la).abs().max() < 1e-2
    # (2) pairs are independent: pair 5 alone gives the same scores as inside the batch (eval-mode BN)
    one = {k: (v[5:6] if torch.is_tensor(v) else v) for k, v in data.items()}
    s1 = model(to_device(one, gpu_device))["scores"]
    # B=32 runs the resident Sinkhorn schedule and the 256x256 GEMM tile, B=1 the streaming schedule and
    # a smaller tile: same arithmetic, different reduction orders, so rounding noise only (scores ~ -26,
    # 100 iterations); both sit inside TOL_SCORES of the oracle, see (5)
    assert (s1[0] - s[5]).abs().max() < 2e-4
    # with the schedule pinned the only batch-size dependence left is the kernel choice: B = 1 runs the 32-token small-batch kernels for the
    # message MLP and the q / k / v projections (mlp_small_kernel, proj_small_kernel: other summation orders), B = 32 the 128 / 256-token tiles
    import os
    prev = os.environ.get("OG_SINKHORN_RESIDENT")
    os.environ["OG_SINKHORN_RESIDENT"] = "0"
    try:
        sb = model(to_device(data, gpu_device))["scores"]
        sa = model(to_device(one, gpu_device))["scores"]
    finally:
        if prev is None:
            os.environ.pop("OG_SINKHORN_RESIDENT")
        else:
            os.environ["OG_SINKHORN_RESIDENT"] = prev
    assert (sa[0] - sb[5]).abs().max() < 1e-4
    # (3) permuting the keypoints of image 1 permutes the columns of scores
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
    dp = dict(one)
    for k in ("keypoints1", "local_descriptors1", "side_info1"):
        dp[k] = one[k][:, perm]
    sp = model(to_device(dp, gpu_device))["scores"]
    assert (sp[0, :, :-1] - s1[0, :, :-1][:, perm.to(s1.device)]).abs().max() < 2e-4
    # (4) mutual matches are a partial bijection and respect the threshold
    m0, m1, ms0 = out["matches0"], out["matches1"], out["matching_scores0"]
    valid = m0 >= 0
    assert (ms0[valid] > MATCH_THRESHOLD).all()
    bi = torch.arange(B, device=s.device)[:, None].expand_as(m0)[valid]
    assert (m1[bi, m0[valid]] == torch.nonzero(valid)[:, 1]).all()
    assert int(valid.sum()) == int((m1 >= 0).sum())
    # (5) EVERY pair of the batch against the float64 oracle: scores within the bar, matches0 identical up to bounded near-ties.
    #     (This is the call that takes the resident Sinkhorn schedule, the 256 x 256 GEMM tiles and the fused message MLP.)
    s_cpu, m0_cpu = s.cpu(), out["matches0"].cpu()
    worst, exempt = 0.0, 0
    for p0 in range(0, B, 8):
        chunk = {k: (v[p0:p0 + 8] if torch.is_tensor(v) else v) for k, v in data.items()}
        ndiff, bad, o64 = _index_agreement(m0_cpu[p0:p0 + 8], s_cpu[p0:p0 + 8], sd, cfg, chunk)
        assert bad == 0, f"pairs {p0}..{p0 + 7}: {ndiff} rows differ, {bad} not explained by float64 near-ties"
        worst = max(worst, (s_cpu[p0:p0 + 8].double() - o64["scores"]).abs().max().item())
        exempt += ndiff
    parity_note(f"[C2 B=32] all 32 pairs: scores err {worst:.2e} exempt={exempt} of {B * m} rows")
    assert worst < TOL_SCORES


def test_ragged_packed_wide_range(gpu_device):
    """Packed ragged kernels on a spread of sizes that crosses the Sinkhorn / attention tile geometries
    (1..2 column parts, partial last key tile, m < 32), against the uniform path run per pair."""
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=7, side_info_size=1)
    sd = syn.make_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    lens = [(5, 1100), (300, 40), (1025, 1030), (64, 64), (130, 257), (1, 1), (777, 512)]
    pairs = []
    for i, (m, n) in enumerate(lens):
        p = to_device(syn.make_pair(m, n, 64, 1, seed=300 + i), gpu_device)
        p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
        pairs.append(p)
    packed = model.match_ragged(pairs, MATCH_THRESHOLD)
    for p, q, (m, n) in zip(pairs, packed, lens):
        one = {k: (v[None] if torch.is_tensor(v) else v) for k, v in p.items()}
        ref = model.match(one, MATCH_THRESHOLD)
        assert q["scores"].shape == (m + 1, n + 1)
        assert (q["scores"] - ref["scores"][0]).abs().max() < 1e-4, (m, n)
        assert torch.equal(q["matches0"], ref["matches0"][0]) and torch.equal(q["matches1"], ref["matches1"][0]), (m, n)


def test_ragged_pairs_equal_per_pair_oracle(gpu_device):
    """BASELINE configs[4] semantics at small scale: every pair has its own (m, n); the result must equal the
    per-pair (B=1) oracle, whatever the bucketing."""
    from tests.ragged_bucketing import bucket_by_shape, match_ragged
    cfg = syn.make_config(descriptor_dim=128, num_stages=2, num_heads=4, num_iters=10, side_info_size=1)
    sd = syn.make_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    lens = [(70, 91), (64, 64), (70, 91), (129, 33), (64, 64)]
    pairs_cpu = []
    for i, (m, n) in enumerate(lens):
        p = syn.make_pair(m, n, 128, 1, seed=100 + i)
        p["image0_size"] = list(syn.IMAGE_WH); p["image1_size"] = list(syn.IMAGE_WH)
        pairs_cpu.append(p)
    assert [len(v) for v in bucket_by_shape(pairs_cpu).values()] == [2, 2, 1]
    dev_pairs = [to_device(p, gpu_device) for p in pairs_cpu]
    res = match_ragged(model, dev_pairs, MATCH_THRESHOLD)
    packed = model.match_ragged(dev_pairs, MATCH_THRESHOLD)          # token-packed kernels (og_forward_ragged)
    for r, q in zip(res, packed):
        assert (r["scores"] - q["scores"]).abs().max() < 1e-4
        assert torch.equal(r["matches0"], q["matches0"]) and torch.equal(r["matches1"], q["matches1"])
    for p, r, (m, n) in zip(pairs_cpu, packed, lens):
        one = {k: (v[None] if torch.is_tensor(v) else v) for k, v in p.items()}
        with torch.no_grad():
            ref = orc.match_pairs(sd, cfg, one, MATCH_THRESHOLD)
        assert r["scores"].shape == (m + 1, n + 1)
        assert (r["scores"].cpu() - ref["scores"][0]).abs().max() < TOL_SCORES
        assert torch.equal(r["matches0"].cpu(), ref["matches0"][0])


# ----------------------------------------------------------------------------- steps either side of the path (f1, f3)
@pytest.mark.parametrize("method", ["none", "scale", "rotation", "scale_rotation", "affine"])
def test_prepare_features_output(gpu_device, method):
    from openglue_amd import features
    g = torch.Generator().manual_seed(len(method))
    B, N, D = 2, 333, 32
    A = torch.randn(B, N, 2, 2, generator=g) + 2.0 * torch.eye(2)          # affine shapes, some with negative determinant
    lafs = torch.cat([A, torch.rand(B, N, 2, 1, generator=g) * 900], dim=-1)
    resp = torch.rand(B, N, generator=g)
    desc = torch.randn(B, N, D, generator=g)
    for log_response in (False, True):
        want = orc.prepare_features_output(lafs, resp, desc, method, log_response=log_response)
        got = features.prepare_features_output(lafs.to(gpu_device), resp.to(gpu_device), desc.to(gpu_device), method,
                                               log_response=log_response)
        assert got["side_info"].shape[-1] == features.side_info_size(method)
        assert torch.equal(got["keypoints"].cpu(), want["keypoints"])
        # thin frames (det ~ 0) make 1/scale ill-conditioned: judge both fp32 results against float64
        truth = orc.prepare_features_output(lafs.double(), resp.double(), desc, method, log_response=log_response)["side_info"]
        err_gpu = (got["side_info"].cpu().double() - truth).abs()
        err_cpu = (want["side_info"].double() - truth).abs()
        assert (err_gpu <= 4.0 * err_cpu + 2e-6 * truth.abs() + 1e-6).all()


def test_prepare_features_against_reference_laf_fixture(gpu_device):
    """og_prepare_features against tests/golden/laf.npz: the reference's prepare_features_output + LAF converters executed unchanged
    (make_golden_laf.py; kornia's get_laf_scale is the one restated stub).  Thin frames make 1/scale ill-conditioned in fp32: the
    bound is relative to the scale's own rounding (a few ulp of the determinant)."""
    from openglue_amd import features
    z = np.load(os.path.join(GOLDEN, "laf.npz"))
    lafs, resp, desc = torch.from_numpy(z["lafs"]), torch.from_numpy(z["responses"]), torch.from_numpy(z["desc"])
    A = lafs[..., :2].double()
    det = (A[..., 0, 0] * A[..., 1, 1] - A[..., 1, 0] * A[..., 0, 1]).abs()
    cond = ((A[..., 0, 0] * A[..., 1, 1]).abs() + (A[..., 1, 0] * A[..., 0, 1]).abs()) / det.clamp_min(1e-30)     # cancellation in the determinant
    for method in ("none", "scale", "rotation", "scale_rotation", "affine"):
        for lr in (0, 1):
            got = features.prepare_features_output(lafs.to(gpu_device), resp.to(gpu_device), desc.to(gpu_device), method, log_response=bool(lr))
            assert np.array_equal(got["keypoints"].cpu().numpy(), z[f"{method}_{lr}_keypoints"])
            ref = torch.from_numpy(z[f"{method}_{lr}_side_info"]).double()
            err = (got["side_info"].cpu().double() - ref).abs()
            tol = (4e-7 * cond).unsqueeze(-1) * (ref.abs() + 1.0) + 2e-6
            assert (err <= tol).all(), (method, lr, float((err / tol).max()))
    with pytest.raises(NameError):
        features.prepare_features_output(lafs.to(gpu_device), resp.to(gpu_device), desc.to(gpu_device), "bogus")


def test_compact_matches(gpu_device):
    from openglue_amd import features
    g = torch.Generator().manual_seed(3)
    B, M, N = 3, 700, 650
    m0 = torch.randint(0, N, (B, M), generator=g)
    m0[torch.rand(B, M, generator=g) < 0.45] = -1
    m0[1] = -1                                                   # a pair without any match
    ms0 = torch.rand(B, M, generator=g)
    lafs0, lafs1 = torch.randn(B, M, 2, 3, generator=g), torch.randn(B, N, 2, 3, generator=g)
    want = orc.compact_matches(m0, ms0, lafs0, lafs1)
    got = features.compact_matches(m0.to(gpu_device), ms0.to(gpu_device), lafs0.to(gpu_device), lafs1.to(gpu_device))
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k].cpu(), want[k]), k
    none = features.compact_matches(torch.full((2, 5), -1, device=gpu_device), torch.zeros(2, 5, device=gpu_device))
    assert none["confidence"].numel() == 0


def test_openglue_matcher_pipeline(gpu_device):
    """examples/openglue_matcher.py: OpenGlueMatcher (inference.py:83-209 with pre-extracted features): LAFs / responses / descriptors of
    two images -> compacted matches, against the same chain restated in the oracle (prepare_features_output -> SuperGlue.forward ->
    mutual-NN extraction -> boolean-mask compaction).  SIFT-like config: `affine` side info (6 channels), log-transformed response."""
    from examples.openglue_matcher import OpenGlueMatcher
    B, m, n, D = 2, 180, 150, 128
    cfg = syn.make_config(descriptor_dim=D, num_stages=2, num_heads=4, num_iters=12, side_info_size=6)
    cfg["laf_to_sideinfo_method"] = "affine"; cfg["log_transform_response"] = True
    sd = syn.make_state_dict(cfg, seed=4)
    model = _build(cfg, sd, gpu_device)
    base = syn.make_batch(B, m, n, D, 6, seed=21)                     # matched keypoints / descriptors; LAFs are built around the keypoints
    g = torch.Generator().manual_seed(5)

    def lafs_of(k):
        A = 0.3 * torch.randn(*k.shape[:2], 2, 2, generator=g) + 3.0 * torch.eye(2)
        return torch.cat([A, k[..., None]], dim=-1)
    lafs0, lafs1 = lafs_of(base["keypoints0"]), lafs_of(base["keypoints1"])
    resp0, resp1 = torch.rand(B, m, generator=g), torch.rand(B, n, generator=g)
    data = {"lafs0": lafs0, "lafs1": lafs1, "responses0": resp0, "responses1": resp1,
            "descriptors0": base["local_descriptors0"], "descriptors1": base["local_descriptors1"],
            "image0": torch.empty(B, 1, 720, 960), "image1": torch.empty(B, 1, 600, 800)}
    mc = {"superglue": cfg, "inference": {"match_threshold": MATCH_THRESHOLD}}
    pipe = OpenGlueMatcher(None, model, mc)
    got = pipe(to_device(data, gpu_device))
    # the oracle chain (inference.py:141-209)
    f0 = orc.prepare_features_output(lafs0, resp0, data["descriptors0"], "affine", log_response=True)
    f1 = orc.prepare_features_output(lafs1, resp1, data["descriptors1"], "affine", log_response=True)
    od = {"keypoints0": f0["keypoints"], "keypoints1": f1["keypoints"], "local_descriptors0": f0["local_descriptors"],
          "local_descriptors1": f1["local_descriptors"], "side_info0": f0["side_info"], "side_info1": f1["side_info"],
          "image0_size": [960, 720], "image1_size": [800, 600]}
    with torch.no_grad():
        ref = orc.match_pairs(sd, cfg, od, MATCH_THRESHOLD)
    ndiff, unexplained, _ = _index_agreement(model.match(to_device(od, gpu_device), MATCH_THRESHOLD)["matches0"].cpu(), None, sd, cfg, od)
    assert unexplained == 0
    if ndiff == 0:
        want = orc.compact_matches(ref["matches0"], ref["matching_scores0"], lafs0, lafs1)
        assert set(got) == set(want) and want["confidence"].numel() > 20
        for k in ("original_matching_idxs", "batch_indexes", "lafs0", "lafs1", "keypoints0", "keypoints1"):
            assert torch.equal(got[k].cpu(), want[k]), k
        assert (got["confidence"].cpu() - want["confidence"]).abs().max() < 1e-3
    with pytest.raises(RuntimeError, match="no local feature extractor"):
        pipe({"image0": data["image0"], "image1": data["image1"]})


def test_hipgraph_replay_equals_eager(gpu_device):
    """The whole launch sequence captured into a hipGraph (launch-bound small shapes) gives identical results."""
    from examples.hipgraph_replay import GraphedMatcher
    z, cfg, sd, data = load_case("c1")
    model = _build(cfg, sd, gpu_device)
    d0 = to_device(data, gpu_device)
    eager = {k: v.clone() for k, v in model.match(d0, MATCH_THRESHOLD).items()}
    gm = GraphedMatcher(model, d0, MATCH_THRESHOLD)
    out = gm(d0)
    for k in eager:
        assert torch.equal(out[k], eager[k]), k
    # new inputs through the same graph
    d1 = to_device(syn.make_batch(1, 64, 64, 64, 1, seed=77), gpu_device)
    want = {k: v.clone() for k, v in model.match(d1, MATCH_THRESHOLD).items()}
    out = gm(d1)
    for k in want:
        assert torch.equal(out[k], want[k]), k


@pytest.mark.parametrize("m,n,kw", [
    (1, 1, dict(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=2)),
    (3, 200, dict(descriptor_dim=64, num_stages=2, num_heads=2, num_iters=4, side_info_size=3)),
    (129, 2, dict(descriptor_dim=128, num_stages=1, num_heads=8, num_iters=5, side_info_size=0 + 1)),
    (70, 90, dict(descriptor_dim=512, num_stages=1, num_heads=8, num_iters=5, hidden_layers_sizes=(64, 256))),
    (40, 40, dict(descriptor_dim=256, num_stages=0, num_heads=4, num_iters=3)),
])
def test_forward_edge_shapes(gpu_device, m, n, kw):
    """Degenerate and unusual shapes through the whole path: single keypoints, strongly non-square pairs, 8 heads,
    512-d descriptors with a non-default encoder, zero GNN stages."""
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=1)
    model = _build(cfg, sd, gpu_device)
    data = syn.make_batch(2, m, n, cfg["descriptor_dim"], cfg["positional_encoding"]["side_info_size"], seed=5)
    out = {k: v.cpu() for k, v in model.match(to_device(data, gpu_device), MATCH_THRESHOLD).items()}
    with torch.no_grad():
        ref = orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
    err = (out["scores"] - ref["scores"]).abs().max().item()
    assert err < TOL_SCORES, err
    assert (out["context_descriptors0"] - ref["context_descriptors0"]).abs().max() < TOL_SCORES
    amb_r, _ = orc.ambiguous_rows(ref["scores"].double(), 1e-3) if m > 1 and n > 1 else (torch.zeros(2, m, dtype=torch.bool), None)
    diff = (out["matches0"] != ref["matches0"]) & ~amb_r
    assert int(diff.sum()) == 0


@pytest.mark.parametrize("D,m,n,B", [(256, 300, 170, 2), (64, 17, 40, 3), (192, 2, 33, 1)])
def test_favor_relu_attention_whole_path(gpu_device, D, m, n, B):
    """attention = 'favor_relu' (one head, 2D ReLU random features; attention.py:43-95) at the widest supported size (D = 256:
    512 features, 8 column slices per problem, two features per thread), at a narrow one and with two queries -- against the
    oracle, which tests/golden/favor.npz pins to the reference; uniform and token-packed (ragged) launches."""
    cfg = syn.make_config(descriptor_dim=D, num_stages=2, num_heads=1, num_iters=6, side_info_size=1, attention="favor_relu")
    sd = syn.make_state_dict(cfg, seed=2)
    model = _build(cfg, sd, gpu_device)
    data = syn.make_batch(B, m, n, D, 1, seed=11)
    out = {k: v.cpu() for k, v in model.match(to_device(data, gpu_device), MATCH_THRESHOLD).items()}
    with torch.no_grad():
        ref = orc.match_pairs(sd, cfg, data, MATCH_THRESHOLD)
    err = (out["scores"] - ref["scores"]).abs().max().item()
    print(f"[favor_relu D={D} {m}x{n}] scores err {err:.2e}")
    assert err < TOL_SCORES, err
    assert (out["context_descriptors0"] - ref["context_descriptors0"]).abs().max() < TOL_SCORES
    ndiff, unexplained, _ = _index_agreement(out["matches0"], out["scores"], sd, cfg, data)
    assert unexplained == 0, (ndiff, unexplained)
    # ragged launch of the same pairs (per-pair row ranges from the descriptor) = the uniform result
    pairs = []
    for b in range(B):
        p = {k: v[b] for k, v in data.items() if torch.is_tensor(v)}
        p["image0_size"] = data["image0_size"]; p["image1_size"] = data["image1_size"]
        pairs.append(to_device(p, gpu_device))
    for b, r in enumerate(model.match_ragged(pairs, MATCH_THRESHOLD)):
        assert (r["scores"].cpu() - out["scores"][b]).abs().max() < 1e-4
        assert torch.equal(r["matches0"].cpu(), out["matches0"][b])


# ----------------------------------------------------------------------------- on-chip-resident Sinkhorn iterations
@pytest.mark.parametrize("B,m,n,iters,reg", [(3, 37, 53, 7, 1.0), (2, 64, 64, 2, 1.0), (1, 130, 1023, 20, 0.7), (2, 257, 1000, 10, 1.0),
                                             (4, 128, 1024, 30, 1.0), (1, 1, 1, 3, 1.0), (2, 300, 17, 6, 2.0), (8, 1024, 1024, 100, 1.0),
                                             (1, 1024, 512, 100, 1.0),
                                             # wider than one wave tile: 2 / 4 waves per row (n <= 2048 / 4096), two-hop column exchange
                                             (2, 300, 2047, 12, 1.0), (1, 77, 4096, 8, 1.0), (3, 1100, 1500, 25, 0.8), (2, 2048, 2048, 100, 1.0),
                                             (1, 4096, 4096, 30, 1.0), (1, 3000, 2500, 15, 1.0),
                                             # pairs as 2-D grids of tiles (column blocks on different XCDs, a row hop per iteration): 2 x 24 tiles of
                                             # 64 x 2048, 2 x 20 tiles of 128 x 1024
                                             (1, 1500, 3000, 12, 1.0), (2, 2500, 1800, 10, 1.0),
                                             # more pairs than one launch holds: rounds of co-resident pairs (G = 8: 32 per launch; G = 32: 8)
                                             (40, 1024, 200, 20, 1.0), (11, 2048, 1030, 10, 1.0)])
def test_sinkhorn_resident_vs_oracle(gpu_device, monkeypatch, B, m, n, iters, reg):
    """sinkhorn_resident.hip (plan entries held in registers + LDS, iterations 2..iters in one launch per round of co-resident
    pairs, column sums exchanged between the workgroups of a pair through {epoch, value} granules) against the float64 oracle AND
    against the streaming kernels: partial row blocks (m % 16, m % 128), masked columns, several workgroups per pair, rows that
    cross 2 or 4 waves (n > 1024), several rounds (B > pairs per launch), 100 iterations."""
    g = torch.Generator().manual_seed(m * 31 + n)
    S = _rand(g, B, m, n, scale=4.0)
    ref = _sinkhorn_ref(S, 0.7, iters, reg)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    out, status = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg, return_status=True)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "0")
    stream = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg).cpu()
    out = out.cpu()
    assert status == 0, "a cross-workgroup wait timed out"
    err = (out.double() - ref).abs().max().item()
    d = (out - stream).abs().max().item()
    print(f"[sinkhorn resident {B}x{m}x{n} it={iters}] err vs float64 {err:.2e}; vs streaming kernels {d:.2e}")
    assert err < 1e-4, err
    assert d < 5e-5, d
    norm = -math.log(m + n)
    lb = torch.full((n + 1,), norm, dtype=torch.float64); lb[-1] += math.log(m)
    assert (torch.logsumexp(out.double() + norm, dim=1) - lb).abs().max() < 1e-4        # column marginals exact after the last v update


@pytest.mark.parametrize("B,m,n,iters,reg", [(1, 1024, 1024, 100, 1.0), (8, 1024, 1024, 30, 1.0), (3, 777, 1000, 25, 0.8), (2, 50, 1024, 12, 1.0), (1, 1000, 513, 20, 1.0),
                                             (1, 1, 1, 3, 1.0), (4, 33, 7, 6, 2.0), (1, 1023, 1, 5, 1.0), (5, 640, 333, 25, 1.0), (12, 1024, 1024, 20, 1.0), (16, 1000, 700, 15, 1.0)])
def test_sinkhorn_resident_few_pairs_geometry(gpu_device, monkeypatch, B, m, n, iters, reg):
    """Round 5: launches of few pairs of <= 1024 x 1024 keypoints (the reference's inference.py regime: ONE pair per call) take 4 rows per wave
    instead of 16 -- a pair is 32 workgroup tiles of 32 x 1024 instead of 8 of 128 x 1024 (sinkhorn_resident_kernel<1, ., 4>); 9 to 16 pairs take 8 rows
    per wave (16 tiles of 64 x 1024).  All three geometries (OG_SINKHORN_FEW = 1 / 8 / 0) against the float64 oracle and against each other; partial last row blocks, single rows / columns, 8 pairs = 256
    workgroups."""
    g = torch.Generator().manual_seed(m * 37 + n + B)
    S = _rand(g, B, m, n, scale=4.0)
    ref = _sinkhorn_ref(S, 0.7, iters, reg)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    outs = {}
    for few in ("1", "8", "0"):                        # the finest geometry that exists (4, else 8 rows per wave), 8 rows per wave, 16
        monkeypatch.setenv("OG_SINKHORN_FEW", few)
        out, status = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg, return_status=True)
        assert status == 0, f"few={few}: a cross-workgroup wait timed out"
        outs[few] = out.cpu()
        err = (outs[few].double() - ref).abs().max().item()
        assert err < 1e-4, (few, err)
    d = max((outs["1"] - outs["0"]).abs().max().item(), (outs["8"] - outs["0"]).abs().max().item())
    print(f"[sinkhorn few-pairs geometry {B}x{m}x{n} it={iters}] 4 / 8 rows per wave vs 16: {d:.2e}")
    assert d < 5e-5, d
    monkeypatch.setenv("OG_SINKHORN_FEW", "1")
    again = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg).cpu()
    assert torch.equal(again, outs["1"])               # fixed summation orders: bit-identical from call to call


@pytest.mark.parametrize("B,m,n,iters", [(1, 2, 257, 2), (2, 711, 2027, 2), (5, 65, 2128, 8), (5, 127, 511, 12), (5, 1025, 3071, 2), (1, 1024, 2992, 8),
                                         (9, 1382, 3071, 3), (2, 3071, 257, 3), (1, 2663, 762, 5), (9, 2427, 65, 8), (3, 257, 438, 12), (1, 15, 2048, 5),
                                         (5, 1915, 33, 3), (1, 1224, 46, 12), (1, 1023, 1024, 8), (1, 17, 190, 12), (1, 127, 2047, 5), (9, 257, 4095, 5),
                                         (3, 4096, 3000, 3), (2, 3100, 4096, 3)])
def test_sinkhorn_resident_edge_shapes(gpu_device, monkeypatch, B, m, n, iters):
    """A fixed pseudo-random draw of shapes around the tile edges (16 rows per wave, 128 / 64 / 32 rows per workgroup, 1024 columns per wave
    tile, 2 and 4 column blocks, one row, one column block almost empty, several rounds): every geometry rs_geom can produce, against the
    float64 oracle."""
    g = torch.Generator().manual_seed(B * 1000003 + m * 4099 + n)
    S = _rand(g, B, m, n, scale=4.0)
    ref = _sinkhorn_ref(S, 0.3, iters, 0.9)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    out, status = ops.sinkhorn(S.to(gpu_device), 0.3, iters, 0.9, return_status=True)
    assert status == 0
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 1e-4, err


@pytest.mark.parametrize("B,m,n,iters", [(8, 1024, 1024, 60), (2, 257, 1000, 10), (3, 640, 333, 25), (8, 2048, 2048, 40), (5, 700, 1800, 20), (1, 2200, 2100, 8)])
def test_sinkhorn_resident_exchange_scopes(gpu_device, monkeypatch, B, m, n, iters):
    """The column partials travel between the workgroups of a pair either at agent scope or -- when the kernel finds all of them
    on one XCD (B a multiple of 8 with the round-robin dispatch) -- through that XCD's L2 with workgroup-scope streaming loads.
    Same arithmetic, same summation order: the two paths must agree bit for bit, and neither may time out."""
    g = torch.Generator().manual_seed(B * 7 + m)
    S = _rand(g, B, m, n, scale=4.0).to(gpu_device)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    monkeypatch.setenv("OG_SINKHORN_AGENT_SCOPE", "0")
    a, st_a = ops.sinkhorn(S, 0.7, iters, 1.0, return_status=True)
    monkeypatch.setenv("OG_SINKHORN_AGENT_SCOPE", "1")
    b2, st_b = ops.sinkhorn(S, 0.7, iters, 1.0, return_status=True)
    assert st_a == 0 and st_b == 0
    assert torch.equal(a, b2)


def test_sinkhorn_resident_extreme_range_and_repeatability(gpu_device, monkeypatch):
    g = torch.Generator().manual_seed(5)
    B, m, n, iters = 2, 200, 900, 40
    S = _rand(g, B, m, n, scale=25.0)
    S[0, 5, :] = -100.0
    S[1, :, 7] = 100.0
    ref = _sinkhorn_ref(S, 0.7, iters, 0.5)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    a, st = ops.sinkhorn(S.to(gpu_device), 0.7, iters, 0.5, return_status=True)
    b2 = ops.sinkhorn(S.to(gpu_device), 0.7, iters, 0.5)
    assert st == 0 and bool(torch.isfinite(a).all())
    assert torch.equal(a, b2)                       # fixed summation orders everywhere: bit-identical from call to call
    tol = 1e-4 + 2e-6 * ref.abs().max().item()
    assert (a.cpu().double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("scale,reg,z", [(25.0, 0.5, 0.7), (8.0, 0.1, -30.0), (60.0, 1.0, 50.0), (1e-3, 1.0, 0.0)])
def test_sinkhorn_resident_refresh_on_extreme_range(gpu_device, monkeypatch, scale, reg, z):
    """The resident state is the plan matrix in the LINEAR domain; entries below 2^-126 are lost until the workgroup re-evaluates its
    rows from the scores, which it does when the duals have moved by more than 40 bits since the last evaluation
    (sinkhorn_resident.hip, RS_DRIFT_BITS).  On these inputs (|S/reg| of several hundred, duals moving by 100-180 bits after the first
    iteration) a solver WITHOUT the refresh is off by 30-60 in the log-scores (tests/emulate_sinkhorn_linear.py)."""
    g = torch.Generator().manual_seed(int(scale * 10) + 3)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    # the second shape: 11 workgroups per pair, rows over two waves -- the workgroups of a pair refresh at DIFFERENT iterations (each
    # bounds the drift of its own rows), so what they exchange must not depend on when a workgroup last evaluated its entries
    for (B, m, n, iters) in [(2, 96, 200, 30), (1, 700, 1500, 30)]:
        S = _rand(g, B, m, n, scale=scale)
        S[0, 5, :] = -4.0 * scale
        S[B - 1, :, 7] = 4.0 * scale
        ref = _sinkhorn_ref(S, z, iters, reg)
        out, status = ops.sinkhorn(S.to(gpu_device), z, iters, reg, return_status=True)
        out = out.cpu()
        assert status == 0 and bool(torch.isfinite(out).all())
        err = (out.double() - ref).abs().max().item()
        tol = 1e-4 + 2e-6 * ref.abs().max().item()
        print(f"[sinkhorn resident extreme {B}x{m}x{n} scale={scale} reg={reg} z={z}] max err {err:.2e} (tol {tol:.1e}), max |score| {ref.abs().max().item():.0f}")
        assert err <= tol


def test_sinkhorn_resident_timeout_is_survivable(gpu_device, monkeypatch):
    """A time-out of the on-chip-resident kernel (peer workgroups not co-resident: another stream / process holds CUs) must not
    hand back garbage: the safety-net kernel enqueued behind it recomputes the whole solve, one workgroup per pair
    (optimal_transport.py:20-28 from u = v = 0), and the status says so.  OG_SINKHORN_FORCE_TIMEOUT=1 makes the resident launch
    behave as if it had timed out."""
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    g = torch.Generator().manual_seed(77)
    for (B, m, n, iters, reg) in [(3, 300, 257, 12, 1.0), (2, 129, 1000, 25, 0.7)]:
        S = _rand(g, B, m, n, scale=2.0)
        ref = _sinkhorn_ref(S, 0.7, iters, reg)
        monkeypatch.setenv("OG_SINKHORN_FORCE_TIMEOUT", "1")
        out, status = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg, return_status=True)
        assert status == 2
        assert (out.cpu().double() - ref).abs().max() < 1e-4
        monkeypatch.delenv("OG_SINKHORN_FORCE_TIMEOUT")
        out2, status2 = ops.sinkhorn(S.to(gpu_device), 0.7, iters, reg, return_status=True)
        assert status2 == 0
        assert (out2.cpu().double() - ref).abs().max() < 1e-4


def test_forward_status_after_forced_timeout_and_on_dirty_workspace(gpu_device, monkeypatch):
    """model.check_status(): 2 (+ RuntimeWarning) when the resident kernel of the last call timed out and the fallback recomputed --
    the matches still equal the normal run's; 0 for a streaming-schedule call even when the workspace held a stale status word."""
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=8, side_info_size=1)
    sd = syn.make_state_dict(cfg, seed=3)
    model = _build(cfg, sd, gpu_device)
    data = to_device(syn.make_batch(2, 200, 180, 64, 1, seed=9), gpu_device)
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    base = model.match(data, MATCH_THRESHOLD)
    assert model.check_status() == 0
    monkeypatch.setenv("OG_SINKHORN_FORCE_TIMEOUT", "1")
    out = model.match(data, MATCH_THRESHOLD)
    with pytest.warns(RuntimeWarning):
        assert model.check_status() == 2
    assert (out["scores"] - base["scores"]).abs().max().item() < 1e-4
    assert (out["matches0"] != base["matches0"]).sum().item() <= 1
    monkeypatch.delenv("OG_SINKHORN_FORCE_TIMEOUT")
    # streaming schedule on the same (now dirty: status word = 2) workspace
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "0")
    out = model.match(data, MATCH_THRESHOLD)
    assert model.check_status() == 0
    assert (out["scores"] - base["scores"]).abs().max().item() < 1e-4


# ----------------------------------------------------------------------------- per-stage goldens, og_forward_tap
@pytest.mark.parametrize("name", ["mid", "flags", "d256", "d128"])
def test_stage_taps_against_reference_layers(gpu_device, name):
    """The residual stream at every stored stage boundary against the reference's OWN sub-modules (tests/golden/make_golden.py layers):
    tap 0 = local_descriptors + positional_encoding (superglue.py:41-55), tap k = attention_gnn.layers[k-1] (attention_gnn.py:57-77:
    ResidualAttentionMessagePropagation on both images; cross layers use the UPDATED image-0 descriptors).  d256 runs the fused
    message-MLP kernel, flags the use_offset form.  A stage-local failure shows up at its own tap, not only in the scores."""
    import ast
    z = np.load(os.path.join(GOLDEN, "stage_layers_d128.npz" if name == "d128" else "stage_layers.npz"))     # d128 (round 5): 9 stages, s = 6, 520 x 512: the 128-d kernel family
    meta = ast.literal_eval(str(z[f"{name}/meta"]))
    cfg = syn.make_config(**meta["kw"])
    sd = syn.make_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    data = to_device(syn.make_batch(meta["batch"], meta["m"], meta["n"], cfg["descriptor_dim"], cfg["positional_encoding"]["side_info_size"],
                                    seed=meta["seed"]), gpu_device)
    checked = 0
    for t in range(meta["taps"]):
        if f"{name}/x0_tap{t}" not in z.files:
            continue
        x0, x1 = model.forward_tap(data, t)
        ref0, ref1 = z[f"{name}/x0_tap{t}"], z[f"{name}/x1_tap{t}"]
        e0 = np.abs(x0.cpu().numpy() - ref0).max(); e1 = np.abs(x1.cpu().numpy() - ref1).max()
        scale = max(1.0, float(np.abs(ref0).max()))
        print(f"[{name}] tap {t}: err {max(e0, e1):.2e} (|x| max {scale:.1f})")
        assert max(e0, e1) < 1e-4 * scale, (name, t, e0, e1)
        checked += 1
    assert checked >= 3
    # the taps do not disturb the call: same scores as the plain forward
    plain = model(data)["scores"]
    model.forward_tap(data, 1)
    assert torch.equal(model(data)["scores"], plain)


@pytest.mark.parametrize("name", ["c1", "mid", "flags", "nodesc", "siren"])
def test_keypoint_encoder_against_stored_encoder0(gpu_device, name):
    """The keypoint encoder alone (positional_encoding.py:16-19: the `encoder0` array of every whole-path fixture): tap 0 minus the
    descriptors (no_descriptors: tap 0 itself)."""
    z, cfg, sd, data = load_case(name)
    model = _build(cfg, sd, gpu_device)
    x0, _ = model.forward_tap(to_device(data, gpu_device), 0)
    pe0 = x0.cpu() if cfg.get("no_descriptors", False) else x0.cpu() - data["local_descriptors0"]
    ref = torch.from_numpy(z["encoder0"]).transpose(1, 2)              # stored channel-first [B, D, m]
    scale = max(1.0, float(data["local_descriptors0"].abs().max()), float(ref.abs().max()))
    err = (pe0 - ref).abs().max().item()
    print(f"[{name}] encoder err {err:.2e} (scale {scale:.1f})")
    assert err < 2e-5 * scale


def test_keypoint_encoder_and_scores_stage_entries(gpu_device):
    """og_keypoint_encoder (the encoder stage alone) = og_forward_tap(0), bit for bit, and within the bar of the oracle's
    local_descriptors + keypoint_encoder (superglue.py:44-55); og_scores (exact fp32) = g0 g1^T D^-1/2 (superglue.py:64, 80-86)."""
    import ctypes as C
    from openglue_amd import _lib
    z, cfg, sd, data = load_case("mid")
    model = _build(cfg, sd, gpu_device)
    dd = to_device(data, gpu_device)
    x0, x1 = model.encode_keypoints(dd)
    t0, t1 = model.forward_tap(dd, 0)
    assert torch.equal(x0, t0) and torch.equal(x1, t1)
    with torch.no_grad():
        kn0 = orc.normalize_keypoints(data["keypoints0"], *orc._image_wh(data, 0))
        want0 = data["local_descriptors0"] + orc.keypoint_encoder(kn0, data["side_info0"], sd, cfg)
    assert (x0.cpu() - want0).abs().max() < 1e-4 * max(1.0, want0.abs().max().item())
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, m, n, D = 3, 70, 53, 64
    g0, g1 = _rand(g, B, m, D).to(gpu_device), _rand(g, B, n, D).to(gpu_device)
    S = torch.empty(B, m, 56, device=gpu_device)
    _lib.check(lib.og_scores(g0.data_ptr(), g1.data_ptr(), B, m, n, D, S.data_ptr(), 56, torch.cuda.current_stream().cuda_stream), "og_scores")
    want = (g0.double() @ g1.double().transpose(1, 2)) * D ** -0.5
    assert (S[:, :, :n].double() - want).abs().max() < 1e-5


@pytest.mark.parametrize("tag,scale", [("unit", 1.0), ("x4", 4.0)])
def test_forward_on_trained_like_checkpoint_fixture(gpu_device, tag, scale):
    """A checkpoint with the statistics training produces and random initialisation never does (dead BatchNorm channels whose
    eval-mode fold multiplies a conv column by ~3000, gamma / sigma ~ 100, weights of 150): og_pack_weights gives those matrices a
    smaller power-of-two pre-scale (round 2 refused them with OG_E_RANGE) and the kernels read it from the packed blob.  Against
    the REFERENCE's outputs (tests/golden/make_golden.py trained), unit-norm descriptors included."""
    import ast
    z = np.load(os.path.join(GOLDEN, "trained.npz"))
    cfg = syn.make_config(**ast.literal_eval(str(z["config_kwargs"])))
    sd = syn.make_trained_like_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    data = syn.make_batch(int(z["batch"]), int(z["m"]), int(z["n"]), 256, 1, seed=int(z["seed"]), desc_scale=scale)
    out = {k: v.cpu() for k, v in model.match(to_device(data, gpu_device), MATCH_THRESHOLD).items()}
    err = np.abs(out["scores"].numpy() - z[f"{tag}_scores"]).max()
    print(f"[trained/{tag}] scores err {err:.2e} (|scores| max {np.abs(z[f'{tag}_scores']).max():.1f})")
    assert err < TOL_SCORES
    assert np.abs(out["context_descriptors0"].numpy() - z[f"{tag}_context_descriptors0"]).max() < TOL_SCORES
    ndiff, unexplained, _ = _index_agreement(out["matches0"], out["scores"], sd, cfg, data)
    assert unexplained == 0, (ndiff, unexplained)


def test_sinkhorn_resident_ignores_dirty_padding_columns(gpu_device, monkeypatch):
    """og_sinkhorn on a caller's buffer whose padding columns [n, lds) hold NaN (n % 4 != 0): the resident schedule masks columns
    >= n through -inf duals, which does not neutralise a NaN -- the straddling chunk is cleared as the rows are loaded."""
    from openglue_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv("OG_SINKHORN_RESIDENT", "2")
    B, m, n, iters = 2, 260, 1023, 12
    g = torch.Generator().manual_seed(5)
    S = _rand(g, B, m, n, scale=2.0)
    Sp = torch.full((B, m, 1024), float("nan"))
    Sp[:, :, :n] = S
    Sp = Sp.to(gpu_device)
    ws = torch.empty(lib.og_sinkhorn_workspace_bytes(B, m, n), device=gpu_device, dtype=torch.uint8)
    out = torch.empty(B, m + 1, n + 1, device=gpu_device)
    assert lib.og_sinkhorn_schedule(B, m, n, iters) == 1           # one resident launch
    assert lib.og_sinkhorn_schedule(64, 1024, 1024, 100) == 2 and lib.og_sinkhorn_schedule(32, 2048, 2048, 100) == 4 and lib.og_sinkhorn_schedule(8, 4096, 4096, 100) == 4
    assert lib.og_sinkhorn_schedule(4, 5000, 100, 100) == 0          # more than 4096 rows: streaming
    _lib.check(lib.og_sinkhorn(Sp.data_ptr(), 1024, 0.7, B, m, n, iters, 1.0, out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "og_sinkhorn")
    assert lib.og_sinkhorn_status(ws.data_ptr(), B, m, n) == 0
    assert torch.isfinite(out).all()
    assert (out.cpu().double() - _sinkhorn_ref(S, 0.7, iters, 1.0)).abs().max() < 1e-4


def test_activation_overflow_is_reported_not_silent(gpu_device):
    """Activations beyond the binary16 range of the (hi, lo) operands (|x| >= 65504) turn into inf inside the GNN; the non-finite
    values reach the scores and og_forward_status / model.check_status() report it (status 3) instead of NaN scores going unnoticed."""
    cfg = syn.make_config(descriptor_dim=64, num_stages=1, num_heads=4, num_iters=5, side_info_size=1)
    sd = syn.make_state_dict(cfg, seed=0)
    model = _build(cfg, sd, gpu_device)
    data = to_device(syn.make_batch(2, 80, 70, 64, 1, seed=3), gpu_device)
    model.match(data, MATCH_THRESHOLD)
    assert model.check_status() == 0
    big = dict(data)
    big["local_descriptors0"] = data["local_descriptors0"] * 1e5          # ~3e6: far outside binary16
    out = model.match(big, MATCH_THRESHOLD)
    assert not torch.isfinite(out["scores"]).all()
    with pytest.raises(RuntimeError, match="non-finite"):
        model.check_status()
    model.match(data, MATCH_THRESHOLD)                                     # the flag describes the LAST call only
    assert model.check_status() == 0

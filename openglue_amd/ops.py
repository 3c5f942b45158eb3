"""Thin torch-tensor wrappers over the per-stage C-ABI entry points (include/openglue_amd.h).

PyTorch is plumbing here: it owns the device buffers and the current HIP stream; all arithmetic
happens in libopenglue_amd.so.  Every function requires CUDA(HIP) fp32 tensors and raises otherwise.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU; openglue_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
            res: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    """epilogue(a @ b^T): a [M,K] or [Z,M,K], b [N,K] or [Z,N,K] (exact fp32 MFMA)."""
    lib = _lib.load()
    a, b = _req(a, "a"), _req(b, "b")
    batched = a.dim() == 3
    Z = a.shape[0] if batched else 1
    M, K = a.shape[-2:]
    N = b.shape[-2]
    out = torch.empty((Z, M, N) if batched else (M, N), device=a.device, dtype=torch.float32)
    if bias is not None: bias = _req(bias, "bias")
    if res is not None: res = _req(res, "res")
    if alpha is not None: alpha = _req(alpha, "alpha")
    rc = lib.og_gemm_nt(a.data_ptr(), K, M * K if batched else 0, b.data_ptr(), K, N * K if b.dim() == 3 else 0,
                        out.data_ptr(), N, M * N, M, N, K, Z, _ptr(bias), int(relu), _ptr(res), N, _ptr(alpha),
                        float(scale), _stream())
    _lib.check(rc, "og_gemm_nt")
    return out


def split_f16(x: torch.Tensor):
    """fp32 tensor -> (hi, lo) float16 planes with x = hi + lo, lo = f16(x - hi) (the GNN's operand format)."""
    lib = _lib.load()
    x = _req(x, "x")
    if x.numel() % 4:
        raise ValueError("split_f16: numel must be a multiple of 4")
    hi = torch.empty(x.shape, device=x.device, dtype=torch.float16)
    lo = torch.empty(x.shape, device=x.device, dtype=torch.float16)
    _lib.check(lib.og_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), _stream()), "og_split_f16")
    return hi, lo


def merge_f16(hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    """Inverse of split_f16 (plain torch; for tests and debugging only)."""
    return hi.float() + lo.float()


def split_f16_hl(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, cols] (cols % 32 == 0) -> float16 [rows, 2*cols] in the GEMM's hl32 row format: hi and
    lo interleaved in groups of 32 channels, so a 32-channel k-slab is one 128-byte line."""
    lib = _lib.load()
    x = _req(x, "x")
    rows, cols = x.shape
    out = torch.empty(rows, 2 * cols, device=x.device, dtype=torch.float16)
    _lib.check(lib.og_split_f16_hl(x.data_ptr(), rows, cols, cols, out.data_ptr(), 2 * cols, _stream()), "og_split_f16_hl")
    return out


def merge_f16_hl(t: torch.Tensor) -> torch.Tensor:
    """Inverse of split_f16_hl (plain torch; for tests and debugging only)."""
    rows, c2 = t.shape
    g = t.view(rows, c2 // 64, 2, 32).float()
    return (g[:, :, 0] + g[:, :, 1]).reshape(rows, c2 // 2)


def gemm_nt_f16x3(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                  res: Optional[torch.Tensor] = None, want: Optional[str] = None):
    """epilogue(a @ b^T) through the split-f16 3-pass MFMA kernel: a [M,K], b [N,K] fp32 are converted to
    hl32 rows on the device first (b pre-scaled by 256, like the packed weights, so its lo parts are normal
    f16 numbers).  Returns the fp32 result; with want='planes' also the (hi, lo) output
    planes, with want='hl' also the hl32 output rows."""
    lib = _lib.load()
    a, b = _req(a, "a"), _req(b, "b")
    M, K = a.shape
    N = b.shape[0]
    a_hl, b_hl = split_f16_hl(a), split_f16_hl(b * 256.0)
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    ch = cl = None
    ldch = N
    if want == "planes":
        ch = torch.empty(M, N, device=a.device, dtype=torch.float16)
        cl = torch.empty(M, N, device=a.device, dtype=torch.float16)
    elif want == "hl":
        ch = torch.empty(M, 2 * N, device=a.device, dtype=torch.float16)
        ldch = 2 * N
    elif want is not None:
        raise ValueError(want)
    if bias is not None: bias = _req(bias, "bias")
    if res is not None: res = _req(res, "res")
    rc = lib.og_gemm_nt_f16x3(a_hl.data_ptr(), 2 * K, b_hl.data_ptr(), 2 * K, M, N, K, 1.0 / 256.0, _ptr(bias), int(relu), _ptr(res), N,
                              out.data_ptr(), N, _ptr(ch), _ptr(cl), ldch, int(want == "hl"), _stream())
    _lib.check(rc, "og_gemm_nt_f16x3")
    if want == "planes":
        return out, ch, cl
    if want == "hl":
        return out, ch
    return out


def gemm_nt_f16x3_split_only(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                             res: Optional[torch.Tensor] = None, planes: bool = False) -> torch.Tensor:
    """The forms the GNN launches (no fp32 output, so whole-tile shapes take the 256-tile kernel's compile-time epilogues):
    q/k/v projections (planes=True), fc.0 (relu=True), fc.3 (res: fp32 here, handed to the kernel as hl32 rows like the
    residual stream of og_forward).  Returns the output merged back to fp32."""
    lib = _lib.load()
    a, b = _req(a, "a"), _req(b, "b")
    M, K = a.shape
    N = b.shape[0]
    a_hl, b_hl = split_f16_hl(a), split_f16_hl(b * 256.0)
    if bias is not None: bias = _req(bias, "bias")
    res_hl = split_f16_hl(_req(res, "res")) if res is not None else None
    if planes:
        ch = torch.empty(M, N, device=a.device, dtype=torch.float16); cl = torch.empty_like(ch); ldch = N
    else:
        ch = torch.empty(M, 2 * N, device=a.device, dtype=torch.float16); cl = None; ldch = 2 * N
    rc = lib.og_gemm_nt_f16x3_reshl(a_hl.data_ptr(), 2 * K, b_hl.data_ptr(), 2 * K, M, N, K, 1.0 / 256.0, _ptr(bias), int(relu),
                                    _ptr(res_hl), 2 * N, None, N, ch.data_ptr(), _ptr(cl), ldch, int(not planes), _stream())
    _lib.check(rc, "og_gemm_nt_f16x3_reshl")
    return merge_f16(ch, cl) if planes else merge_f16_hl(ch)


def mlp_block(x: torch.Tensor, o: torch.Tensor, w0: torch.Tensor, b0: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor,
              return_rows: bool = False):
    """The message MLP of one GNN layer as ONE launch (og_mlp_block): x + w3 @ relu(w0 @ [x ; o] + b0) + b3 on token-major
    fp32 x, o [M, D]; w0 [2D, 2D], w3 [D, 2D] (BatchNorm / out_proj already folded).  x and o are converted to the [x | O] hl32
    rows of og_forward on the device, the weights to the kernel's fragment-major stream on the host."""
    lib = _lib.load()
    x, o = _req(x, "x"), _req(o, "o")
    M, D = x.shape
    nbytes = lib.og_mlp_block_stream_bytes(D)
    if nbytes == 0:
        raise RuntimeError(f"og_mlp_block: no fused message-MLP kernel for D = {D}")
    w0h, w3h = w0.detach().float().cpu().contiguous(), w3.detach().float().cpu().contiguous()
    stream_host = torch.empty(nbytes, dtype=torch.uint8)
    _lib.check(lib.og_mlp_block_pack(D, w0h.data_ptr(), w3h.data_ptr(), stream_host.data_ptr()), "og_mlp_block_pack")
    stream_dev = stream_host.to(x.device)
    rows = split_f16_hl(torch.cat([x, o], dim=1).contiguous())          # [M][4D halves]: x | O
    b0, b3 = _req(b0, "b0"), _req(b3, "b3")
    _lib.check(lib.og_mlp_block(D, rows.data_ptr(), 4 * D, M, stream_dev.data_ptr(), b0.data_ptr(), b3.data_ptr(), _stream()), "og_mlp_block")
    out = merge_f16_hl(rows)[:, :D].contiguous()
    return (out, rows) if return_rows else out


def proj_block(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, split_row: int = 0, cols_a=None, cols_b=None):
    """The small-batch projection kernel (og_proj_block): y = x @ w.T + bias on token-major fp32 x [M, K], w [N, K], K = 256 or 128, N a multiple of 32.
    Rows below split_row get the output columns cols_a = (c0, c1), the others cols_b (multiples of 32; default: all N); columns outside a
    row's range come back as zeros."""
    lib = _lib.load()
    x = _req(x, "x")
    M, K = x.shape
    N = w.shape[0]
    nbytes = lib.og_proj_block_stream_bytes(N, K)
    if nbytes == 0:
        raise RuntimeError(f"og_proj_block: no kernel for N = {N}, K = {K}")
    wh = w.detach().float().cpu().contiguous()
    stream_host = torch.empty(nbytes, dtype=torch.uint8)
    _lib.check(lib.og_proj_block_pack(N, K, wh.data_ptr(), stream_host.data_ptr()), "og_proj_block_pack")
    stream_dev = stream_host.to(x.device)
    rows = split_f16_hl(x)                                       # [M][2K halves]
    bias = _req(bias, "bias")
    inv = torch.full((1,), 1.0 / 256.0, device=x.device, dtype=torch.float32)
    yh = torch.zeros(M, N, device=x.device, dtype=torch.float16)
    yl = torch.zeros_like(yh)
    a, b = cols_a or (0, N), cols_b or (0, N)
    _lib.check(lib.og_proj_block(rows.data_ptr(), 2 * K, M, K, N, stream_dev.data_ptr(), bias.data_ptr(), inv.data_ptr(), yh.data_ptr(), yl.data_ptr(), N,
                                 split_row, a[0] // 32, a[1] // 32, b[0] // 32, b[1] // 32, _stream()), "og_proj_block")
    return merge_f16(yh, yl)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, return_lse: bool = False):
    """Multi-head softmax attention on token-major fp32 tensors q [Z,nq,D], k,v [Z,nk,D]; head h owns
    channels h*d..(h+1)*d-1.  q must already carry the d^-1/2 scale (the log2(e) factor of the kernel's
    base-2 softmax is applied here).  Operands are converted to the kernel's split-f16 planes on the
    device; the result planes are merged back to fp32.  return_lse: also the row log-sum-exp of the scaled scores
    [Z, num_heads, nq] (natural units), straight from the kernel's online-softmax state."""
    lib = _lib.load()
    q, k, v = _req(q * 1.4426950408889634, "q"), _req(k, "k"), _req(v, "v")
    Z, nq, D = q.shape
    nk = k.shape[1]
    (qh, ql), (kh, kl), (vh, vl) = split_f16(q), split_f16(k), split_f16(v)
    oh = torch.empty(Z, nq, D, device=q.device, dtype=torch.float16)
    ol = torch.empty_like(oh)
    lse = torch.empty(Z, num_heads, nq, device=q.device, dtype=torch.float32) if return_lse else None
    rc = lib.og_attention(qh.data_ptr(), ql.data_ptr(), D, kh.data_ptr(), kl.data_ptr(), D, vh.data_ptr(), vl.data_ptr(), D,
                          oh.data_ptr(), ol.data_ptr(), D, Z, nq, nk, num_heads, D // num_heads,
                          None if lse is None else lse.data_ptr(), _stream())
    _lib.check(rc, "og_attention")
    return (merge_f16(oh, ol), lse) if return_lse else merge_f16(oh, ol)


def sinkhorn(S: torch.Tensor, dustbin: float, iters: int, reg: float = 1.0, return_status: bool = False):
    """S [B,m,n] raw scores -> log-assignment [B,m+1,n+1] (superglue.py:88-111)."""
    lib = _lib.load()
    S = _req(S, "S")
    B, m, n = S.shape
    lds = (n + 3) // 4 * 4
    if lds != n:
        Sp = torch.zeros(B, m, lds, device=S.device, dtype=torch.float32)
        Sp[:, :, :n] = S
        S = Sp
    ws = torch.empty(lib.og_sinkhorn_workspace_bytes(B, m, n), device=S.device, dtype=torch.uint8)
    out = torch.empty(B, m + 1, n + 1, device=S.device, dtype=torch.float32)
    rc = lib.og_sinkhorn(S.data_ptr(), lds, float(dustbin), B, m, n, int(iters), float(reg), out.data_ptr(),
                         ws.data_ptr(), _stream())
    _lib.check(rc, "og_sinkhorn")
    if return_status:        # og_sinkhorn_status: 0 = ok; 2 = the resident kernel timed out and the fallback recomputed (valid); 1 = invalid
        torch.cuda.current_stream(S.device).synchronize()       # og_sinkhorn_status waits for the NULL stream only
        return out, int(lib.og_sinkhorn_status(ws.data_ptr(), B, m, n))
    return out


def extract_matches(scores: torch.Tensor, match_threshold: float, both_sides: bool = True) -> Dict[str, torch.Tensor]:
    """scores [B,m+1,n+1] -> matches0/matching_scores0 (matching_module.py:174-187) and, if
    both_sides, matches1/matching_scores1 (inference.py:183-188)."""
    lib = _lib.load()
    scores = _req(scores, "scores")
    B, m1, n1 = scores.shape
    m, n = m1 - 1, n1 - 1
    dev = scores.device
    ws = torch.empty(lib.og_matches_workspace_bytes(B, m, n), device=dev, dtype=torch.uint8)
    out = {"matches0": torch.empty(B, m, device=dev, dtype=torch.int64),
           "matching_scores0": torch.empty(B, m, device=dev, dtype=torch.float32)}
    if both_sides:
        out["matches1"] = torch.empty(B, n, device=dev, dtype=torch.int64)
        out["matching_scores1"] = torch.empty(B, n, device=dev, dtype=torch.float32)
    rc = lib.og_extract_matches(scores.data_ptr(), B, m, n, float(match_threshold), out["matches0"].data_ptr(),
                                out["matching_scores0"].data_ptr(), _ptr(out.get("matches1")),
                                _ptr(out.get("matching_scores1")), ws.data_ptr(), _stream())
    _lib.check(rc, "og_extract_matches")
    return out

"""Training slice: the optimal-transport layer as a torch.autograd.Function backed by HIP kernels.

`matching_log_probs(S, dustbin_score, num_iters, reg)` is a differentiable drop-in for `SuperGlue.get_matching_probs`
(reference superglue.py:88-111 -> log_otp_solver, optimal_transport.py:20-28): forward = og_sinkhorn_train_forward (keeps
the dual trajectory), backward = og_sinkhorn_backward (unrolled iterations, reverse order) -- so the NLL of
utils/losses.py:7-53 computed on its result back-propagates into the score matrix S and into `dustbin_score` without any
torch math on the way.  PyTorch is used for what it is here: autograd bookkeeping, device memory, the current stream.

`batch_norm_train` / `feed_forward_train` are the train-mode forward of the reference's MLP building block (models/utils.py:48-58:
Conv1d -> ReLU -> BatchNorm1d with batch statistics and running-statistics update) on token-major activations.

`feed_forward_train_autograd` adds the backward of that block (1x1 conv: dX, dW, db on the exact-fp32 GEMM; ReLU + train-mode
BatchNorm: og_batchnorm_train_backward), so the keypoint-encoder MLP / a message MLP can be trained end to end on HIP kernels.

`superglue_forward_train` (below) wires them -- plus `Conv1x1`, `SoftmaxAttention` (materialised attention matrix, batched exact-fp32
GEMMs, row-softmax forward / backward kernels) and `MatchingScores` -- into the whole training-mode forward of the reference
(superglue.py:29-72): `SuperGlue(config).train()(data)` returns tensors whose `loss.backward()` reaches every parameter.  `LinearAttentionCore`
(+ `linear_attention_elu_train`, `favor_relu_attention_train`) is the O(N) attention of attention.py:22-40 / :86-95 under autograd, the
Siren encoder (models/utils.py:32-45) runs through `Conv1x1` + sin(30 x).
"""
from __future__ import annotations

import os as _os

import torch
from typing import Optional

from . import _lib


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class SinkhornOT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S: torch.Tensor, dustbin_score: torch.Tensor, num_iters: int, reg: float) -> torch.Tensor:
        if not S.is_cuda:
            raise RuntimeError("openglue_amd.train: S must be on the MI355X; there is no CPU fallback")
        if S.dim() != 3 or num_iters < 1:
            raise ValueError("S must be [B, m, n] and num_iters >= 1")
        lib = _lib.load()
        B, m, n = S.shape
        lds = (n + 3) // 4 * 4
        Sp = S.detach().to(torch.float32)
        if lds != n or not Sp.is_contiguous():
            buf = torch.zeros(B, m, lds, device=S.device, dtype=torch.float32)
            buf[:, :, :n] = Sp
            Sp = buf
        nbytes = lib.og_sinkhorn_train_workspace_bytes(B, m, n, int(num_iters))
        if nbytes == 0:
            raise RuntimeError("og_sinkhorn_train_workspace_bytes: unsupported shape (n <= 4159, num_iters >= 1)")
        ws = torch.empty(nbytes + 256, device=S.device, dtype=torch.uint8)
        off = (-ws.data_ptr()) % 256
        scores = torch.empty(B, m + 1, n + 1, device=S.device, dtype=torch.float32)
        # the learnable dustbin score is read ON the device (ABI v6): no float(tensor) here, the host never waits for the GPU mid-step
        zdev = dustbin_score.detach().to(device=S.device, dtype=torch.float32).reshape(1).contiguous()
        with torch.cuda.device(S.device):
            _lib.check(lib.og_sinkhorn_train_forward(Sp.data_ptr(), lds, 0.0, zdev.data_ptr(), B, m, n, int(num_iters), float(reg),
                                                     scores.data_ptr(), ws.data_ptr() + off, _stream(S)), "og_sinkhorn_train_forward")
        ctx.save_for_backward(Sp, zdev)
        ctx.ws, ctx.off, ctx.args = ws, off, (B, m, n, lds, int(num_iters), float(reg))
        ctx.dustbin_meta = (dustbin_score.dtype, dustbin_score.shape)
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        lib = _lib.load()
        Sp, zdev = ctx.saved_tensors
        B, m, n, lds, iters, reg = ctx.args
        g = grad_scores.detach().to(torch.float32).contiguous()
        dS = torch.empty(B, m, lds, device=Sp.device, dtype=torch.float32)
        dz = torch.zeros(1, device=Sp.device, dtype=torch.float32)
        with torch.cuda.device(Sp.device):
            _lib.check(lib.og_sinkhorn_backward(Sp.data_ptr(), lds, 0.0, zdev.data_ptr(), B, m, n, iters, reg, g.data_ptr(), ctx.ws.data_ptr() + ctx.off,
                                                dS.data_ptr(), lds, dz.data_ptr(), _stream(Sp)), "og_sinkhorn_backward")
        dtype, shape = ctx.dustbin_meta
        return dS[:, :, :n], dz.reshape(shape).to(dtype), None, None


def matching_log_probs(S: torch.Tensor, dustbin_score: torch.Tensor, num_iters: int, reg: float = 1.0) -> torch.Tensor:
    """Differentiable `scores` [B, m+1, n+1] from the raw score matrix S [B, m, n] (superglue.py:88-111)."""
    return SinkhornOT.apply(S, dustbin_score, num_iters, reg)


def batch_norm_train(x: torch.Tensor, weight, bias, running_mean, running_var, momentum: float = 0.1, eps: float = 1e-5,
                     return_stats: bool = False, out: Optional[torch.Tensor] = None):
    """nn.BatchNorm1d in training mode on TOKEN-MAJOR activations x [T, C] (T = B*N rows of the reference's [B, C, N] tensor):
    batch statistics per channel, `running_mean` / `running_var` updated IN PLACE like torch.  Returns y [T, C] (and the saved
    mean / inverse std when `return_stats`).  Forward only."""
    if not x.is_cuda:
        raise RuntimeError("openglue_amd.train: x must be on the MI355X; there is no CPU fallback")
    if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError("x must be a float32 [T, C] tensor with contiguous channels")
    lib = _lib.load()
    T, C = x.shape
    nbytes = lib.og_batchnorm_train_workspace_bytes(T, C)
    if nbytes == 0:
        raise ValueError("og_batchnorm_train_workspace_bytes: channels must be a multiple of 4, rows >= 1")
    for name, t in (("weight", weight), ("bias", bias), ("running_mean", running_mean), ("running_var", running_var)):
        if t is not None and (t.device != x.device or t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous float32 [C] tensor on the device of x")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    y = torch.empty(T, C, device=x.device, dtype=torch.float32) if out is None else out          # out: a [T, C] row slice of a larger buffer
    mean = torch.empty(C, device=x.device, dtype=torch.float32) if return_stats else None
    invstd = torch.empty(C, device=x.device, dtype=torch.float32) if return_stats else None
    p = lambda t: None if t is None else t.data_ptr()       # noqa: E731
    with torch.cuda.device(x.device):
        _lib.check(lib.og_batchnorm_train_forward(x.data_ptr(), x.stride(0), T, C, p(weight), p(bias), float(eps), float(momentum),
                                                  p(running_mean), p(running_var), y.data_ptr(), y.stride(0), p(mean), p(invstd),
                                                  ws.data_ptr(), _stream(x)), "og_batchnorm_train_forward")
    return (y, mean, invstd) if return_stats else y


def feed_forward_train(x: torch.Tensor, state_dict, prefix: str = "", momentum: float = 0.1) -> torch.Tensor:
    """The reference's FeedForwardNet (models/utils.py:48-58) in TRAINING mode on token-major x [T, C_in]: for every hidden layer
    the exact-fp32 MFMA GEMM with fused bias + ReLU (og_gemm_nt), then train-mode BatchNorm (og_batchnorm_train_forward: batch
    statistics, the running statistics inside `state_dict` are updated in place), then the last 1x1 conv.  Parameter names are
    the nn.Sequential ones: `{prefix}{3i}.weight|bias` (Conv1d, weight [out, in, 1]), `{prefix}{3i+2}.*` (BatchNorm1d)."""
    from . import ops
    n_conv = len({k for k in state_dict if k.startswith(prefix) and k.endswith(".weight") and state_dict[k].dim() == 3})
    for i in range(n_conv):
        w = state_dict[f"{prefix}{3 * i}.weight"]
        x = ops.gemm_nt(x, w.reshape(w.shape[0], w.shape[1]).contiguous(), state_dict[f"{prefix}{3 * i}.bias"], relu=i + 1 < n_conv)
        if i + 1 < n_conv:
            bn = f"{prefix}{3 * i + 2}"
            x = batch_norm_train(x, state_dict[bn + ".weight"], state_dict[bn + ".bias"], state_dict[bn + ".running_mean"],
                                 state_dict[bn + ".running_var"], momentum)
    return x


# ------------------------------------------------------------------------------------------------------------------
# backward of the MLP block
def _ws(x: torch.Tensor, rows: int, C: int) -> torch.Tensor:
    lib = _lib.load()
    n = lib.og_batchnorm_train_workspace_bytes(rows, max(4, (C + 3) // 4 * 4))
    return torch.empty(max(n, 256), device=x.device, dtype=torch.uint8)


def _transpose_pad(x: torch.Tensor, mult: int = 4) -> torch.Tensor:
    """[R, C] -> [C, round_up(R, mult)] (zero tail): both operands of the weight-gradient GEMM must be K-contiguous, K = R
    (mult = 32: the contraction length the split-f16 kernel needs)."""
    lib = _lib.load()
    R, C = x.shape
    R4 = (R + mult - 1) // mult * mult
    out = torch.zeros(C, R4, device=x.device, dtype=torch.float32) if R4 != R else torch.empty(C, R, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(lib.og_transpose_f32(x.data_ptr(), x.stride(0), R, C, out.data_ptr(), R4, _stream(x)), "og_transpose_f32")
    return out


# The 1x1 convs of the training step run on the exact-fp32 MFMA kernel by DEFAULT; OG_TRAIN_F16X3=1 moves them to the split-f16 3-pass
# MFMA GEMM of the inference path (fp32-class accuracy at ~3x the rate: DESIGN.md 4.1) wherever the contraction length is a multiple of 32.  Gradients can be far below the binary16 range (an NLL averaged over thousands of keypoints: 1e-7 ... 1e-3), so
# a gradient operand is multiplied by a power of two that brings its largest entry to ~2^11 before the (hi, lo) split and the product is
# scaled back -- exact, and computed ON THE DEVICE (no host synchronisation).
def _use_f16x3() -> bool:
    import os
    return os.environ.get("OG_TRAIN_F16X3", "0") != "0"


def _pow2_to(t: torch.Tensor, target: float) -> torch.Tensor:
    """0-d device tensor 2^k with amax(t) * 2^k in (target / 2, target]."""
    amax = t.abs().amax().clamp_min(1e-30)
    return torch.exp2(torch.floor(torch.log2(target / amax)))


def _gemm_fast(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False, scale_a: bool = False) -> torch.Tensor:
    """epilogue(a @ b^T), a [M, K], b [N, K] fp32: split-f16 kernel when K % 32 == 0 and N % 4 == 0, else the exact-fp32 kernel.
    scale_a: `a` is a gradient (see above).  b (weights / activations, additionally x256 inside the kernel wrapper) is scaled DOWN only
    when its largest entry would leave binary16."""
    from . import ops
    M, K = a.shape
    N = b.shape[0]
    if not (_use_f16x3() and K % 32 == 0 and N % 4 == 0 and K >= 32):
        return ops.gemm_nt(a, b, bias, relu=relu)
    sb = torch.clamp(_pow2_to(b, 64.0), max=1.0)                  # 256 * 64 = 2^14
    sa = _pow2_to(a, 2048.0) if scale_a else torch.clamp(_pow2_to(a, 16384.0), max=1.0)
    s = sa * sb
    out = ops.gemm_nt_f16x3((a * sa).contiguous(), (b * sb).contiguous(), None if bias is None else (bias * s).contiguous(), relu=relu)
    return out * (1.0 / s)


def _env_int(name: str, default: int, lo: int = 1) -> int:
    """An experiment knob from the environment: a malformed or out-of-range value falls back to the default with a warning (never an import error)."""
    raw = _os.environ.get(name)
    if raw is None:
        return default
    try:
        v = int(raw)
    except ValueError:
        v = None
    if v is None or v < lo:
        import warnings
        warnings.warn(f"{name}={raw!r} ignored (expected an integer >= {lo}); using {default}", RuntimeWarning)
        return default
    return v


_SPLITK_WGS = _env_int("OG_TRAIN_SPLITK_WGS", 512)           # experiments: the workgroup count a split-K weight gradient aims at


def _gemm_splitk(dz: torch.Tensor, x: torch.Tensor, with_colsum: bool = False):
    """dW = dz^T x for dz [T, Cout], x [T, Cin]: a [Cout, Cin] result (at most a few 128 x 128 tiles) contracted over ALL T tokens.
    As one GEMM launch that is a grid of <= 16 workgroups on a 256-CU chip (round 2: 250-310 us per launch, 72 % of the training step);
    here the token axis is cut into `parts` chunks that run as the problems of ONE batched launch and the partial products are summed.
    Both operands are token-major = K-MAJOR for this product: og_gemm_kmajor reads them as they lie (no transposed copies).
    with_colsum: also db = column sums of dz, out of the same launch (an extra output column) -> (dW, db)."""
    T, Cout = dz.shape
    Cin = x.shape[1]
    if Cout % 4 or Cin % 4 or dz.stride(0) % 4 or x.stride(0) % 4:
        raise ValueError("_gemm_splitk: channel counts / row strides must be multiples of 4")
    tiles = ((Cout + 127) // 128) * ((Cin + 63) // 64)                    # 128 x 64 tiles (csrc/gemm_f32.hip)
    parts = max(1, min(T // 64, (_SPLITK_WGS + tiles - 1) // tiles))      # ~512 workgroups of 128 x 64 (measured best of 512 ... 1536: r05_y), at least two 32-row k-steps each
    Kc = (T + parts - 1) // parts
    ldc = Cin + 4 if with_colsum else Cin
    part = torch.empty(parts, Cout, ldc, device=dz.device, dtype=torch.float32)
    _gemm_km(dz.device, dz.data_ptr(), dz.stride(0), Kc * dz.stride(0), 1, x.data_ptr(), x.stride(0), Kc * x.stride(0),
             part.data_ptr(), ldc, Cout * ldc, Cout, Cin, Kc, parts, k_total=T, a_colsum=with_colsum)
    # the partial products summed in part order by ONE launch that writes dW and db where they belong (round 4: a torch reduction over the
    # padded rows plus two strided copies)
    dW = torch.empty(Cout, Cin, device=dz.device, dtype=torch.float32)
    db = torch.empty(Cout, device=dz.device, dtype=torch.float32) if with_colsum else None
    with torch.cuda.device(dz.device):
        _lib.check(_lib.load().og_splitk_reduce(part.data_ptr(), parts, Cout, ldc, Cin, dW.data_ptr(), None if db is None else db.data_ptr(),
                                                _stream(dz)), "og_splitk_reduce")
    return (dW, db) if with_colsum else dW


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A 2-D fp32 tensor the GEMMs can read as it lies: unit column stride, row stride a multiple of 4 floats, 16-byte aligned base (a
    column range of a wider matrix -- the gradient of one operand of a concatenation -- qualifies); anything else is copied."""
    t = t.detach()
    if t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t
    return t.to(torch.float32).contiguous()


def _conv_backward(x: torch.Tensor, W: torch.Tensor, dz: torch.Tensor, need_dx: bool, need_dw: bool = True):
    """1x1 conv on token rows, z = x W^T + b:  dx = dz W,  dW = dz^T x,  db = column sums of dz -- all on HIP kernels.
    need_dw False (W is a buffer: the FAVOR projection): only dx."""
    lib = _lib.load()
    T, Cout = dz.shape
    dx = None
    if need_dx and _use_f16x3() and Cout % 32 == 0:                                              # (K = Cout must be a multiple of 32 on BOTH operands)
        dx = _gemm_fast(dz, _transpose_pad(W, 32), scale_a=True)                                 # [T, Cout] x [Cin, Cout]^T
    elif need_dx:                                                                                # dz [T, Cout] x W [Cout][Cin] as it lies (k-major B)
        Cin = W.shape[1]
        dx = torch.empty(T, Cin, device=dz.device, dtype=torch.float32)
        _gemm_km(dz.device, dz.data_ptr(), dz.stride(0), 0, 0, W.data_ptr(), W.stride(0), 0, dx.data_ptr(), Cin, 0, T, Cin, Cout, 1)
    if not need_dw:
        return dx, None, None
    dW, db = _gemm_splitk(dz, x, with_colsum=True)                                               # dz^T [x | 1]: one launch
    return dx, dW, db


class Conv1x1(torch.autograd.Function):
    """y = x W^T + b on token-major x [T, Cin] (nn.Conv1d(kernel_size=1) of the reference on [B, Cin, N])."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return _gemm_fast(x.detach(), W.detach().contiguous(), b.detach())

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx, dW, db = _conv_backward(x.detach(), W.detach().contiguous(), _rows(dy) if not _use_f16x3() else dy.detach().contiguous(),
                                    ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dW, db


class ConvReluBNTrain(torch.autograd.Function):
    """The hidden block of FeedForwardNet in training mode (models/utils.py:52-56): y = BatchNorm_train(relu(x W^T + b)); running
    statistics are updated in place in forward.
    splits = (T0, T1, ...): x holds the token rows of SEVERAL calls of the reference (the two images of a self layer,
    attention_gnn.py:63-66) stacked: the conv and its backward run once over all rows, BatchNorm keeps its per-call semantics
    (batch statistics and one running-statistics update per row range, in order)."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, running_mean, running_var, momentum, eps, splits=None):
        a = _gemm_fast(x.detach(), W.detach().contiguous(), b.detach(), relu=True)
        splits = tuple(splits) if splits else (a.shape[0],)
        if sum(splits) != a.shape[0]:
            raise ValueError("ConvReluBNTrain: splits must add up to the rows of x")
        y = torch.empty_like(a)
        stats, r0 = [], 0
        for rows in splits:
            _, mean, invstd = batch_norm_train(a[r0:r0 + rows], gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps,
                                               return_stats=True, out=y[r0:r0 + rows])
            stats += [mean, invstd]
            r0 += rows
        ctx.save_for_backward(x, W, gamma, a, *stats)
        ctx.splits = splits
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, W, gamma, a, *stats = ctx.saved_tensors
        T, C = a.shape
        dy = dy.detach().contiguous()
        dz = torch.empty_like(a)
        dgamma = dbeta = None
        r0 = 0
        for i, rows in enumerate(ctx.splits):
            mean, invstd = stats[2 * i], stats[2 * i + 1]
            dg = torch.empty(C, device=a.device, dtype=torch.float32)
            dbt = torch.empty(C, device=a.device, dtype=torch.float32)
            a_s, dy_s, dz_s = a[r0:r0 + rows], dy[r0:r0 + rows], dz[r0:r0 + rows]
            ws = _ws(a, rows, C)
            with torch.cuda.device(a.device):
                _lib.check(lib.og_batchnorm_train_backward(a_s.data_ptr(), a_s.stride(0), dy_s.data_ptr(), dy_s.stride(0), rows, C,
                                                           gamma.detach().data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1, dz_s.data_ptr(),
                                                           dz_s.stride(0), dg.data_ptr(), dbt.data_ptr(), ws.data_ptr(), _stream(a)),
                           "og_batchnorm_train_backward")
            dgamma = dg if dgamma is None else dgamma + dg
            dbeta = dbt if dbeta is None else dbeta + dbt
            r0 += rows
        dx, dW, db = _conv_backward(x.detach(), W.detach().contiguous(), dz, ctx.needs_input_grad[0])
        return dx, dW, db, dgamma, dbeta, None, None, None, None, None


class MLPBlockTrain(torch.autograd.Function):
    """The two-conv FeedForwardNet of a GNN layer in training mode (attention_gnn.py:41-44, models/utils.py:48-58) as ONE node:
    z = (BatchNorm_train(relu(x W0^T + b0))) W3^T + b3.  Same kernels as ConvReluBNTrain followed by Conv1x1, but the BatchNorm output
    (the widest activation of the layer, 2D channels) was NOT kept for the backward up to round 4 (one fused multiply-add of the saved
    pre-normalisation activation, recomputed there); it IS kept since round 5: the recomputation was four small launches per row range on
    the critical path of a step that is bound by launch count, and the 16 MB per layer it saved (0.45 GB per step at 4 x 1024 keypoints)
    are nothing on a 288 GB part.  OG_TRAIN_KEEP_BN=0 = the old behaviour.  splits: see ConvReluBNTrain."""

    @staticmethod
    def forward(ctx, x, W0, b0, gamma, beta, running_mean, running_var, momentum, eps, splits, W3, b3):
        a = _gemm_fast(x.detach(), W0.detach().contiguous(), b0.detach(), relu=True)
        splits = tuple(splits) if splits else (a.shape[0],)
        if sum(splits) != a.shape[0]:
            raise ValueError("MLPBlockTrain: splits must add up to the rows of x")
        y = torch.empty_like(a)
        stats, r0 = [], 0
        for rows in splits:
            _, mean, invstd = batch_norm_train(a[r0:r0 + rows], gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps,
                                               return_stats=True, out=y[r0:r0 + rows])
            stats += [mean, invstd]
            r0 += rows
        z = _gemm_fast(y, W3.detach().contiguous(), b3.detach())
        ctx.keep_y = _os.environ.get("OG_TRAIN_KEEP_BN", "1") != "0"
        ctx.save_for_backward(x, W0, gamma, beta, a, W3, *stats, *((y,) if ctx.keep_y else ()))
        ctx.splits = splits
        return z

    @staticmethod
    def backward(ctx, dzo):
        lib = _lib.load()
        x, W0, gamma, beta, a, W3, *stats = ctx.saved_tensors
        T, C = a.shape
        dzo = dzo.detach().contiguous()
        g, bt = gamma.detach(), beta.detach()
        if ctx.keep_y:
            y = stats.pop()
        else:
            y = torch.empty_like(a)
            r0 = 0
            for i, rows in enumerate(ctx.splits):                  # y = (a - mean) invstd gamma + beta, recomputed
                sc = stats[2 * i + 1] * g
                torch.addcmul(bt - stats[2 * i] * sc, a[r0:r0 + rows], sc, out=y[r0:r0 + rows])
                r0 += rows
        dy, dW3, db3 = _conv_backward(y, W3.detach().contiguous(), dzo, True)
        del y
        dz = torch.empty_like(a)
        dgamma = dbeta = None
        r0 = 0
        for i, rows in enumerate(ctx.splits):
            mean, invstd = stats[2 * i], stats[2 * i + 1]
            dg = torch.empty(C, device=a.device, dtype=torch.float32)
            dbt = torch.empty(C, device=a.device, dtype=torch.float32)
            a_s, dy_s, dz_s = a[r0:r0 + rows], dy[r0:r0 + rows], dz[r0:r0 + rows]
            ws = _ws(a, rows, C)
            with torch.cuda.device(a.device):
                _lib.check(lib.og_batchnorm_train_backward(a_s.data_ptr(), a_s.stride(0), dy_s.data_ptr(), dy_s.stride(0), rows, C,
                                                           g.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1, dz_s.data_ptr(),
                                                           dz_s.stride(0), dg.data_ptr(), dbt.data_ptr(), ws.data_ptr(), _stream(a)),
                           "og_batchnorm_train_backward")
            dgamma = dg if dgamma is None else dgamma + dg
            dbeta = dbt if dbeta is None else dbeta + dbt
            r0 += rows
        del dy
        dx, dW0, db0 = _conv_backward(x.detach(), W0.detach().contiguous(), dz, ctx.needs_input_grad[0])
        return dx, dW0, db0, dgamma, dbeta, None, None, None, None, None, dW3, db3


def feed_forward_train_autograd(x: torch.Tensor, net_params, buffers, prefix: str = "", momentum: float = 0.1, eps: float = 1e-5,
                                splits=None) -> torch.Tensor:
    """FeedForwardNet (models/utils.py:48-58) in TRAINING mode with gradients: `net_params` maps the nn.Sequential parameter names
    (`{prefix}{3i}.weight|bias`, `{prefix}{3i+2}.weight|bias`) to tensors (requires_grad as wanted; conv weights [out, in, 1] or
    [out, in]), `buffers` the BatchNorm running statistics (updated in place).  x: token-major [T, C_in]."""
    n_conv = len({k for k in net_params if k.startswith(prefix) and k.endswith(".weight") and net_params[k].dim() >= 2})
    if n_conv == 2:                                               # the message MLP of a GNN layer: one node, BatchNorm output not kept
        W0, W3 = (net_params[f"{prefix}{j}.weight"] for j in (0, 3))
        bn = f"{prefix}2"
        return MLPBlockTrain.apply(x, W0.reshape(W0.shape[0], W0.shape[1]), net_params[f"{prefix}0.bias"], net_params[bn + ".weight"],
                                   net_params[bn + ".bias"], buffers[bn + ".running_mean"], buffers[bn + ".running_var"], momentum, eps,
                                   splits, W3.reshape(W3.shape[0], W3.shape[1]), net_params[f"{prefix}3.bias"])
    for i in range(n_conv):
        W = net_params[f"{prefix}{3 * i}.weight"]
        W = W.reshape(W.shape[0], W.shape[1])
        b = net_params[f"{prefix}{3 * i}.bias"]
        if i + 1 < n_conv:
            bn = f"{prefix}{3 * i + 2}"
            x = ConvReluBNTrain.apply(x, W, b, net_params[bn + ".weight"], net_params[bn + ".bias"], buffers[bn + ".running_mean"],
                                      buffers[bn + ".running_var"], momentum, eps, splits)
        else:
            x = Conv1x1.apply(x, W, b)
    return x


# ------------------------------------------------------------------------------------------------------------------
# training-mode softmax attention (attention matrix materialised like the reference, attention.py:8-19) and the score matrix
def _gemm_raw(dev, A, lda, sA, Bp, ldb, sB, Cp, ldc, sC, M, N, K, batch, scale=1.0):
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.og_gemm_nt(A, lda, sA, Bp, ldb, sB, Cp, ldc, sC, M, N, K, batch, None, 0, None, 0, None, float(scale),
                                  torch.cuda.current_stream(dev).cuda_stream), "og_gemm_nt")


def _gemm_km(dev, A, lda, sA, a_kmajor, Bp, ldb, sB, Cp, ldc, sC, M, N, K, batch, k_total=0, scale=1.0, a_colsum=False):
    """og_gemm_kmajor: C[z] = op(A[z]) B[z] with B stored [K][N] and, with a_kmajor, A stored [K][M] (the layouts of the backward products:
    no transposed copies)."""
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.og_gemm_kmajor(A, lda, sA, int(a_kmajor), Bp, ldb, sB, Cp, ldc, sC, M, N, K, batch, int(k_total), int(a_colsum), float(scale),
                                      torch.cuda.current_stream(dev).cuda_stream), "og_gemm_kmajor")


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


def _heads_first(x: torch.Tensor, H: int) -> torch.Tensor:
    """[B, N, H*d] -> [B*H, N, d] contiguous (pure data movement): one uniformly strided batch of B*H problems per GEMM launch."""
    B, N, D = x.shape
    return x.reshape(B, N, H, D // H).permute(0, 2, 1, 3).reshape(B * H, N, D // H).contiguous()


def _heads_last(x: torch.Tensor, B: int) -> torch.Tensor:
    """[B*H, N, d] -> [B, N, H*d] contiguous."""
    BH, N, d = x.shape
    H = BH // B
    return x.reshape(B, H, N, d).permute(0, 2, 1, 3).reshape(B, N, H * d).contiguous()


def _flash_backward_enabled(dh: int) -> bool:
    import os
    return os.environ.get("OG_TRAIN_FLASH_BWD", "1") != "0" and dh in (16, 32, 64)


_LOG2E = 1.4426950408889634


def _attention_rows(q2: torch.Tensor, k2: torch.Tensor, v2: torch.Tensor, Bz: int, nq: int, nk: int, H: int, qkv_of_one: Optional[torch.Tensor] = None):
    """Forward flash attention (the inference kernel, og_attention) on ROW-STRIDED fp32 operands: q2 [Bz * nq, D], k2, v2 [Bz * nk, D] may be
    column ranges of one wider matrix (the [tokens, 3D] output of the merged q | k | v projection).  The operands go to the kernel's (hi, lo)
    binary16 planes in one launch per source matrix (og_split_f16_rows: the q columns scaled by dh^-1/2 and log2(e) on the way -- the same
    two roundings as the tensor multiplications of ops.attention), the output planes come back as fp32 in one (og_merge_f16).
    qkv_of_one: the matrix q2, k2 and v2 are the three column thirds of (then ONE split launch).  -> out [Bz * nq, D], lse [Bz, H, nq]."""
    lib = _lib.load()
    D = q2.shape[1]
    dh = D // H
    dev = q2.device
    st = _stream(q2)
    f16 = torch.float16

    def split(src, scale_cols):
        rows, cols = src.shape
        hi = torch.empty(rows, cols, device=dev, dtype=f16)
        lo = torch.empty(rows, cols, device=dev, dtype=f16)
        _lib.check(lib.og_split_f16_rows(src.data_ptr(), src.stride(0), rows, cols, scale_cols, float(dh ** -0.5), _LOG2E, hi.data_ptr(), lo.data_ptr(),
                                         cols, st), "og_split_f16_rows")
        return hi, lo

    with torch.cuda.device(dev):
        if qkv_of_one is not None:
            hi, lo = split(qkv_of_one, D)
            ld = 3 * D
            planes = [(hi.data_ptr() + 2 * D * i, lo.data_ptr() + 2 * D * i, ld) for i in range(3)]
        else:
            qh, ql = split(q2, D)
            if k2.data_ptr() + 4 * D == v2.data_ptr() and k2.stride(0) == v2.stride(0) == 2 * D:      # k | v: the two halves of one matrix
                kvh, kvl = split(torch.as_strided(k2, (k2.shape[0], 2 * D), (2 * D, 1)), 0)
                planes = [(qh.data_ptr(), ql.data_ptr(), D), (kvh.data_ptr(), kvl.data_ptr(), 2 * D), (kvh.data_ptr() + 2 * D, kvl.data_ptr() + 2 * D, 2 * D)]
            else:
                kh, kl = split(k2, 0)
                vh, vl = split(v2, 0)
                planes = [(qh.data_ptr(), ql.data_ptr(), D), (kh.data_ptr(), kl.data_ptr(), D), (vh.data_ptr(), vl.data_ptr(), D)]
        oh = torch.empty(Bz * nq, D, device=dev, dtype=f16)
        ol = torch.empty_like(oh)
        lse = torch.empty(Bz, H, nq, device=dev, dtype=torch.float32)
        (qh_, ql_, ldq), (kh_, kl_, ldk), (vh_, vl_, ldv) = planes
        _lib.check(lib.og_attention(qh_, ql_, ldq, kh_, kl_, ldk, vh_, vl_, ldv, oh.data_ptr(), ol.data_ptr(), D, Bz, nq, nk, H, dh, lse.data_ptr(), st),
                   "og_attention")
        out = torch.empty(Bz * nq, D, device=dev, dtype=torch.float32)
        _lib.check(lib.og_merge_f16(oh.data_ptr(), ol.data_ptr(), oh.numel(), out.data_ptr(), st), "og_merge_f16")
    return out, lse


def _flash_backward_rows(q2, k2, v2, out, dout, lse, Bz, nq, nk, H, dq_out, dk_out, dv_out):
    """Flash backward on row-strided operands (og_attention_backward_ld): q2 [Bz * nq, D], k2, v2 [Bz * nk, D] fp32 views with unit column
    stride; dk_out, dv_out: views of the same kind the kernel writes into (column ranges of the [tokens, 3D] gradient matrix of the merged
    projection: no concatenation afterwards); dq_out receives the sum of the per-key-block partials."""
    lib = _lib.load()
    D = q2.shape[1]
    dh = D // H
    dev = q2.device
    st = _stream(q2)
    do = dout.detach().to(torch.float32).contiguous()
    delta = torch.empty(Bz * nq * H, device=dev, dtype=torch.float32)
    parts = lib.og_attention_backward_parts(nk)
    dq_part = torch.empty(parts, Bz * nq, D, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.og_attention_delta(do.data_ptr(), out.data_ptr(), Bz * nq, H, dh, delta.data_ptr(), st), "og_attention_delta")
        _lib.check(lib.og_attention_backward_ld(q2.data_ptr(), q2.stride(0), k2.data_ptr(), k2.stride(0), v2.data_ptr(), v2.stride(0), do.data_ptr(),
                                                lse.data_ptr(), delta.data_ptr(), Bz, nq, nk, H, dh, dh ** -0.5, dq_part.data_ptr(),
                                                dk_out.data_ptr(), dk_out.stride(0), dv_out.data_ptr(), dv_out.stride(0), st),
                   "og_attention_backward_ld")
    if parts > 1:
        torch.sum(dq_part, 0, out=dq_out)
    else:
        dq_out.copy_(dq_part[0])


def _flash_attention_backward(q32, k32, v32, out, dout, H, lse=None):
    """Flash backward (csrc/attention_train.hip): P is recomputed tile by tile in registers from q, k and the row log-sum-exp; nothing
    of size Nq x Nk is ever written.  Token-major contiguous tensors in and out ([B, N, D]): no head-major copies either.  -> dq, dk, dv.
    (The contiguous form of _flash_backward_rows; lse None = no forward kernel left one: an exact-fp32 pass computes it.)"""
    lib = _lib.load()
    B, Nq, D = q32.shape
    Nk = k32.shape[1]
    dh = D // H
    dev = q32.device
    if lse is None:
        lse = torch.empty(B, H, Nq, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.og_attention_train_lse(q32.data_ptr(), k32.data_ptr(), B, Nq, Nk, H, dh, dh ** -0.5, lse.data_ptr(), _stream(q32)),
                       "og_attention_train_lse")
    dq, dk, dv = torch.empty_like(q32), torch.empty_like(k32), torch.empty_like(v32)
    _flash_backward_rows(q32.reshape(B * Nq, D), k32.reshape(B * Nk, D), v32.reshape(B * Nk, D), out.reshape(B * Nq, D), dout.reshape(B * Nq, D), lse,
                         B, Nq, Nk, H, dq.reshape(B * Nq, D), dk.reshape(B * Nk, D), dv.reshape(B * Nk, D))
    return dq, dk, dv


class SoftmaxAttention(torch.autograd.Function):
    """out[b, i, h*d:(h+1)*d] = softmax_j(q_h[b, i] . k_h[b, j] / sqrt(d)) v_h[b, j]  on token-major q [B, Nq, D], k, v [B, Nk, D]
    (heads = contiguous channel blocks, attention_gnn.py:24-26).
    Forward: the flash kernel of the inference path (og_attention: split-f16, never materialises the attention matrix).  Backward:
    flash too (og_attention_train_lse + og_attention_backward, exact fp32): the attention matrix is recomputed tile by tile in
    registers -- the reference's autograd keeps B*H*Nq*Nk floats per layer alive (36 layers x 64 MB at 4 x 1024 keypoints); here only
    q, k, v and the output are saved and nothing of that size is ever written.  OG_TRAIN_FLASH_BWD=0: the attention matrix of the
    layer is recomputed as a whole (batched GEMM + og_softmax_rows) and the five products run as GEMM launches (other head sizes
    than 16 / 32 / 64 always do); OG_TRAIN_FLASH=0: round 2's materialising forward."""

    @staticmethod
    def forward(ctx, q, k, v, num_heads):
        import os
        from . import ops
        B, Nq, D = q.shape
        d = D // num_heads
        q32, k32, v32 = (t.detach().to(torch.float32).contiguous() for t in (q, k, v))
        ctx.heads = num_heads
        lse = None
        if os.environ.get("OG_TRAIN_FLASH", "1") != "0" and d in (16, 32, 64):
            out, lse = ops.attention(q32 * d ** -0.5, k32, v32, num_heads, return_lse=True)
        else:
            P, vh = SoftmaxAttention._probs(q32, k32, v32, num_heads)
            out = SoftmaxAttention._pv(P, vh, B)
        ctx.has_lse = lse is not None
        ctx.save_for_backward(q32, k32, v32, out, *(() if lse is None else (lse,)))   # `out` is also the saved input of the out-projection conv
        return out

    @staticmethod
    def _probs(q32, k32, v32, H):
        """P = softmax(scale * Q K^T) [B*H, Nq, r4(Nk)] and the heads-first operands."""
        lib = _lib.load()
        B, Nq, D = q32.shape
        Nk = k32.shape[1]
        d = D // H
        qh, kh, vh = (_heads_first(t, H) for t in (q32, k32, v32))                               # [B*H, N, d]
        dev = q32.device
        Z, Nk4 = B * H, _r4(Nk)
        P = torch.empty(Z, Nq, Nk4, device=dev, dtype=torch.float32)
        _gemm_raw(dev, qh.data_ptr(), d, Nq * d, kh.data_ptr(), d, Nk * d, P.data_ptr(), Nk4, Nq * Nk4, Nq, Nk, d, Z, d ** -0.5)
        with torch.cuda.device(dev):
            _lib.check(lib.og_softmax_rows(P.data_ptr(), Nk4, Z * Nq, Nk, torch.cuda.current_stream(dev).cuda_stream), "og_softmax_rows")
        return P, (qh, kh, vh)

    @staticmethod
    def _pv(P, heads, B):
        qh, kh, vh = heads
        Z, Nq, Nk4 = P.shape
        Nk, d = vh.shape[1], vh.shape[2]
        dev = P.device
        vp = _pad_rows(vh, Nk4)
        oh = torch.empty(Z, Nq, d, device=dev, dtype=torch.float32)
        _gemm_km(dev, P.data_ptr(), Nk4, Nq * Nk4, 0, vp.data_ptr(), d, Nk4 * d, oh.data_ptr(), d, Nq * d, Nq, d, Nk4, Z)      # P V
        return _heads_last(oh, B)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        q32, k32, v32, out, *rest = ctx.saved_tensors
        H = ctx.heads
        B, Nq, D = q32.shape
        Nk = k32.shape[1]
        dh = D // H
        if _flash_backward_enabled(dh):
            return (*_flash_attention_backward(q32, k32, v32, out, dout, H, rest[0] if ctx.has_lse else None), None)
        P, (qh, kh, vh) = SoftmaxAttention._probs(q32, k32, v32, H)
        Z, Nq, d = qh.shape
        Nk = kh.shape[1]
        dev = qh.device
        st = torch.cuda.current_stream(dev).cuda_stream
        Nk4 = _r4(Nk)
        doh = _heads_first(dout.detach().to(torch.float32), H)                                  # [Z, Nq, d]
        # every product below reads its operands as they lie (og_gemm_kmajor): no transposed copy of P, dS, dO, Q or K
        # dV = P^T dO
        dvh = torch.empty(Z, Nk, d, device=dev, dtype=torch.float32)
        _gemm_km(dev, P.data_ptr(), Nk4, Nq * Nk4, 1, doh.data_ptr(), d, Nq * d, dvh.data_ptr(), d, Nk * d, Nk, d, Nq, Z)
        # dP = dO V^T  ->  dS = scale * P o (dP - rowsum(dP o P))
        dS = torch.empty(Z, Nq, Nk4, device=dev, dtype=torch.float32)
        _gemm_raw(dev, doh.data_ptr(), d, Nq * d, vh.data_ptr(), d, Nk * d, dS.data_ptr(), Nk4, Nq * Nk4, Nq, Nk, d, Z)
        with torch.cuda.device(dev):
            _lib.check(lib.og_softmax_rows_backward(P.data_ptr(), dS.data_ptr(), Nk4, Z * Nq, Nk, d ** -0.5, st), "og_softmax_rows_backward")
        del P
        # dQ = dS K   (columns [Nk, Nk4) of dS are zero; K gets zero rows up to Nk4)
        kp = _pad_rows(kh, Nk4)
        dqh = torch.empty(Z, Nq, d, device=dev, dtype=torch.float32)
        _gemm_km(dev, dS.data_ptr(), Nk4, Nq * Nk4, 0, kp.data_ptr(), d, Nk4 * d, dqh.data_ptr(), d, Nq * d, Nq, d, Nk4, Z)
        # dK = dS^T Q
        dkh = torch.empty(Z, Nk, d, device=dev, dtype=torch.float32)
        _gemm_km(dev, dS.data_ptr(), Nk4, Nq * Nk4, 1, qh.data_ptr(), d, Nq * d, dkh.data_ptr(), d, Nk * d, Nk, d, Nq, Z)
        return _heads_last(dqh, B), _heads_last(dkh, B), _heads_last(dvh, B), None


class ProjectedAttention(torch.autograd.Function):
    """The q / k / v projections and the multi-head softmax attention of a GNN layer (attention_gnn.py:16-33) as ONE node:
    out = attention(xq Wq^T + bq, xkv Wk^T + bk, xkv Wv^T + bv).  Only the layer inputs and the attention output are saved (both are
    kept by their neighbours anyway); q, k, v -- three activations per layer in the reference's graph -- are RECOMPUTED in the backward
    by the same GEMM launch, then the flash backward and the conv backward run.  xkv None = self attention (one [T, 3D] projection
    launch).  xq [Bz * nq, D], xkv [Bz * nk, D] token-major; returns [Bz * nq, D].
    Round 5: q, k and v are never copied out of the projection matrix -- the attention kernels read column ranges of it (row stride 3D, or
    2D for the k | v matrix of a cross layer) and the flash backward writes dk and dv into column ranges of the gradient matrix the conv
    backward contracts; the stacked weights are built once per layer and forward call and kept for the backward."""

    @staticmethod
    def stacked(mha, is_self: bool):
        """The stacked projection weights of a layer from its CURRENT parameters -- self: (W [3D, D], b [3D]); cross: (Wq, bq, W [2D, D] of k | v,
        b [2D]).  superglue_forward_train builds them once per layer and forward call (the two propagations of a cross layer share them, the
        backward reuses what the forward saved).  Deliberately not kept across calls: a parameter changed through `.data` does not move its
        version counter, and a stale stack would be a silently wrong forward for the sake of 0.3 ms."""
        ps = (mha.in_proj_q.weight, mha.in_proj_q.bias, mha.in_proj_k.weight, mha.in_proj_k.bias, mha.in_proj_v.weight, mha.in_proj_v.bias)
        with torch.no_grad():
            Wq, bq, Wk, bk, Wv, bv = (p.detach().reshape(p.shape[0], -1) if p.dim() > 1 else p.detach() for p in ps)
            return (torch.cat([Wq, Wk, Wv]), torch.cat([bq, bk, bv])) if is_self else (Wq.contiguous(), bq, torch.cat([Wk, Wv]), torch.cat([bk, bv]))

    @staticmethod
    def forward(ctx, xq, xkv, Wq, bq, Wk, bk, Wv, bv, Bz, nq, nk, H, stack=None):
        xq = xq.detach().contiguous()
        xkv = None if xkv is None else xkv.detach().contiguous()
        Wq, bq, Wk, bk, Wv, bv = (t.detach() for t in (Wq, bq, Wk, bk, Wv, bv))
        D = Wq.shape[0]
        if xkv is None:
            Wc, bc = stack if stack is not None else (torch.cat([Wq, Wk, Wv]), torch.cat([bq, bk, bv]))
            qkv = _gemm_fast(xq, Wc, bc)                                                         # [T, 3D]
            out, lse = _attention_rows(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], Bz, nq, nk, H, qkv_of_one=qkv)
            saved = (xq, Wc, bc, out, lse)
        else:
            Wq_, bq, Wkv, bkv = stack if stack is not None else (Wq.contiguous(), bq, torch.cat([Wk, Wv]), torch.cat([bk, bv]))
            q = _gemm_fast(xq, Wq_, bq)
            kv = _gemm_fast(xkv, Wkv, bkv)                                                       # [Tk, 2D]
            out, lse = _attention_rows(q, kv[:, :D], kv[:, D:], Bz, nq, nk, H)
            saved = (xq, Wq_, bq, out, lse, xkv, Wkv, bkv)
        ctx.geom = (Bz, nq, nk, H, xkv is None)
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        Bz, nq, nk, H, is_self = ctx.geom
        if is_self:
            xq, Wc, bc, out, lse = ctx.saved_tensors
            D = Wc.shape[0] // 3
            qkv = _gemm_fast(xq, Wc, bc)
            dqkv = torch.empty_like(qkv)
            _flash_backward_rows(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, dout.reshape(Bz * nq, D), lse, Bz, nq, nk, H,
                                 dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
            del qkv
            dx, dW, db = _conv_backward(xq, Wc, dqkv, ctx.needs_input_grad[0])
            return (dx, None, dW[:D], db[:D], dW[D:2 * D], db[D:2 * D], dW[2 * D:], db[2 * D:], None, None, None, None, None)
        xq, Wq, bq, out, lse, xkv, Wkv, bkv = ctx.saved_tensors
        D = Wq.shape[0]
        q = _gemm_fast(xq, Wq, bq)
        kv = _gemm_fast(xkv, Wkv, bkv)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        _flash_backward_rows(q, kv[:, :D], kv[:, D:], out, dout.reshape(Bz * nq, D), lse, Bz, nq, nk, H, dq, dkv[:, :D], dkv[:, D:])
        del q, kv
        dxq, dWq, dbq = _conv_backward(xq, Wq, dq, ctx.needs_input_grad[0])
        dxkv, dWkv, dbkv = _conv_backward(xkv, Wkv, dkv, ctx.needs_input_grad[1])
        return (dxq, dxkv, dWq, dbq, dWkv[:D], dbkv[:D], dWkv[D:], dbkv[D:], None, None, None, None, None)


def _pad_rows(x: torch.Tensor, rows: int) -> torch.Tensor:
    """[Z, R, C] -> [Z, rows, C] with zero rows (only when R is not a multiple of 4: the K-contiguous operand's K)."""
    return x if x.shape[1] == rows else torch.nn.functional.pad(x, (0, 0, 0, rows - x.shape[1]))


def _bmm_nt(A: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
    """C[z] = A[z] Bm[z]^T for contiguous fp32 A [Z, M, K], Bm [Z, N, K], K % 4 == 0: one batched launch of the exact-fp32 MFMA GEMM."""
    Z, M, K = A.shape
    N = Bm.shape[1]
    C = torch.empty(Z, M, N, device=A.device, dtype=torch.float32)
    _gemm_raw(A.device, A.data_ptr(), K, M * K, Bm.data_ptr(), K, N * K, C.data_ptr(), N, M * N, M, N, K, Z)
    return C


def _bmm_nn(A: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
    """C[z] = A[z] Bm[z] for contiguous A [Z, M, K] (K % 4 == 0), Bm [Z, K, N]."""
    Z, M, K = A.shape
    N = Bm.shape[2]
    C = torch.empty(Z, M, N, device=A.device, dtype=torch.float32)
    _gemm_km(A.device, A.data_ptr(), K, M * K, 0, Bm.data_ptr(), N, K * N, C.data_ptr(), N, M * N, M, N, K, Z)
    return C


def _bmm_tn(A: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
    """C[z] = A[z]^T Bm[z] for contiguous A [Z, K, M], Bm [Z, K, N] (M, N multiples of 4; any K)."""
    Z, K, M = A.shape
    N = Bm.shape[2]
    C = torch.empty(Z, M, N, device=A.device, dtype=torch.float32)
    _gemm_km(A.device, A.data_ptr(), M, K * M, 1, Bm.data_ptr(), N, K * N, C.data_ptr(), N, M * N, M, N, K, Z)
    return C


class LinearAttentionCore(torch.autograd.Function):
    """out[z] = (Q' (K'^T V)) / (Q' . sum_j K'_j)  -- the reference's `linear_attention` (attention.py:29-40) on positive feature maps
    Q' [Z, Nq, F], K' [Z, Nk, F] and values V [Z, Nk, d] (Z = batch x heads, F % 4 == 0, d % 4 == 0).  O(N F d): no N x N matrix exists
    in either direction.  Every contraction is a batched launch of the exact-fp32 MFMA GEMM on the operands as they lie (og_gemm_nt /
    og_gemm_kmajor); the division and the two rank-one terms are tensor algebra.  Backward, with num = Q' kv, den = Q' z:
        dnum = dO / den,  dden = -sum_c(dO o out) / den,
        dQ' = dnum kv^T + dden z^T,   dkv = Q'^T dnum,   dz = Q'^T dden,
        dK' = V dkv^T + 1 dz^T,       dV = K' dkv."""

    @staticmethod
    def forward(ctx, fq, fk, v):
        fq, fk, v = (t.detach().to(torch.float32).contiguous() for t in (fq, fk, v))
        if fq.shape[2] % 4 or v.shape[2] % 4:
            raise ValueError("LinearAttentionCore: feature and value widths must be multiples of 4")
        kv = _bmm_tn(fk, v)                                         # [Z, F, d]:  kv[f][c] = sum_j k'[j][f] v[j][c]
        z = fk.sum(1)                                               # [Z, F]
        num = _bmm_nn(fq, kv)                                       # [Z, Nq, d]
        den = (fq * z[:, None, :]).sum(-1, keepdim=True)            # [Z, Nq, 1]
        out = num / den
        ctx.save_for_backward(fq, fk, v, kv, z, den, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        fq, fk, v, kv, z, den, out = ctx.saved_tensors
        dout = dout.detach().to(torch.float32).contiguous()
        dnum = (dout / den).contiguous()
        dden = -(dout * out).sum(-1, keepdim=True) / den            # [Z, Nq, 1]
        dfq = _bmm_nt(dnum, kv) + dden * z[:, None, :]
        dkv = _bmm_tn(fq, dnum)                                     # [Z, F, d]
        dz = (fq * dden).sum(1)                                     # [Z, F]
        dfk = _bmm_nt(v, dkv) + dz[:, None, :]
        dv = _bmm_nn(fk, dkv)
        return dfq, dfk, dv


def linear_attention_elu_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int) -> torch.Tensor:
    """attention = 'linear' under autograd (attention.py:22-40): phi(x) = elu(x) + 1 + 1e-6 per head, then LinearAttentionCore.
    q [B, Nq, D], k, v [B, Nk, D] token-major -> [B, Nq, D]."""
    B = q.shape[0]
    fq = torch.nn.functional.elu(_heads_first(q, num_heads)) + (1.0 + 1e-6)
    fk = torch.nn.functional.elu(_heads_first(k, num_heads)) + (1.0 + 1e-6)
    return _heads_last(LinearAttentionCore.apply(fq, fk, _heads_first(v, num_heads)), B)


def favor_relu_attention_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, projection: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """attention = 'favor_relu' under autograd (GeneralizedFavorAttention.randomized_kernel, attention.py:91-95, with the ReLU kernel
    and eps of __init__.py:19-25; one head, as the reference requires): phi(x) = relu(P (x d^-1/4)) + eps with the [2D, D] buffer P
    (the projection is a 1x1 conv without bias or weight gradient), then LinearAttentionCore."""
    B, Nq, D = q.shape
    Nk = k.shape[1]
    P = projection.detach().to(torch.float32).contiguous()
    zero = torch.zeros(P.shape[0], device=q.device, dtype=torch.float32)
    fq = torch.relu(Conv1x1.apply((q * D ** -0.25).reshape(B * Nq, D).contiguous(), P, zero)).reshape(B, Nq, -1) + eps
    fk = torch.relu(Conv1x1.apply((k * D ** -0.25).reshape(B * Nk, D).contiguous(), P, zero)).reshape(B, Nk, -1) + eps
    return LinearAttentionCore.apply(fq, fk, v)


class MatchingScores(torch.autograd.Function):
    """S[b] = g0[b] g1[b]^T * scale on token-major g0 [B, m, D], g1 [B, n, D] (superglue.py:81-86 with its D^-1/2 factor)."""

    @staticmethod
    def forward(ctx, g0, g1, scale):
        g0, g1 = g0.detach().to(torch.float32).contiguous(), g1.detach().to(torch.float32).contiguous()
        B, m, D = g0.shape
        n = g1.shape[1]
        S = torch.empty(B, m, n, device=g0.device, dtype=torch.float32)
        _gemm_raw(g0.device, g0.data_ptr(), D, m * D, g1.data_ptr(), D, n * D, S.data_ptr(), n, m * n, m, n, D, B, scale)
        ctx.save_for_backward(g0, g1)
        ctx.scale = float(scale)
        return S

    @staticmethod
    def backward(ctx, dS):
        g0, g1 = ctx.saved_tensors
        B, m, D = g0.shape
        n = g1.shape[1]
        dev = g0.device
        n4 = _r4(n)
        dSp = dS.detach().to(torch.float32)
        if n4 != n or not dSp.is_contiguous():
            buf = torch.zeros(B, m, n4, device=dev, dtype=torch.float32)
            buf[:, :, :n] = dSp
            dSp = buf
        g1p = _pad_rows(g1, n4)
        dg0 = torch.empty_like(g0)                                                   # dS g1
        _gemm_km(dev, dSp.data_ptr(), n4, m * n4, 0, g1p.data_ptr(), D, n4 * D, dg0.data_ptr(), D, m * D, m, D, n4, B, scale=ctx.scale)
        dg1 = torch.empty_like(g1)                                                   # dS^T g0
        _gemm_km(dev, dSp.data_ptr(), n4, m * n4, 1, g0.data_ptr(), D, m * D, dg1.data_ptr(), D, n * D, n, D, m, B, scale=ctx.scale)
        return dg0, dg1, None


# ------------------------------------------------------------------------------------------------------------------
# the whole module in training mode (superglue.py:29-72 under autograd)
def _pad_k4(x2d: torch.Tensor, W: torch.Tensor):
    """The exact-fp32 GEMM wants K % 4 == 0: zero-pad the 2 + side_info input columns of the first encoder conv (and its weight)."""
    K = x2d.shape[1]
    if K % 4 == 0:
        return x2d, W
    pad = 4 - K % 4
    return torch.nn.functional.pad(x2d, (0, pad)), torch.nn.functional.pad(W.reshape(W.shape[0], K), (0, pad))


def _seq_state(seq: torch.nn.Module):
    return dict(seq.named_parameters()), dict(seq.named_buffers())


def _mlp_train(x2d: torch.Tensor, seq: torch.nn.Module, momentum: float = 0.1, eps: float = 1e-5, splits=None) -> torch.Tensor:
    params, buffers = _seq_state(seq)
    W0 = params["0.weight"]
    x2d, W0p = _pad_k4(x2d, W0.reshape(W0.shape[0], W0.shape[1]))
    params = dict(params)
    params["0.weight"] = W0p
    y = feed_forward_train_autograd(x2d.contiguous(), params, buffers, "", momentum, eps, splits)
    for k, b in buffers.items():                      # nn.BatchNorm1d bookkeeping (unused by the arithmetic: momentum is fixed)
        if k.endswith("num_batches_tracked"):
            b.add_(len(splits) if splits else 1)
    return y


def _mlp_frozen(x2d: torch.Tensor, seq: torch.nn.Module, siren: bool = False, eps: float = 1e-5) -> torch.Tensor:
    """FeedForwardNet / FeedForwardNetSiren in EVAL mode under autograd (models/utils.py:23-58 with BatchNorm on its running
    statistics): the 1x1 convs on the exact-fp32 GEMM Function, ReLU / sin(30 x) and the per-channel affine of eval-mode
    BatchNorm as torch tensor algebra (differentiable w.r.t. the BatchNorm weight and bias too)."""
    params, buffers = _seq_state(seq)
    idx = sorted({int(k.split(".")[0]) for k in params if k.endswith(".weight") and params[k].dim() == 3})
    y = x2d
    for n_, i in enumerate(idx):
        W = params[f"{i}.weight"]
        W2 = W.reshape(W.shape[0], W.shape[1])
        if n_ == 0:
            y, W2 = _pad_k4(y, W2)
        y = Conv1x1.apply(y.contiguous(), W2, params[f"{i}.bias"])
        if n_ + 1 < len(idx):
            if siren:
                y = torch.sin(30.0 * y)
            else:
                y = torch.relu(y)
                j = i + 2                                       # Conv, ReLU, BatchNorm1d
                g = params[f"{j}.weight"] * torch.rsqrt(buffers[f"{j}.running_var"] + eps)
                y = (y - buffers[f"{j}.running_mean"]) * g + params[f"{j}.bias"]
    return y


def superglue_forward_train(model, data, frozen_bn: bool = False):
    """`SuperGlue.forward` in training mode (reference superglue.py:29-72 with the modules in train()): batch-statistics BatchNorm
    (running statistics updated like torch's), every 1x1 conv / attention product / score matrix on the exact-fp32 MFMA GEMM,
    softmax + its backward, BatchNorm + ReLU backward and the optimal-transport layer on HIP kernels, all wired through
    torch.autograd.Functions -- loss.backward() reaches every parameter.  Glue that stays torch tensor algebra: keypoint
    normalisation, concatenations, residual adds, the sigmoid mix, the elementwise feature maps of the linear attentions and sin(30 x).
    Supported: everything the inference path runs -- both encoders (FeedForwardNet, FeedForwardNetSiren), attention 'softmax' / 'linear' /
    'favor_relu', use_offset, residual, no_descriptors."""
    # frozen_bn: the module is in eval() and the caller wants gradients (fine-tuning on frozen BatchNorm statistics, saliency):
    # the reference's eval-mode forward is differentiable (superglue.py:29-72 under autograd), so is this one
    mlp = ((lambda x_, seq_, splits_=None: _mlp_frozen(x_, seq_)) if frozen_bn
           else (lambda x_, seq_, splits_=None: _mlp_train(x_, seq_, splits=splits_)))
    D, H = model.descriptor_dim, model.num_heads
    k0, k1 = data["keypoints0"], data["keypoints1"]
    d0, d1 = data["local_descriptors0"], data["local_descriptors1"]                # [B, N, D] token-major as they arrive
    s0, s1 = data["side_info0"], data["side_info1"]
    for t in (k0, k1, d0, d1, s0, s1):
        if not t.is_cuda:
            raise RuntimeError("openglue_amd.SuperGlue: inputs must be on the MI355X; there is no CPU fallback")
    B, m, n = k0.shape[0], k0.shape[1], k1.shape[1]
    from .superglue import _get_wh

    T0, T1 = B * m, B * n

    def encoder_input(k, s, wh):
        # the divisor as a DEVICE tensor built by fill kernels: torch.tensor([...], device=...) is a host-to-device copy, which a stream
        # capture refuses (GraphedTrainStep); dividing by Python scalars instead multiplies by a rounded reciprocal -- one ulp away from
        # the reference's tensor / tensor division, which the Siren encoder's sin(30 x) layers then amplify
        wh1 = torch.stack([torch.full((), float(wh[0]) - 1.0, device=k.device, dtype=torch.float32),
                           torch.full((), float(wh[1]) - 1.0, device=k.device, dtype=torch.float32)])
        kn = 2.0 * k.to(torch.float32) / wh1 - 1.0                                  # superglue.py:74-78
        inp = torch.cat([kn, s.to(torch.float32).reshape(k.shape[0], k.shape[1], -1)], dim=-1)
        return inp.reshape(-1, inp.shape[-1])

    # both images through the keypoint encoder as ONE token matrix (the reference calls it twice: BatchNorm on the two row ranges)
    inp01 = torch.cat([encoder_input(k0, s0, _get_wh(data, 0)), encoder_input(k1, s1, _get_wh(data, 1))])
    enc = model.positional_encoding.encoder
    pe01 = _mlp_frozen(inp01, enc, True) if model.siren else mlp(inp01, enc, (T0, T1))   # FeedForwardNetSiren has no BatchNorm (models/utils.py:32-45)
    d01 = torch.cat([d0.to(torch.float32).reshape(T0, D), d1.to(torch.float32).reshape(T1, D)])
    d0f, d1f = d01[:T0], d01[T0:]
    xs = pe01 if model.no_descriptors else d01 + pe01
    x0, x1 = xs[:T0], xs[T0:]

    def conv(x2d, c):
        return Conv1x1.apply(x2d.contiguous(), c.weight.reshape(c.weight.shape[0], c.weight.shape[1]), c.bias)

    def conv_many(x2d, *cs):
        """Several 1x1 convs of the SAME input as one GEMM (weights stacked along the output channels; autograd splits the gradient):
        at 4096 tokens a [T, 256] x [256, 256] launch is 64 workgroups on 256 CUs -- merged launches fill the chip."""
        W = torch.cat([c.weight.reshape(c.weight.shape[0], c.weight.shape[1]) for c in cs])
        b = torch.cat([c.bias for c in cs])
        return Conv1x1.apply(x2d.contiguous(), W, b).split([c.weight.shape[0] for c in cs], dim=1)

    def attend(mha, q3, k3, v3):
        if model.linear_attention:
            return linear_attention_elu_train(q3, k3, v3, H)
        if model.favor_relu:
            return favor_relu_attention_train(q3, k3, v3, mha.attention_func.projection_matrix)
        return SoftmaxAttention.apply(q3, k3, v3, H)

    def finish(layer, xq, msg, splits=None):                                       # attention_gnn.py:51-55 (BatchNorm statistics per call)
        y = torch.cat([xq - msg if model.use_offset else xq, msg], dim=-1)
        return xq + mlp(y, layer.module.fc, splits)

    import os
    fused_attn = (not model.linear_attention and not model.favor_relu and _flash_backward_enabled(D // H)
                  and os.environ.get("OG_TRAIN_FLASH", "1") != "0")

    def w2(c):
        return c.weight.reshape(c.weight.shape[0], c.weight.shape[1])

    stacks = {}                                                                    # per forward call: (layer, form) -> stacked projection weights

    def proj_attend(mha, xq, xkv, Bz, nq, nk):                                     # projections + attention, q / k / v not kept
        key = (id(mha), xkv is None)
        if key not in stacks:
            stacks[key] = ProjectedAttention.stacked(mha, xkv is None)
        return ProjectedAttention.apply(xq, xkv, w2(mha.in_proj_q), mha.in_proj_q.bias, w2(mha.in_proj_k), mha.in_proj_k.bias,
                                        w2(mha.in_proj_v), mha.in_proj_v.bias, Bz, nq, nk, H, stacks[key])

    for li, layer in enumerate(model.attention_gnn.layers):
        mha = layer.module.mha
        if li % 2 == 0:                                                            # self (attention_gnn.py:63-66): the two images are independent:
            xs = torch.cat([x0, x1])                                               # one token matrix, every conv in one launch, BatchNorm on the
            if fused_attn and m == n:                                              # two row ranges like the reference's two calls
                o = proj_attend(mha, xs, None, 2 * B, m, m)
            elif fused_attn:
                o = torch.cat([proj_attend(mha, xs[:T0], None, B, m, m), proj_attend(mha, xs[T0:], None, B, n, n)])
            else:
                q, k, v = conv_many(xs, mha.in_proj_q, mha.in_proj_k, mha.in_proj_v)
                if m == n:
                    o = attend(mha, q.reshape(2 * B, m, D), k.reshape(2 * B, m, D), v.reshape(2 * B, m, D)).reshape(T0 + T1, D)
                else:
                    o = torch.cat([attend(mha, q[r0:r1].reshape(B, nx, D), k[r0:r1].reshape(B, nx, D), v[r0:r1].reshape(B, nx, D)).reshape(r1 - r0, D)
                                   for r0, r1, nx in ((0, T0, m), (T0, T0 + T1, n))])
            xs = finish(layer, xs, conv(o, mha.out_proj), (T0, T1))
            x0, x1 = xs[:T0], xs[T0:]
        elif fused_attn:                                                           # cross: image 1 sees the UPDATED image 0 (:74-77)
            x0 = finish(layer, x0, conv(proj_attend(mha, x0, x1, B, m, n), mha.out_proj))
            x1 = finish(layer, x1, conv(proj_attend(mha, x1, x0, B, n, m), mha.out_proj))
        else:
            q1, k1, v1 = conv_many(x1, mha.in_proj_q, mha.in_proj_k, mha.in_proj_v)   # x1 is unchanged until the second propagate
            q0 = conv(x0, mha.in_proj_q)
            o0 = attend(mha, q0.reshape(B, m, D), k1.reshape(B, n, D), v1.reshape(B, n, D))
            x0 = finish(layer, x0, conv(o0.reshape(T0, D), mha.out_proj))
            k0, v0 = conv_many(x0, mha.in_proj_k, mha.in_proj_v)
            o1 = attend(mha, q1.reshape(B, n, D), k0.reshape(B, m, D), v0.reshape(B, m, D))
            x1 = finish(layer, x1, conv(o1.reshape(T1, D), mha.out_proj))
    g01 = conv(torch.cat([x0, x1]), model.linear_proj)                             # superglue.py:58, both images in one launch
    g0, g1 = g01[:T0], g01[T0:]
    if model.residual:
        alpha = torch.sigmoid(model.mix_coefs).reshape(1, D)
        g0 = alpha * g0 + (1.0 - alpha) * d0f
        g1 = alpha * g1 + (1.0 - alpha) * d1f
    S = MatchingScores.apply(g0.reshape(B, m, D), g1.reshape(B, n, D), D ** -0.5)
    otp = model.config["otp"]
    scores = SinkhornOT.apply(S, model.dustbin_score, int(otp["num_iters"]), float(otp["reg"]))
    return {"context_descriptors0": g0.reshape(B, m, D).transpose(1, 2), "context_descriptors1": g1.reshape(B, n, D).transpose(1, 2),
            "scores": scores}

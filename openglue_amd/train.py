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

Not yet built (stated in DESIGN.md): backward of attention and of the split-f16 GNN GEMMs; `SuperGlue.forward` therefore still
refuses `train()` mode.
"""
from __future__ import annotations

import torch

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
        z = float(dustbin_score.detach())
        with torch.cuda.device(S.device):
            _lib.check(lib.og_sinkhorn_train_forward(Sp.data_ptr(), lds, z, B, m, n, int(num_iters), float(reg), scores.data_ptr(),
                                                     ws.data_ptr() + off, _stream(S)), "og_sinkhorn_train_forward")
        ctx.save_for_backward(Sp)
        ctx.ws, ctx.off, ctx.args = ws, off, (B, m, n, lds, int(num_iters), float(reg), z)
        ctx.dustbin_meta = (dustbin_score.dtype, dustbin_score.shape)
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        lib = _lib.load()
        (Sp,) = ctx.saved_tensors
        B, m, n, lds, iters, reg, z = ctx.args
        g = grad_scores.detach().to(torch.float32).contiguous()
        dS = torch.empty(B, m, lds, device=Sp.device, dtype=torch.float32)
        dz = torch.zeros(1, device=Sp.device, dtype=torch.float32)
        with torch.cuda.device(Sp.device):
            _lib.check(lib.og_sinkhorn_backward(Sp.data_ptr(), lds, z, B, m, n, iters, reg, g.data_ptr(), ctx.ws.data_ptr() + ctx.off,
                                                dS.data_ptr(), lds, dz.data_ptr(), _stream(Sp)), "og_sinkhorn_backward")
        dtype, shape = ctx.dustbin_meta
        return dS[:, :, :n], dz.reshape(shape).to(dtype), None, None


def matching_log_probs(S: torch.Tensor, dustbin_score: torch.Tensor, num_iters: int, reg: float = 1.0) -> torch.Tensor:
    """Differentiable `scores` [B, m+1, n+1] from the raw score matrix S [B, m, n] (superglue.py:88-111)."""
    return SinkhornOT.apply(S, dustbin_score, num_iters, reg)


def batch_norm_train(x: torch.Tensor, weight, bias, running_mean, running_var, momentum: float = 0.1, eps: float = 1e-5,
                     return_stats: bool = False):
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
    y = torch.empty(T, C, device=x.device, dtype=torch.float32)
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


def _transpose_pad(x: torch.Tensor) -> torch.Tensor:
    """[R, C] -> [C, round_up(R, 4)] (zero tail): both operands of the weight-gradient GEMM must be K-contiguous, K = R."""
    lib = _lib.load()
    R, C = x.shape
    R4 = (R + 3) // 4 * 4
    out = torch.zeros(C, R4, device=x.device, dtype=torch.float32) if R4 != R else torch.empty(C, R, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(lib.og_transpose_f32(x.data_ptr(), x.stride(0), R, C, out.data_ptr(), R4, _stream(x)), "og_transpose_f32")
    return out


def _conv_backward(x: torch.Tensor, W: torch.Tensor, dz: torch.Tensor, need_dx: bool):
    """1x1 conv on token rows, z = x W^T + b:  dx = dz W,  dW = dz^T x,  db = column sums of dz -- all on HIP kernels."""
    from . import ops
    lib = _lib.load()
    T, Cout = dz.shape
    dx = ops.gemm_nt(dz, _transpose_pad(W)) if need_dx else None              # [T, Cout] x [Cin, Cout]^T
    dW = ops.gemm_nt(_transpose_pad(dz), _transpose_pad(x))                   # [Cout, T] x [Cin, T]^T
    db = torch.empty(Cout, device=dz.device, dtype=torch.float32)
    ws = _ws(dz, T, Cout)
    with torch.cuda.device(dz.device):
        _lib.check(lib.og_colsum_f32(dz.data_ptr(), dz.stride(0), T, Cout, db.data_ptr(), ws.data_ptr(), _stream(dz)), "og_colsum_f32")
    return dx, dW, db


class Conv1x1(torch.autograd.Function):
    """y = x W^T + b on token-major x [T, Cin] (nn.Conv1d(kernel_size=1) of the reference on [B, Cin, N])."""

    @staticmethod
    def forward(ctx, x, W, b):
        from . import ops
        ctx.save_for_backward(x, W)
        return ops.gemm_nt(x.detach(), W.detach().contiguous(), b.detach())

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx, dW, db = _conv_backward(x.detach(), W.detach().contiguous(), dy.detach().contiguous(), ctx.needs_input_grad[0])
        return dx, dW, db


class ConvReluBNTrain(torch.autograd.Function):
    """The hidden block of FeedForwardNet in training mode (models/utils.py:52-56): y = BatchNorm_train(relu(x W^T + b)); running
    statistics are updated in place in forward."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, running_mean, running_var, momentum, eps):
        from . import ops
        a = ops.gemm_nt(x.detach(), W.detach().contiguous(), b.detach(), relu=True)
        y, mean, invstd = batch_norm_train(a, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps, return_stats=True)
        ctx.save_for_backward(x, W, gamma, a, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, W, gamma, a, mean, invstd = ctx.saved_tensors
        T, C = a.shape
        dy = dy.detach().contiguous()
        dz = torch.empty_like(a)
        dgamma = torch.empty(C, device=a.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=a.device, dtype=torch.float32)
        ws = _ws(a, T, C)
        with torch.cuda.device(a.device):
            _lib.check(lib.og_batchnorm_train_backward(a.data_ptr(), a.stride(0), dy.data_ptr(), dy.stride(0), T, C, gamma.detach().data_ptr(),
                                                       mean.data_ptr(), invstd.data_ptr(), 1, dz.data_ptr(), dz.stride(0), dgamma.data_ptr(),
                                                       dbeta.data_ptr(), ws.data_ptr(), _stream(a)), "og_batchnorm_train_backward")
        dx, dW, db = _conv_backward(x.detach(), W.detach().contiguous(), dz, ctx.needs_input_grad[0])
        return dx, dW, db, dgamma, dbeta, None, None, None, None


def feed_forward_train_autograd(x: torch.Tensor, net_params, buffers, prefix: str = "", momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """FeedForwardNet (models/utils.py:48-58) in TRAINING mode with gradients: `net_params` maps the nn.Sequential parameter names
    (`{prefix}{3i}.weight|bias`, `{prefix}{3i+2}.weight|bias`) to tensors (requires_grad as wanted; conv weights [out, in, 1] or
    [out, in]), `buffers` the BatchNorm running statistics (updated in place).  x: token-major [T, C_in]."""
    n_conv = len({k for k in net_params if k.startswith(prefix) and k.endswith(".weight") and net_params[k].dim() >= 2})
    for i in range(n_conv):
        W = net_params[f"{prefix}{3 * i}.weight"]
        W = W.reshape(W.shape[0], W.shape[1])
        b = net_params[f"{prefix}{3 * i}.bias"]
        if i + 1 < n_conv:
            bn = f"{prefix}{3 * i + 2}"
            x = ConvReluBNTrain.apply(x, W, b, net_params[bn + ".weight"], net_params[bn + ".bias"], buffers[bn + ".running_mean"],
                                      buffers[bn + ".running_var"], momentum, eps)
        else:
            x = Conv1x1.apply(x, W, b)
    return x

"""Training slice: the optimal-transport layer as a torch.autograd.Function backed by HIP kernels.

`matching_log_probs(S, dustbin_score, num_iters, reg)` is a differentiable drop-in for `SuperGlue.get_matching_probs`
(reference superglue.py:88-111 -> log_otp_solver, optimal_transport.py:20-28): forward = og_sinkhorn_train_forward (keeps
the dual trajectory), backward = og_sinkhorn_backward (unrolled iterations, reverse order) -- so the NLL of
utils/losses.py:7-53 computed on its result back-propagates into the score matrix S and into `dustbin_score` without any
torch math on the way.  PyTorch is used for what it is here: autograd bookkeeping, device memory, the current stream.

Not yet built (stated in DESIGN.md): backward of the GNN / encoder GEMMs and of attention, train-mode BatchNorm statistics;
`SuperGlue.forward` therefore still refuses `train()` mode.
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

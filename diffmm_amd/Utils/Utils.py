"""Loss functions of the reference's Utils/Utils.py:45-98 with the same signatures, on HIP kernels.

Each is a torch.autograd.Function: forward and backward both call libdiffmm_hip (no eager-PyTorch math), so code
written against the reference -- `InfoNCE(v1, v2, idx, temp)`, `bpr_loss(u, p, n)`, `l2_reg_loss(reg, [..], device)`
followed by `.backward()` -- keeps working.  The fused training path (engine.JointEngine) does not go through these
wrappers; it batches all terms of a step into single launches.
"""
from __future__ import annotations

from typing import List

import torch
from torch import Tensor

from .. import ops


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v1: Tensor, v2: Tensor, idx: Tensor, temperature: float, b_cos: bool):
        v1c, v2c, idx32 = v1.contiguous(), v2.contiguous(), idx.to(torch.int32).contiguous()
        B, D, dev = idx32.shape[0], v1c.shape[1], v1c.device
        slots = ops.Slots([(v1c, idx32, None), (v2c, idx32, None)])
        terms = ops.make_terms([(0, 1, float(temperature), 1.0)])
        zhat, nrm = torch.empty(2, B, D, device=dev), torch.empty(2, B, device=dev)
        lse, rl, out = torch.empty(1, B, device=dev), torch.empty(1, B, device=dev), torch.empty(1, device=dev)
        if b_cos:
            ops.gather_normalize(slots, B, D, zhat, nrm)
        else:  # raw dot products (Utils.py:71 skipped): the gathered rows enter the logits as they are
            ops.gather_rows(slots, B, D, None, zhat)
        ops.infonce_fwd(zhat, terms, 1, B, D, lse, rl)
        ops.segment_sum(rl, 1, B, 1.0, out)
        ctx.save_for_backward(v1c, v2c, idx32, zhat, nrm, lse)
        ctx.temperature, ctx.b_cos = float(temperature), bool(b_cos)
        return out[0]

    @staticmethod
    def backward(ctx, gout: Tensor):
        v1c, v2c, idx32, zhat, nrm, lse = ctx.saved_tensors
        B, D = idx32.shape[0], v1c.shape[1]
        g1, g2 = torch.zeros_like(v1c), torch.zeros_like(v2c)
        dz = torch.zeros_like(zhat)
        terms = ops.make_terms([(0, 1, ctx.temperature, 1.0)])
        ops.infonce_bwd(zhat, lse, terms, 1, B, D, dz)
        slots = ops.Slots([(v1c, idx32, g1), (v2c, idx32, g2)])
        if ctx.b_cos:
            ops.scatter_normalize_bwd(slots, B, D, zhat, nrm, dz)
        else:
            ops.scatter_add_rows(slots, B, D, None, dz)
        return g1 * gout, g2 * gout, None, None, None


def InfoNCE(batch_view1: Tensor, batch_view2: Tensor, idx: Tensor, temperature: float, b_cos: bool = True) -> Tensor:
    """-mean(diag(log_softmax(normalize(v1[idx]) normalize(v2[idx])^T / temperature))) (Utils.py:57-75); b_cos=False skips the
    normalisation (Utils.py:71-72)."""
    if batch_view1.shape[1] != batch_view2.shape[1]:
        raise ValueError(f"InfoNCE expected the same shape for two views. But got view1.shape={batch_view1[idx].shape} and "
                         f"view2.shape={batch_view2[idx].shape}.")
    return _InfoNCE.apply(batch_view1, batch_view2, idx, temperature, b_cos)


class _BPR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u: Tensor, p: Tensor, n: Tensor):
        B, dev = u.shape[0], u.device
        table = torch.cat([u, p, n], dim=0).contiguous()
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        dtab, rl, out = torch.zeros_like(table), torch.empty(B, device=dev), torch.empty(1, device=dev)
        ops.bpr_fwd_bwd(table, ar, ar, ar + B, B, rl, dtab)
        ops.segment_sum(rl, 1, B, 1.0, out)
        ctx.save_for_backward(dtab)
        ctx.B = B
        return out[0]

    @staticmethod
    def backward(ctx, gout: Tensor):
        (dtab,), B = ctx.saved_tensors, ctx.B
        g = dtab * gout
        return g[:B], g[B:2 * B], g[2 * B:]


def bpr_loss(user_emb: Tensor, pos_item_emb: Tensor, neg_item_embs: Tensor) -> Tensor:
    """mean(-log(1e-5 + sigmoid(u.p - u.n))) (Utils.py:78-98)."""
    return _BPR.apply(user_emb, pos_item_emb, neg_item_embs)


class _SumSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, reg: float):
        xc = x.contiguous()
        out = torch.empty(1, device=x.device)
        ops.sumsq(xc.view(-1), float(reg), out)
        ctx.save_for_backward(xc)
        ctx.reg = float(reg)
        return out[0]

    @staticmethod
    def backward(ctx, gout: Tensor):
        (xc,) = ctx.saved_tensors
        return xc * (2.0 * ctx.reg) * gout, None


def l2_reg_loss(reg: float, embeddings: List[Tensor], device: torch.device) -> Tensor:
    """reg * sum_k sum(emb_k ** 2) (Utils.py:45-54)."""
    total = torch.zeros((), device=device)
    for e in embeddings:
        total = total + _SumSq.apply(e, reg)
    return total

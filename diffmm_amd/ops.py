"""Tensor-level wrappers over the C ABI (one Python function per exported kernel family).

Every function launches asynchronously on torch's current stream and returns torch tensors that alias
caller-visible HBM buffers.  No CPU path exists: non-CUDA tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L
from .graph import Adjacency

Tensor = torch.Tensor


def _f32c(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _i32c(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous int32 tensor")
    return t


# ------------------------------------------------------------------------------------------------
# SpMM
# ------------------------------------------------------------------------------------------------
class SpmmOp:
    """One problem of a batched SpMM launch; see include/diffmm_hip.h (dmm_spmm_problem)."""

    def __init__(self, adj: Adjacency, x_lo: Tensor, y: Optional[Tensor] = None, x_hi: Optional[Tensor] = None, split: Optional[int] = None,
                 alpha: float = 1.0, c1: Optional[Tensor] = None, beta: float = 0.0, c2: Optional[Tensor] = None, gamma: float = 0.0,
                 y_pert: Optional[Tensor] = None, noise: Optional[Tensor] = None, noise_degree: float = 0.0, seed: int = 0, offset: int = 0,
                 y_mean: Optional[Tensor] = None, m1: Optional[Tensor] = None, m2: Optional[Tensor] = None, partial: Optional[Tensor] = None,
                 raw: Optional[Tensor] = None, raw_from: int = 0, rng_rows: Optional[tuple] = None, y16: Optional[Tensor] = None,
                 y_pert16: Optional[Tensor] = None):
        """raw / raw_from: user-sharded launch -- rows >= raw_from park their raw partial sums in raw [n - raw_from, D] (all-reduce it over
        the replicas, then spmm_epilogue_rows finishes those rows).  rng_rows = (split, add_lo, add_hi): global row ids of the Philox draws."""
        D = adj.D
        n = adj.n
        if x_hi is None:
            split = 1 << 30
            if tuple(x_lo.shape) != (n, D):
                raise ValueError(f"SpMM operand must be [{n}, {D}], got {tuple(x_lo.shape)}")
        else:
            split = x_lo.shape[0] if split is None else split
            if x_lo.shape[0] != split or x_lo.shape[0] + x_hi.shape[0] != n or x_lo.shape[1] != D or x_hi.shape[1] != D:
                raise ValueError("split SpMM operand: row counts must add up to the adjacency size")
        xdt = torch.bfloat16 if adj.vec == 8 else torch.float32  # the adjacency's task list fixes the operand type
        for nm, t in (("x_lo", x_lo), ("x_hi", x_hi)):
            if t is not None and (t.dtype != xdt or not t.is_contiguous()):
                raise ValueError(f"SpMM {nm}: this adjacency is planned for contiguous {xdt} operands, got {t.dtype}")
        # (with `raw`, rows >= raw_from are parked there and never touch the outputs / epilogue operands: those may end after raw_from rows --
        # an item-sharded replica's tensors hold [own users ; own item slice], the adjacency's rows are [own users ; all items])
        rows_ok = (lambda r: r == n) if raw is None else (lambda r: r >= raw_from)
        for nm, t in (("y", y), ("c1", c1), ("c2", c2), ("y_pert", y_pert), ("noise", noise), ("y_mean", y_mean), ("m1", m1), ("m2", m2)):
            if t is not None:
                _f32c(t, nm)
                if t.dim() != 2 or t.shape[1] != D or not rows_ok(t.shape[0]):
                    raise ValueError(f"SpMM {nm} must be [{n}, {D}], got {tuple(t.shape)}")
        for nm, t in (("y16", y16), ("y_pert16", y_pert16)):
            if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != D or not rows_ok(t.shape[0])):
                raise ValueError(f"SpMM {nm} must be a contiguous bfloat16 [{n}, {D}] tensor")
        self.keep = (adj, x_lo, x_hi, y, c1, c2, y_pert, noise, y_mean, m1, m2, y16, y_pert16)
        p = L.SpmmProblem()
        p.tasks, p.col, p.val = L.ptr(adj.tasks), L.ptr(adj.col), L.ptr(adj.val)
        p.x_lo, p.x_hi, p.split = L.ptr(x_lo), L.ptr(x_hi), split
        p.y, p.c1, p.c2 = L.ptr(y), L.ptr(c1), L.ptr(c2)
        p.alpha, p.beta, p.gamma = alpha, beta, gamma
        p.y_pert, p.noise, p.noise_degree, p.seed, p.offset = L.ptr(y_pert), L.ptr(noise), noise_degree, seed, offset
        p.y_mean, p.m1, p.m2 = L.ptr(y_mean), L.ptr(m1), L.ptr(m2)
        p.y16, p.y_pert16 = L.ptr(y16), L.ptr(y_pert16)
        p.n_rows = adj.n
        p.n_partial_blocks, p.n_fix = adj.n_partial_blocks, (int(adj.fix.shape[0]) if adj.fix is not None else 0)
        # (a persistent adjacency -- rebuilt in place, its launches replayed from recorded plans -- may have NO giant row when a plan is recorded and some
        # after a later rebuild: the plan must hold the fix buffer's address even while the current slice of it is empty, and Tensor.data_ptr() of an
        # empty tensor is null)
        p.fix = (adj.fix.untyped_storage().data_ptr() + adj.fix.storage_offset() * adj.fix.element_size()) if adj.fix is not None else None
        p.partial = L.ptr(partial) if partial is not None else None
        self.adj = adj
        p.n_tasks = adj.n_tasks
        p.n_head_tasks = adj.n_head_tasks
        if raw is not None:
            _f32c(raw, "raw")
            if not 0 <= raw_from <= n or tuple(raw.shape) != (n - raw_from, D):
                raise ValueError(f"SpMM raw must be [{n - raw_from}, {D}], got {tuple(raw.shape)}")
            p.raw, p.raw_from = L.ptr(raw), raw_from
            self.keep += (raw,)
        if rng_rows is not None:
            p.rng_row_split, p.rng_row_add_lo, p.rng_row_add_hi = (int(v) for v in rng_rows)
        self.p, self.D = p, D


# DMM_SPMM_FOLD=1: the giant rows are finished by the last-arriving segment workgroup of the main launch (dmm_spmm_problem.fix_count) instead of by the
# second launch (spmm_fixup_kernel).  Bit-identical, and measured in round 4: SLOWER at TikTok size -- the release / add / acquire of the hand-off sits
# on the launch's critical path (joint phase 7.92 -> 8.10 ms for the six launches per iteration it saves) -- so the second launch stays the default.
_SPMM_FOLD = os.environ.get("DMM_SPMM_FOLD", "0") != "0"


def spmm_batched(ops: Sequence[SpmmOp]):
    if not 1 <= len(ops) <= L.SPMM_MAX_BATCH:
        raise ValueError("1..4 problems per launch")
    D = ops[0].D
    uses = {}
    for o in ops:  # giant-row scratch: one buffer per concurrent use of an adjacency, owned by the adjacency
        if (o.adj.n_partial_blocks or o.adj.persistent) and not o.p.partial:
            i = uses.get(id(o.adj), 0)
            uses[id(o.adj)] = i + 1
            o.p.partial = L.ptr(o.adj.partial_buffer(i))
            if _SPMM_FOLD:  # giant rows finished by the last-arriving segment workgroup of the main launch: no fix-up launch
                o.p.fix_count = L.ptr(o.adj.fix_counter(i))
    if any(o.adj.vec != ops[0].adj.vec for o in ops):
        raise ValueError("one launch takes fp32-operand or bf16-operand problems, not both")
    arr = (L.SpmmProblem * len(ops))(*[o.p for o in ops])
    L.call("dmm_spmm_csr_bf16" if ops[0].adj.vec == 8 else "dmm_spmm_csr_f32", arr, len(ops), D, L.stream())
    return arr  # the launch's argument block (plans patch the Philox offsets in place)


def spmm_epilogue_rows(arr, n: int, D: int) -> None:
    """Second half of a user-sharded SpMM launch (`arr` = what spmm_batched returned): epilogue of the raw rows after their all-reduce."""
    L.call("dmm_spmm_epilogue_rows", arr, n, D, L.stream())


def spmm(adj: Adjacency, x: Tensor, y: Optional[Tensor] = None, **kw) -> Tensor:
    """Y = A X (+ epilogues, see SpmmOp) as a single-problem launch."""
    if y is None:
        y = torch.empty((adj.n, adj.D), dtype=torch.float32, device=x.device)
    spmm_batched([SpmmOp(adj, x, y=y, **kw)])
    return y


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm(A: Tensor, B: Tensor, out: Optional[Tensor] = None, **kw) -> Tensor:
    """out[M,N] = post(act(alpha * op(A) op(B) + bias)); op(A) = A^T if trans_a (A stored [K, M]); op(B) = B^T if trans_b
    (B stored [N, K], the nn.Linear weight layout).  A, B, out, residual may be row-strided views (last dim contiguous)."""
    g, out = _gemm_args(A, B, out, **kw)
    L.call("dmm_gemm", C.byref(g), L.stream())
    return out


def gemm_batched(problems: Sequence[dict]) -> None:
    """Up to 4 problems (dicts of gemm() arguments incl. A, B, out) of one operand layout and precision in ONE launch."""
    args = [_gemm_args(**p)[0] for p in problems]
    arr = (L.GemmArgs * len(args))(*args)
    L.call("dmm_gemm_batched", arr, len(args), L.stream())


def _gemm_args(A: Tensor, B: Tensor, out: Optional[Tensor] = None, *, trans_a: bool = False, trans_b: bool = False, bias: Optional[Tensor] = None,
               act: int = L.ACT_NONE, alpha: float = 1.0, residual: Optional[Tensor] = None, e1: float = 1.0, e2: float = 1.0,
               precision: int = 1, split_k: int = 1, accumulate: bool = False, M: Optional[int] = None, N: Optional[int] = None,
               K: Optional[int] = None):
    for nm, t in (("A", A), ("B", B)):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm {nm}: need a 2-D float32 tensor with unit inner stride")
    m = (A.shape[1] if trans_a else A.shape[0]) if M is None else M
    k = (A.shape[0] if trans_a else A.shape[1]) if K is None else K
    n = (B.shape[0] if trans_b else B.shape[1]) if N is None else N
    kb = B.shape[1] if trans_b else B.shape[0]
    if K is None and kb != k:
        raise ValueError(f"gemm: inner dimensions differ ({k} vs {kb})")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=A.device)
        if split_k > 1:
            out.zero_()
    if out.dtype != torch.float32 or out.stride(1) != 1 or out.shape[0] < m or out.shape[1] < n:
        raise ValueError("gemm out: bad shape / dtype / stride")
    g = L.GemmArgs()
    g.A, g.B, g.C = L.ptr(A), L.ptr(B), L.ptr(out)
    g.sam, g.sak = (1, A.stride(0)) if trans_a else (A.stride(0), 1)
    g.sbk, g.sbn = (1, B.stride(0)) if trans_b else (B.stride(0), 1)
    g.ldc = out.stride(0)
    g.bias = L.ptr(_f32c(bias, "bias")) if bias is not None else None
    if bias is not None and bias.numel() < n:
        raise ValueError("gemm bias too short")
    if residual is not None:
        if residual.dtype != torch.float32 or residual.stride(1) != 1 or residual.shape[0] < m or residual.shape[1] < n:
            raise ValueError("gemm residual: bad shape / dtype / stride")
        g.R, g.ldr = L.ptr(residual), residual.stride(0)
    g.M, g.N, g.K = m, n, k
    g.alpha, g.e1, g.e2, g.act = alpha, e1, e2, act
    g.split_k, g.precision, g.accumulate = split_k, precision, int(accumulate)
    g._keep = (A, B, out, bias, residual)
    return g, out


def _bf16_args(A: Tensor, B: Tensor, out: Optional[Tensor] = None, out_bf16: Optional[Tensor] = None, *, bias: Optional[Tensor] = None,
               act: int = L.ACT_NONE, alpha: float = 1.0, residual: Optional[Tensor] = None, e1: float = 1.0, e2: float = 1.0,
               N: Optional[int] = None, K: Optional[int] = None, residual_bf16: Optional[Tensor] = None,
               bias2: Optional[Tensor] = None) -> "L.GemmBf16Args":
    for nm, t in (("A", A), ("B", B)):
        if t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm_bf16_nt {nm}: need a 2-D bfloat16 tensor with unit inner stride")
    m, k = A.shape[0], (A.shape[1] if K is None else K)
    n = B.shape[0] if N is None else N
    if k > A.shape[1] or k > B.shape[1] or n > B.shape[0]:
        raise ValueError("gemm_bf16_nt: shape mismatch")
    g = L.GemmBf16Args()
    g.A, g.lda, g.B, g.ldb = L.ptr(A), A.stride(0), L.ptr(B), B.stride(0)
    if out is not None:
        if out.dtype != torch.float32 or out.stride(1) != 1 or out.shape[0] < m or out.shape[1] < n:
            raise ValueError("gemm_bf16_nt out: bad shape / dtype / stride")
        g.C, g.ldc = L.ptr(out), out.stride(0)
    if out_bf16 is not None:
        if out_bf16.dtype != torch.bfloat16 or out_bf16.stride(1) != 1 or out_bf16.shape[0] < m or out_bf16.shape[1] < n:
            raise ValueError("gemm_bf16_nt out_bf16: bad shape / dtype / stride")
        g.Cb, g.ldcb = L.ptr(out_bf16), out_bf16.stride(0)
    if bias is not None:
        if bias.numel() < n:
            raise ValueError("gemm_bf16_nt bias too short")
        g.bias = L.ptr(_f32c(bias, "bias"))
    if bias2 is not None:  # out_bf16 = tanh(out + bias2) instead of a copy of out (latent reverse chain)
        if bias2.numel() < n or out is None or out_bf16 is None:
            raise ValueError("gemm_bf16_nt bias2: needs both outputs and N entries")
        g.bias2 = L.ptr(_f32c(bias2, "bias2"))
    if residual is not None:
        if residual.dtype != torch.float32 or residual.stride(1) != 1 or residual.shape[0] < m or residual.shape[1] < n:
            raise ValueError("gemm_bf16_nt residual: bad shape / dtype / stride")
        g.R, g.ldr = L.ptr(residual), residual.stride(0)
    if residual_bf16 is not None:
        if residual is not None or residual_bf16.dtype != torch.bfloat16 or residual_bf16.stride(1) != 1 or residual_bf16.shape[0] < m or residual_bf16.shape[1] < n:
            raise ValueError("gemm_bf16_nt residual_bf16: bad shape / dtype / stride (and exclusive with residual)")
        g.Rb, g.ldrb = L.ptr(residual_bf16), residual_bf16.stride(0)
    g.M, g.N, g.K, g.alpha, g.e1, g.e2, g.act = m, n, k, alpha, e1, e2, act
    return g


def gemm_bf16_nt(A: Tensor, B: Tensor, out: Optional[Tensor] = None, out_bf16: Optional[Tensor] = None, **kw) -> None:
    """out[M,N] (fp32) and/or out_bf16 = post(act(alpha * A B^T + bias)); A [M,K], B [N,K] bf16, row-strided views allowed."""
    g = _bf16_args(A, B, out, out_bf16, **kw)
    L.call("dmm_gemm_bf16_nt", C.byref(g), L.stream())


def gemm_bf16_nt_batched(problems: Sequence[dict]) -> None:
    """Up to 4 same-shape problems (dicts of gemm_bf16_nt keyword arguments incl. A, B) in one launch."""
    arr = (L.GemmBf16Args * len(problems))(*[_bf16_args(**p) for p in problems])
    L.call("dmm_gemm_bf16_nt_batched", arr, len(problems), L.stream())


def latent_chain_groups_per_launch(H: int) -> int:
    """Row-block groups (256 rows of one modality) a launch of the persistent chain holds; 0: it cannot run (or slices by itself)."""
    return int(L.lib().dmm_latent_chain_groups_per_launch(int(H)))


def latent_chain(problems: Sequence[dict], rows: int, T: int, coef: Tensor, s_init: float, flags: Tensor, part: Optional[int] = None) -> None:
    """All T - 1 latent-chain steps of up to 4 modalities in one persistent launch sequence (dmm_latent_chain).  problems: dicts with G (bf16
    [H, H]), g (fp32 [H]), z (fp32 [rows, H]: z_T), h_a / h_b (bf16 [rows, H]; h_a = h_{T-1} on entry), bias_t (fp32 [T, H]).  h_0 ends up in h_a when
    T - 1 is even, else in h_b.  part: only that launch of the sequence (dmm_latent_chain_part; parts in order, on one stream)."""
    H = problems[0]["G"].shape[0]
    arr = (L.LatentChainArgs * len(problems))()
    for a, p in zip(arr, problems):
        G, g, z, ha, hb, bt = p["G"], p["g"], p["z"], p["h_a"], p["h_b"], p["bias_t"]
        if G.dtype != torch.bfloat16 or ha.dtype != torch.bfloat16 or hb.dtype != torch.bfloat16 or z.dtype != torch.float32 or bt.dtype != torch.float32 or \
                any(t.stride(1) != 1 for t in (G, z, ha, hb, bt)) or G.shape != (H, H) or any(t.shape[0] < rows or t.shape[1] < H for t in (z, ha, hb)) or \
                bt.shape[0] < T or bt.shape[1] < H or g.numel() < H:
            raise ValueError("latent_chain: bad operands")
        a.G, a.ldg, a.g, a.z, a.ldz = L.ptr(G), G.stride(0), L.ptr(_f32c(g, "g")), L.ptr(z), z.stride(0)
        a.h_a, a.h_b, a.ldh, a.bias_t, a.ldbt = L.ptr(ha), L.ptr(hb), ha.stride(0), L.ptr(bt), bt.stride(0)
        if hb.stride(0) != ha.stride(0):
            raise ValueError("latent_chain: h_a and h_b must share one leading dimension")
    if coef.dtype != torch.float32 or coef.numel() < 2 * T or flags.dtype != torch.int32 or not flags.is_contiguous():
        raise ValueError("latent_chain: coef must be fp32 [2 T], flags int32")
    if part is None:
        L.call("dmm_latent_chain", arr, len(problems), rows, H, T, L.ptr(coef), float(s_init), L.ptr(flags), flags.numel(), L.stream())
    else:
        L.call("dmm_latent_chain_part", arr, len(problems), rows, H, T, L.ptr(coef), float(s_init), L.ptr(flags), flags.numel(), int(part), L.stream())


def cast_bf16(src: Tensor, dst: Tensor, cols: Optional[int] = None, cols_pad: Optional[int] = None) -> None:
    cols = src.shape[1] if cols is None else cols
    cols_pad = cols if cols_pad is None else cols_pad
    if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or src.stride(1) != 1 or dst.stride(1) != 1 or dst.shape[0] < src.shape[0]:
        raise ValueError("cast_bf16: bad operands")
    L.call("dmm_cast_bf16", L.ptr(src), src.stride(0), L.ptr(dst), dst.stride(0), src.shape[0], cols, cols_pad, L.stream())


def transpose_cast_bf16(src: Tensor, dst: Tensor, rows: Optional[int] = None, cols: Optional[int] = None, rows_pad: Optional[int] = None) -> None:
    """dst[c, r] = bf16(src[r, c]); dst is [cols, ld >= rows_pad] and its columns rows..rows_pad are zeroed (rows_pad defaults to
    round_up(rows, 8); the GEMM that contracts over `rows` wants whole 64-wide K tiles)."""
    rows = src.shape[0] if rows is None else rows
    cols = src.shape[1] if cols is None else cols
    rp = (rows + 7) & ~7 if rows_pad is None else rows_pad
    if rp < rows or rp % 8:
        raise ValueError("transpose_cast_bf16: rows_pad must be a multiple of 8 that covers the rows")
    if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or src.stride(1) != 1 or dst.stride(1) != 1 or dst.shape[0] < cols or dst.stride(0) < rp:
        raise ValueError("transpose_cast_bf16: bad operands")
    L.call("dmm_transpose_cast_bf16", L.ptr(src), src.stride(0), L.ptr(dst), dst.stride(0), rows, cols, rp, L.stream())


def cast_bf16_split(src: Tensor, hi: Tensor, lo: Tensor, cols: Optional[int] = None, cols_pad: Optional[int] = None) -> None:
    """hi = bf16(src), lo = bf16(src - hi): together 16 mantissa bits of src (columns cols..cols_pad zero)."""
    cols = src.shape[1] if cols is None else cols
    cols_pad = cols if cols_pad is None else cols_pad
    if src.dtype != torch.float32 or src.stride(1) != 1 or any(t.dtype != torch.bfloat16 or t.stride(1) != 1 or t.shape[0] < src.shape[0] for t in (hi, lo)) or \
            hi.stride(0) != lo.stride(0):
        raise ValueError("cast_bf16_split: bad operands")
    L.call("dmm_cast_bf16_split", L.ptr(src), src.stride(0), L.ptr(hi), L.ptr(lo), hi.stride(0), src.shape[0], cols, cols_pad, L.stream())


def transpose_cast_bf16_split(src: Tensor, hi: Tensor, lo: Tensor, rows: Optional[int] = None, cols: Optional[int] = None, rows_pad: Optional[int] = None) -> None:
    """hi[c, r] = bf16(src[r, c]), lo[c, r] = bf16(src[r, c] - hi[c, r]); destination columns rows..rows_pad zeroed."""
    rows = src.shape[0] if rows is None else rows
    cols = src.shape[1] if cols is None else cols
    rp = (rows + 7) & ~7 if rows_pad is None else rows_pad
    if rp < rows or rp % 8 or src.dtype != torch.float32 or src.stride(1) != 1 or hi.stride(0) != lo.stride(0) or \
            any(t.dtype != torch.bfloat16 or t.stride(1) != 1 or t.shape[0] < cols or t.stride(0) < rp for t in (hi, lo)):
        raise ValueError("transpose_cast_bf16_split: bad operands")
    L.call("dmm_transpose_cast_bf16_split", L.ptr(src), src.stride(0), L.ptr(hi), L.ptr(lo), hi.stride(0), rows, cols, rp, L.stream())


def gemv_rows(W: Tensor, x: Tensor, out: Tensor, cols: Optional[int] = None) -> None:
    """out[r] = sum_{c < cols} W[r, c] x[c] (fp32)."""
    cols = W.shape[1] if cols is None else cols
    if W.dtype != torch.float32 or W.stride(1) != 1 or cols > W.shape[1] or x.numel() < cols or out.numel() < W.shape[0]:
        raise ValueError("gemv_rows: bad operands")
    L.call("dmm_gemv_rows", L.ptr(W), W.stride(0), L.ptr(_f32c(x, "x")), L.ptr(_f32c(out, "out")), W.shape[0], cols, L.stream())


def temb_bias_table(w_emb: Tensor, b_emb: Tensor, w1: Tensor, I: int, b1: Tensor, table: Tensor) -> None:
    T, H = table.shape
    L.call("dmm_temb_bias_table", L.ptr(w_emb), L.ptr(b_emb), w_emb.shape[0], L.ptr(w1), w1.stride(0), I, L.ptr(b1), H, T, L.ptr(_f32c(table, "table")),
           L.stream())


# ------------------------------------------------------------------------------------------------
# normalisation / mixing
# ------------------------------------------------------------------------------------------------
def rownorm_fwd(x: Tensor, y: Optional[Tensor] = None, norm: Optional[Tensor] = None, y16: Optional[Tensor] = None):
    """y16: also write the bf16 copy of y (a gather operand of dmm_spmm_csr_bf16)."""
    _f32c(x, "x")
    y = torch.empty_like(x) if y is None else _f32c(y, "y")
    norm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if norm is None else norm
    if y16 is not None:
        L.call("dmm_rownorm_fwd_x16", L.ptr(x), L.ptr(y), L.ptr(_bf16_like(y16, x, "y16")), L.ptr(norm), x.shape[0], x.shape[1], L.stream())
    else:
        L.call("dmm_rownorm_fwd", L.ptr(x), L.ptr(y), L.ptr(norm), x.shape[0], x.shape[1], L.stream())
    return y, norm


def rownorm_bwd_batched(problems: Sequence[dict]) -> None:
    """rownorm_bwd for up to 4 tensors of one width in one launch (dicts: dy, y, norm, dx and optionally csum [D], pre-zeroed, which receives
    the column sums of dx)."""
    D = problems[0]["dy"].shape[1]
    arr = (L.RownormBwdProblem * len(problems))()
    for a, p in zip(arr, problems):
        dy, y, nrm, dx = p["dy"], p["y"], p["norm"], p["dx"]
        for t in (dy, y, dx):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1] != D or t.shape[0] != dy.shape[0]:
                raise ValueError("rownorm_bwd_batched: contiguous fp32 [rows, D] tensors")
        a.dy, a.y, a.norm, a.dx, a.rows = L.ptr(dy), L.ptr(y), L.ptr(_f32c(nrm, "norm")), L.ptr(dx), dy.shape[0]
        if p.get("csum") is not None:
            a.csum = L.ptr(_f32c(p["csum"], "csum"))
    L.call("dmm_rownorm_bwd_batched", arr, len(problems), D, L.stream())


def rownorm_bwd(dy: Tensor, y: Tensor, norm: Tensor, dx: Optional[Tensor] = None) -> Tensor:
    dx = torch.empty_like(dy) if dx is None else dx
    L.call("dmm_rownorm_bwd", L.ptr(_f32c(dy, "dy")), L.ptr(_f32c(y, "y")), L.ptr(norm), L.ptr(_f32c(dx, "dx")), dy.shape[0], dy.shape[1],
                                    L.stream())
    return dx


def _ptr_array(ts: Sequence[Tensor]):
    return (C.c_void_p * len(ts))(*[L.ptr(_f32c(t, "tensor list")) for t in ts])


def _ptr_array_any(ts: Sequence[Tensor]):
    return (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])


def _bf16_like(t: Tensor, ref: Tensor, nm: str) -> Tensor:
    if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.numel() != ref.numel():
        raise ValueError(f"{nm}: need a contiguous bfloat16 tensor of {ref.numel()} elements")
    return t


def gcn_mix_fwd(bx: Tensor, z: Sequence[Tensor], w_logits: Tensor, maw: float, mx: Optional[Tensor] = None, mx16: Optional[Tensor] = None) -> Tensor:
    """mx16: also write a bf16 copy of the mix (the gather operand of the next SpMM on graphs planned for bf16 operands)."""
    mx = torch.empty_like(bx) if mx is None else mx
    if mx16 is not None:
        L.call("dmm_gcn_mix_fwd_x16", L.ptr(_f32c(bx, "bx")), _ptr_array(z), L.ptr(w_logits), len(z), maw, L.ptr(mx), L.ptr(_bf16_like(mx16, bx, "mx16")),
               bx.numel(), L.stream())
    else:
        L.call("dmm_gcn_mix_fwd", L.ptr(_f32c(bx, "bx")), _ptr_array(z), L.ptr(w_logits), len(z), maw, L.ptr(mx), bx.numel(), L.stream())
    return mx


def zero_batch_rows(dst: Sequence[Tensor], users: Tensor, pos: Tensor, neg: Tensor, U: int, seg: Optional[Tensor] = None) -> None:
    """Rows users[b], U + pos[b], U + neg[b] of up to 8 contiguous [N, D] tensors (fp32 and / or bf16) cleared in one launch.  seg (device int32 [2]):
    users[b] counts only for b in [seg[0], seg[1]) -- a user-sharded replica's own segment of the minibatch."""
    rb = []
    for t in dst:
        if not t.is_contiguous() or t.dim() != 2 or (t.shape[1] * t.element_size()) % 16:
            raise ValueError("zero_batch_rows: contiguous 2-D tensors whose rows are whole 16-byte pieces")
        rb.append(t.shape[1] * t.element_size())
    for i_ in (users, pos, neg):
        if i_.dtype != torch.int32 or not i_.is_contiguous() or i_.numel() != users.numel():
            raise ValueError("zero_batch_rows: index lists must be contiguous int32 of one length")
    L.call("dmm_zero_batch_rows", _ptr_array_any(dst), (C.c_int32 * len(rb))(*rb), len(dst), L.ptr(users), L.ptr(pos), L.ptr(neg), users.numel(), U,
           L.ptr(_i32c(seg, "seg")) if seg is not None else None, L.stream())


def cast_batch_rows_bf16(src: Optional[Sequence[Tensor]], dst: Sequence[Tensor], users: Tensor, pos: Tensor, neg: Tensor, U: int,
                         seg: Optional[Tensor] = None) -> None:
    """Rows users[b], U + pos[b], U + neg[b] of up to 4 [N, D] tensors: dst[t] row = bf16(src[t] row); src = None writes zero rows.  seg: as in
    zero_batch_rows."""
    D = dst[0].shape[1]
    for t in dst:
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.dim() != 2 or t.shape != dst[0].shape:
            raise ValueError("cast_batch_rows_bf16: dst tensors must be contiguous bfloat16 [N, D] of one shape")
    if src is not None:
        if len(src) != len(dst) or any(_f32c(s_, "src").shape != dst[0].shape for s_ in src):
            raise ValueError("cast_batch_rows_bf16: src / dst shape mismatch")
    for i_ in (users, pos, neg):
        if i_.dtype != torch.int32 or not i_.is_contiguous() or i_.numel() != users.numel():
            raise ValueError("cast_batch_rows_bf16: index lists must be contiguous int32 of one length")
    L.call("dmm_cast_batch_rows_bf16", _ptr_array(src) if src is not None else None, _ptr_array_any(dst), len(dst), D, L.ptr(users), L.ptr(pos), L.ptr(neg),
           users.numel(), U, L.ptr(_i32c(seg, "seg")) if seg is not None else None, L.stream())


def gcn_mix_bwd(dmx: Tensor, bx: Tensor, z: Sequence[Tensor], dz: Sequence[Tensor], dbx_extra: Optional[Tensor], w_logits: Tensor, maw: float,
                dbx: Optional[Tensor] = None, dw: Optional[Tensor] = None, partials: Optional[Tensor] = None, count: Optional[tuple] = None,
                dbx16: Optional[Tensor] = None, dz16: Optional[Sequence[Tensor]] = None, dz_overwrite: bool = False):
    """count = (end0, begin1, end1) in elements: only [0, end0) and [begin1, end1) enter dw (user-sharded replicas); None = everything.
    dbx16 / dz16: also write bf16 copies of dbx and of the updated dz[m] (gather operands of the next SpMM); with them dz_overwrite: dz[m] is
    written, not accumulated into (nothing was scattered into it before this launch: it needs no clearing and is not read)."""
    n = dmx.numel()
    dbx = torch.empty_like(dmx) if dbx is None else dbx
    dw = torch.empty(len(z), dtype=torch.float32, device=dmx.device) if dw is None else dw
    if partials is None:
        partials = torch.empty(int(L.lib().dmm_gcn_mix_partials(n)), dtype=torch.float32, device=dmx.device)
    c = (n, n, n) if count is None else tuple(int(v) for v in count)
    if dbx16 is not None or dz16 is not None:
        if dbx16 is None or dz16 is None or len(dz16) != len(z):
            raise ValueError("gcn_mix_bwd: dbx16 and dz16 go together (one per modality)")
        L.call("dmm_gcn_mix_bwd_x16", L.ptr(_f32c(dmx, "dmx")), L.ptr(_f32c(bx, "bx")), _ptr_array(z), _ptr_array(dz), L.ptr(dbx_extra), L.ptr(dbx),
               L.ptr(w_logits), L.ptr(dw), len(z), maw, L.ptr(partials), n, c[0], c[1], c[2], L.ptr(_bf16_like(dbx16, dmx, "dbx16")),
               _ptr_array_any([_bf16_like(t, dmx, "dz16") for t in dz16]), 1 if dz_overwrite else 0, L.stream())
        return dbx, dw
    if dz_overwrite:
        raise ValueError("gcn_mix_bwd: dz_overwrite needs dbx16 / dz16")
    L.call("dmm_gcn_mix_bwd_counted", L.ptr(_f32c(dmx, "dmx")), L.ptr(_f32c(bx, "bx")), _ptr_array(z), _ptr_array(dz), L.ptr(dbx_extra), L.ptr(dbx),
           L.ptr(w_logits), L.ptr(dw), len(z), maw, L.ptr(partials), n, c[0], c[1], c[2], L.stream())
    return dbx, dw


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
class Slots:
    """(tensor, index) pairs whose gathered rows feed InfoNCE terms; `grad` receives the scatter-add backward."""

    def __init__(self, entries):  # entries: list of (src, idx, grad-or-None)
        self.entries = entries
        self.arr = (L.GatherDesc * len(entries))()
        for i, (src, idx, grad) in enumerate(entries):
            _f32c(src, "slot src"); _i32c(idx, "slot idx")
            self.arr[i].src, self.arr[i].idx, self.arr[i].grad = L.ptr(src), L.ptr(idx), L.ptr(grad)
        self.n = len(entries)


def make_terms(terms):  # list of (a, b, temp, weight)
    arr = (L.InfoNCETerm * len(terms))()
    for i, (a, b, temp, w) in enumerate(terms):
        arr[i].a, arr[i].b, arr[i].temp, arr[i].weight = a, b, temp, w
    return arr


def infonce_pack(n_slots: int, B: int, D: int, device) -> Tensor:
    """bf16 operand pack for the MFMA InfoNCE kernels (filled by gather_normalize)."""
    return torch.zeros(int(L.lib().dmm_infonce_pack_bytes(n_slots, B, D)) // 2, dtype=torch.bfloat16, device=device)


def gather_normalize(slots: Slots, B: int, D: int, zhat: Tensor, norm: Tensor, pack: Optional[Tensor] = None, hi_only: bool = False) -> None:
    """hi_only: the pack's lo / loT blocks are not written (the InfoNCE launches that read this pack run at precision 1)."""
    L.call("dmm_gather_normalize_hi" if (hi_only and pack is not None) else "dmm_gather_normalize", slots.arr, slots.n, B, D, L.ptr(zhat), L.ptr(norm),
           L.ptr(pack), L.stream())


def infonce_fwd(zhat: Tensor, terms, n_terms: int, B: int, D: int, lse: Tensor, row_loss: Tensor, pack: Optional[Tensor] = None,
                n_slots: int = 0, precision: int = 0) -> None:
    L.call("dmm_infonce_fwd", L.ptr(zhat), L.ptr(pack), n_slots, precision, terms, n_terms, B, D, L.ptr(lse), L.ptr(row_loss), L.stream())


def infonce_fwd_bwd(zhat: Tensor, terms, n_terms: int, B: int, D: int, lse: Tensor, row_loss: Tensor, dzhat: Tensor, pack: Optional[Tensor] = None,
                    n_slots: int = 0, precision: int = 0) -> None:
    """Forward and backward of all terms for UNIT-NORM rows (gather_normalize's output) in one call; dzhat zeroed by the caller."""
    L.call("dmm_infonce_fwd_bwd", L.ptr(zhat), L.ptr(pack), n_slots, precision, terms, n_terms, B, D, L.ptr(lse), L.ptr(row_loss), L.ptr(dzhat), L.stream())


def infonce_bwd(zhat: Tensor, lse: Tensor, terms, n_terms: int, B: int, D: int, dzhat: Tensor, pack: Optional[Tensor] = None,
                n_slots: int = 0, precision: int = 0) -> None:
    L.call("dmm_infonce_bwd", L.ptr(zhat), L.ptr(pack), n_slots, precision, L.ptr(lse), terms, n_terms, B, D, L.ptr(dzhat), L.stream())


def scatter_normalize_bwd(slots: Slots, B: int, D: int, zhat: Tensor, norm: Tensor, dzhat: Tensor) -> None:
    L.call("dmm_scatter_normalize_bwd", slots.arr, slots.n, B, D, L.ptr(zhat), L.ptr(norm), L.ptr(dzhat), L.stream())


def bpr_fwd_bwd(fin: Tensor, users: Tensor, pos: Tensor, neg: Tensor, U: int, row_loss: Tensor, dfin: Optional[Tensor]) -> None:
    B, D = users.shape[0], fin.shape[1]
    L.call("dmm_bpr_fwd_bwd", L.ptr(_f32c(fin, "fin")), L.ptr(_i32c(users, "users")), L.ptr(_i32c(pos, "pos")), L.ptr(_i32c(neg, "neg")),
                                    B, D, U, L.ptr(row_loss), L.ptr(dfin), L.stream())


def bpr_fwd_bwd_tables(fin_u: Tensor, fin_i: Tensor, users: Tensor, pos: Tensor, neg: Tensor, row_loss: Tensor, dfin_u: Optional[Tensor],
                       dfin_i: Optional[Tensor]) -> None:
    """BPR with the user rows and the item rows in two tables (users index fin_u, pos / neg index fin_i)."""
    B, D = users.shape[0], fin_u.shape[1]
    L.call("dmm_bpr_fwd_bwd_tables", L.ptr(_f32c(fin_u, "fin_u")), L.ptr(_f32c(fin_i, "fin_i")), L.ptr(_i32c(users, "users")), L.ptr(_i32c(pos, "pos")),
           L.ptr(_i32c(neg, "neg")), B, D, L.ptr(row_loss), L.ptr(dfin_u), L.ptr(dfin_i), L.stream())


def gather_rows(slots: "Slots", B: int, D: int, seg: Optional[Tensor], out: Tensor) -> None:
    """out[s][b] = slots[s].src[idx[b]] for b in [seg[0], seg[1]) (seg: device int32[2]; None = every b)."""
    L.call("dmm_gather_rows", slots.arr, slots.n, B, D, L.ptr(seg), L.ptr(_f32c(out, "out")), L.stream())


def scatter_add_rows(slots: "Slots", B: int, D: int, seg: Optional[Tensor], src: Tensor) -> None:
    """slots[s].grad[idx[b]] += src[s][b] for b in [seg[0], seg[1])."""
    L.call("dmm_scatter_add_rows", slots.arr, slots.n, B, D, L.ptr(seg), L.ptr(_f32c(src, "src")), L.stream())


def segment_sum(x: Tensor, n_seg: int, seg_len: int, scale: float, out: Tensor) -> None:
    L.call("dmm_segment_sum", L.ptr(x), n_seg, seg_len, scale, L.ptr(out), L.stream())


def sumsq(x: Tensor, scale: float, out: Tensor, scratch: Optional[Tensor] = None) -> None:
    n = x.numel()
    if scratch is None:
        scratch = torch.empty(int(L.lib().dmm_sumsq_scratch(n)), dtype=torch.float32, device=x.device)
    L.call("dmm_sumsq", L.ptr(x), n, scale, L.ptr(scratch), L.ptr(out), L.stream())


def loss_reduce(row_loss: Tensor, n_seg: int, seg_len: int, seg_scale: float, seg_out: Tensor, x: Tensor, x_scale: float, x_out: Tensor, scratch: Tensor,
                acc: Optional[Tensor] = None, acc_x: int = 0) -> None:
    """segment_sum(row_loss ...) and sumsq(x ...) in one launch (dmm_loss_reduce).  scratch: fp32 [dmm_sumsq_scratch(x.numel()) + 1], zero when first used;
    acc: fp64 running sums (acc[k] += seg_out[k], acc[acc_x] += x_out[0])."""
    n = x.numel()
    if scratch.dtype != torch.float32 or scratch.numel() < int(L.lib().dmm_sumsq_scratch(n)) + 1 or (acc is not None and (acc.dtype != torch.float64 or acc.numel() <= max(acc_x, n_seg - 1))):
        raise ValueError("loss_reduce: scratch too small / acc must be fp64 covering the outputs")
    L.call("dmm_loss_reduce", L.ptr(row_loss), n_seg, seg_len, seg_scale, L.ptr(seg_out), L.ptr(x), n, x_scale, L.ptr(scratch), L.ptr(x_out),
           L.ptr(acc) if acc is not None else None, acc_x, L.stream())


def zero_regions(ts: Sequence[Tensor]) -> None:
    """Up to 4 contiguous tensors cleared by one launch (dmm_zero_regions)."""
    for t in ts:
        if not t.is_contiguous() or (t.numel() * t.element_size()) % 16 or t.data_ptr() % 16:
            raise ValueError("zero_regions: contiguous, 16-byte aligned tensors of whole 16-byte pieces")
    L.call("dmm_zero_regions", (L._P * len(ts))(*[L.ptr(t) for t in ts]), (C.c_int64 * len(ts))(*[t.numel() * t.element_size() for t in ts]), len(ts), L.stream())


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, step: int, extra: Sequence[Tensor] = (), extra_n: Sequence[int] = (),
              l2_coef: float = 0.0, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, l2_n: Optional[int] = None,
              p16: Optional[Tensor] = None, p16_lo: int = 0) -> None:
    """l2_n: the l2 term (l2_coef * p added to the gradient) covers the first l2_n elements only (default: all).  p16 (contiguous bf16): receives the
    bf16 copy of the updated parameters p[p16_lo : p16_lo + p16.numel()]."""
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n):
        raise ValueError("adam: size mismatch")
    ex = (C.c_void_p * max(1, len(extra)))(*[L.ptr(t) for t in extra]) if extra else (C.c_void_p * 1)()
    en = (C.c_int64 * max(1, len(extra)))(*extra_n) if extra else (C.c_int64 * 1)()
    if p16 is not None:  # (argument positions of lr and step as in dmm_adam_step: the recorded plans patch them in place)
        if p16.dtype != torch.bfloat16 or not p16.is_contiguous() or p16_lo < 0 or p16_lo + p16.numel() > n:
            raise ValueError("adam: p16 must be a contiguous bfloat16 tensor covering a window of p")
        L.call("dmm_adam_step_x16", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, ex, en, len(extra), l2_coef, lr, beta1, beta2, eps, step, L.stream(),
               int(n if l2_n is None else l2_n), L.ptr(p16), p16.numel(), int(p16_lo))
    elif l2_n is None:
        L.call("dmm_adam_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, ex, en, len(extra), l2_coef, lr, beta1, beta2, eps, step, L.stream())
    else:  # (the argument positions of lr and step are those of dmm_adam_step: the recorded plans patch them in place)
        L.call("dmm_adam_step_prefix", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, ex, en, len(extra), l2_coef, lr, beta1, beta2, eps, step, L.stream(), int(l2_n))


# ------------------------------------------------------------------------------------------------
# diffusion path
# ------------------------------------------------------------------------------------------------
def qsample_temb(x0: Optional[Tensor], noise: Optional[Tensor], t: Tensor, sqrt_ac: Optional[Tensor], sqrt_1mac: Optional[Tensor],
                 w_emb: Tensor, b_emb: Tensor, xcat: Tensor, I: int, draw_noise: bool = False, seed: int = 0, offset: int = 0,
                 eps_out: Optional[Tensor] = None) -> None:
    B, te = t.shape[0], w_emb.shape[0]
    L.call("dmm_qsample_temb", L.ptr(x0), x0.stride(0) if x0 is not None else 0, L.ptr(noise), L.ptr(_i32c(t, "t")), L.ptr(sqrt_ac),
                                     L.ptr(sqrt_1mac), L.ptr(w_emb), L.ptr(b_emb), te, L.ptr(xcat), xcat.stride(0), B, I, int(draw_noise), seed,
                                     offset, L.ptr(eps_out), L.stream())


def qsample_temb_bits(x0bits: Tensor, noise: Optional[Tensor], t: Tensor, sqrt_ac: Tensor, sqrt_1mac: Tensor, w_emb: Tensor, b_emb: Tensor, xcat: Tensor,
                      I: int, draw_noise: bool = False, seed: int = 0, offset: int = 0, eps_out: Optional[Tensor] = None) -> None:
    """qsample_temb with x_0 given as bit masks (csr_row_bits): int32 [B, >= ceil(I / 32)]."""
    B, te = t.shape[0], w_emb.shape[0]
    if x0bits.dtype != torch.int32 or x0bits.stride(1) != 1 or x0bits.shape[0] < B or x0bits.shape[1] * 32 < I:
        raise ValueError("qsample_temb_bits: x0bits must be int32 [B, >= ceil(I / 32)]")
    L.call("dmm_qsample_temb_bits", L.ptr(x0bits), x0bits.stride(0), L.ptr(noise), L.ptr(_i32c(t, "t")), L.ptr(sqrt_ac), L.ptr(sqrt_1mac), L.ptr(w_emb),
           L.ptr(b_emb), te, L.ptr(xcat), xcat.stride(0), B, I, int(draw_noise), seed, offset, L.ptr(eps_out), L.stream())


def csr_row_bits(ptr: Tensor, col: Tensor, rows: Tensor, I: int, bits: Tensor, col_lo: int = 0) -> None:
    """bits[b, w] bit k = 1 iff item col_lo + 32 w + k is an interaction of user rows[b]; bits int32 [B, ld >= ceil(I / 32)], whole rows written."""
    p, c, r, B = _csr_args(ptr, col, rows)
    if bits.dtype != torch.int32 or bits.stride(1) != 1 or bits.shape[0] < B or bits.shape[1] * 32 < I:
        raise ValueError("csr_row_bits: bits must be int32 [B, >= ceil(I / 32)]")
    L.call("dmm_csr_row_bits_slice", p, c, r, B, int(col_lo), I, L.ptr(bits), bits.stride(0), L.stream())


def diff_loss_bits(out: Tensor, x0bits: Tensor, w: Tensor, um: Tensor, ui: Tensor, I: int, sim_weight: float, reg_term: float, loss: Tensor) -> None:
    """diff_loss forward (loss only) with x_0 given as bit masks."""
    B, d = um.shape
    if w.dtype != torch.float64 or loss.dtype != torch.float64 or x0bits.dtype != torch.int32 or x0bits.shape[1] * 32 < I:
        raise ValueError("diff_loss_bits: w / loss float64, x0bits int32 [B, >= ceil(I / 32)]")
    L.call("dmm_diff_loss_bits", L.ptr(out), out.stride(0), L.ptr(x0bits), x0bits.stride(0), L.ptr(w), L.ptr(um), L.ptr(ui), B, I, d, sim_weight, reg_term,
           L.ptr(loss), L.stream())


def _csr_args(ptr: Tensor, col: Tensor, rows: Tensor):
    if ptr.dtype != torch.int64 or col.dtype != torch.int32 or rows.dtype != torch.int64 or not (ptr.is_contiguous() and col.is_contiguous() and rows.is_contiguous()):
        raise ValueError("CSR rows: ptr int64, col int32, rows int64, all contiguous")
    return L.ptr(ptr), L.ptr(col), L.ptr(rows), int(rows.shape[0])


def csr_dense_rows(ptr: Tensor, col: Tensor, rows: Tensor, I: int, out: Tensor, col_lo: int = 0) -> Tensor:
    """out[b, :] = 0/1 interaction row of user rows[b] over the items col_lo .. col_lo + I - 1 (whole leading dimension written: columns >= I are zero)."""
    p, c, r, B = _csr_args(ptr, col, rows)
    if out.dtype != torch.float32 or out.stride(1) != 1 or out.shape[0] < B or out.shape[1] < I or out.stride(0) % 4 or out.shape[1] != out.stride(0):
        raise ValueError("csr_dense_rows: out must be fp32 [>= B, ld] with ld % 4 == 0, written over its whole width")
    if B:
        L.call("dmm_csr_dense_rows_slice", p, c, r, B, int(col_lo), I, L.ptr(out), out.stride(0), L.stream())
    return out


def csr_gather_sum_bf16(ptr: Tensor, col: Tensor, rows: Tensor, Wt: Tensor, bias: Tensor, out16: Tensor, act: int = L.ACT_NONE) -> None:
    """out16[b] = act(bias + sum of the rows Wt[j] over user rows[b]'s items); Wt bf16 [>= I, H]."""
    p, c, r, B = _csr_args(ptr, col, rows)
    H = bias.numel()
    if Wt.dtype != torch.bfloat16 or out16.dtype != torch.bfloat16 or Wt.stride(1) != 1 or out16.stride(1) != 1 or Wt.shape[1] < H or out16.shape[1] < H or out16.shape[0] < B:
        raise ValueError("csr_gather_sum_bf16: bad operands")
    L.call("dmm_csr_gather_sum_bf16", p, c, r, B, L.ptr(Wt), Wt.stride(0), L.ptr(_f32c(bias, "bias")), H, act, L.ptr(out16), out16.stride(0), L.stream())


def csr_gather_sum_latent(ptr: Tensor, col: Tensor, rows: Tensor, Wt: Tensor, bias: Tensor, z: Tensor, out16: Tensor) -> None:
    """z[b] = sum of the rows Wt[j] over user rows[b]'s items (fp32); out16[b] = bf16(tanh(z[b] + bias)).  Start of the latent chain."""
    p, c, r, B = _csr_args(ptr, col, rows)
    H = bias.numel()
    if Wt.dtype != torch.bfloat16 or out16.dtype != torch.bfloat16 or Wt.stride(1) != 1 or out16.stride(1) != 1 or Wt.shape[1] < H or out16.shape[1] < H or \
            out16.shape[0] < B or z.dtype != torch.float32 or z.stride(1) != 1 or z.shape[0] < B or z.shape[1] < H:
        raise ValueError("csr_gather_sum_latent: bad operands")
    L.call("dmm_csr_gather_sum_latent", p, c, r, B, L.ptr(Wt), Wt.stride(0), L.ptr(_f32c(bias, "bias")), H, L.ptr(z), z.stride(0), L.ptr(out16), out16.stride(0),
           L.stream())


def csr_gather_sum_latent_batched(ptr: Tensor, col: Tensor, rows: Tensor, problems: Sequence[dict]) -> None:
    """csr_gather_sum_latent for up to 4 operand tables (dicts Wt, bias, z, out16) over the same rows in one launch."""
    p, c, r, B = _csr_args(ptr, col, rows)
    H = problems[0]["bias"].numel()
    arr = (L.CsrGatherProblem * len(problems))()
    for a, q in zip(arr, problems):
        Wt, bias, z, out16 = q["Wt"], q["bias"], q["z"], q["out16"]
        if Wt.dtype != torch.bfloat16 or out16.dtype != torch.bfloat16 or Wt.stride(1) != 1 or out16.stride(1) != 1 or Wt.shape[1] < H or out16.shape[1] < H or \
                out16.shape[0] < B or z.dtype != torch.float32 or z.stride(1) != 1 or z.shape[0] < B or z.shape[1] < H or bias.numel() != H:
            raise ValueError("csr_gather_sum_latent_batched: bad operands")
        a.Wt, a.ldw, a.bias, a.z, a.ldz, a.out16, a.ldo = L.ptr(Wt), Wt.stride(0), L.ptr(_f32c(bias, "bias")), L.ptr(z), z.stride(0), L.ptr(out16), out16.stride(0)
    L.call("dmm_csr_gather_sum_latent_batched", p, c, r, B, arr, len(problems), H, L.stream())


def csr_add_rows(ptr: Tensor, col: Tensor, rows: Tensor, I: int, c_: float, y: Optional[Tensor], y16: Optional[Tensor] = None) -> None:
    """y[b, j] += c_ for j in user rows[b]'s items (y16, optional, receives the bf16 of the new values; y None: y16 alone holds the row)."""
    p, c, r, B = _csr_args(ptr, col, rows)
    if (y is None and y16 is None) or (y is not None and (y.dtype != torch.float32 or y.stride(1) != 1 or y.shape[0] < B or y.shape[1] < I)) or \
            (y16 is not None and (y16.dtype != torch.bfloat16 or y16.stride(1) != 1 or y16.shape[1] < I or y16.shape[0] < B)):
        raise ValueError("csr_add_rows: bad operands")
    L.call("dmm_csr_add_rows", p, c, r, B, I, c_, L.ptr(y), y.stride(0) if y is not None else 0, L.ptr(y16), y16.stride(0) if y16 is not None else 0, L.stream())


def rows_axpby(x: Tensor, y: Tensor, t: Tensor, a_table: Tensor, c_table: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[b] = a_table[t_b] * x[b] + c_table[t_b] * y[b]  (fp32 tables on the device; x, y, out row-strided [B, I] views)."""
    B, I = x.shape
    out = torch.empty((B, I), dtype=torch.float32, device=x.device) if out is None else out
    for nm, v in (("x", x), ("y", y), ("out", out)):
        if v.dtype != torch.float32 or v.stride(1) != 1 or tuple(v.shape) != (B, I):
            raise ValueError(f"rows_axpby {nm}: need fp32 [{B}, {I}] with unit inner stride")
    L.call("dmm_rows_axpby", L.ptr(x), x.stride(0), L.ptr(y), y.stride(0), L.ptr(_i32c(t, "t")), L.ptr(_f32c(a_table, "a_table")),
           L.ptr(_f32c(c_table, "c_table")), L.ptr(out), out.stride(0), B, I, L.stream())
    return out


def temb_bwd(t: Tensor, dxcat: Tensor, I: int, te: int, dw: Tensor, db: Tensor) -> None:
    L.call("dmm_temb_bwd", L.ptr(t), L.ptr(dxcat), dxcat.stride(0), I, te, t.shape[0], L.ptr(dw), L.ptr(db), L.stream())


def mul(a: Tensor, b: Tensor, out: Tensor) -> None:
    L.call("dmm_mul_sigmoid_fwd", L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.stream())


def gate_bwd(dout: Tensor, proj: Tensor, gate: Tensor, dproj: Tensor, dpre: Tensor) -> None:
    L.call("dmm_gate_bwd", L.ptr(dout), L.ptr(proj), L.ptr(gate), L.ptr(dproj), L.ptr(dpre), dout.numel(), L.stream())


def zero_(t: Tensor) -> None:
    """hipMemsetAsync on the current stream (recordable, unlike Tensor.zero_())."""
    if not t.is_contiguous():
        raise ValueError("zero_: contiguous tensors only")
    L.call("dmm_fill_zero", L.ptr(t), t.numel() * t.element_size(), L.stream())


def bias_tanh_rows(x: Tensor, bias: Tensor, out16: Optional[Tensor] = None) -> None:
    """x = tanh(x + bias) in place on a contiguous fp32 [rows, cols]; out16: its bf16 copy (contiguous, same shape)."""
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 2 or bias.numel() != x.shape[1] or x.shape[1] % 4:
        raise ValueError("bias_tanh_rows: x contiguous fp32 [rows, cols % 4 == 0], bias [cols]")
    if out16 is not None and (out16.dtype != torch.bfloat16 or not out16.is_contiguous() or out16.shape != x.shape):
        raise ValueError("bias_tanh_rows: out16 must be contiguous bf16 of x's shape")
    if x.shape[0]:
        L.call("dmm_bias_tanh_rows", L.ptr(x), L.ptr(_f32c(bias, "bias")), x.shape[0], x.shape[1], L.ptr(out16) if out16 is not None else None, L.stream())


def tanh_bwd(dh: Tensor, h: Tensor, dpre: Tensor) -> None:
    L.call("dmm_tanh_bwd", L.ptr(dh), L.ptr(h), L.ptr(dpre), dh.numel(), L.stream())


def tanh_bwd_fused(problems: Sequence[dict], rows: int, cols: int, rows_pad: int) -> None:
    """dpre = dh * (1 - h^2) of up to 4 problems in one pass: dicts with dh, h (fp32 [rows, cols] row-strided), d16 (bf16 [rows, cols]), dT16 (bf16
    [cols, >= rows_pad], columns rows..rows_pad zeroed) and csum (fp32 [cols], accumulated into)."""
    arr = (L.TanhBwdFusedArgs * len(problems))()
    for a, p in zip(arr, problems):
        dh, h, d16, dT, cs = p["dh"], p["h"], p["d16"], p["dT16"], p["csum"]
        if dh.dtype != torch.float32 or h.dtype != torch.float32 or d16.dtype != torch.bfloat16 or dT.dtype != torch.bfloat16 or cs.dtype != torch.float32 or \
                any(t.stride(1) != 1 for t in (dh, h, d16, dT)) or any(t.shape[0] < rows or t.shape[1] < cols for t in (dh, h, d16)) or dT.shape[0] < cols or \
                dT.stride(0) < rows_pad or cs.numel() < cols or not cs.is_contiguous() or rows_pad < rows or rows_pad % 8:
            raise ValueError("tanh_bwd_fused: bad operands")
        a.dh, a.ld_dh, a.h, a.ld_h, a.d16, a.ld16, a.dT16, a.ldT, a.csum = L.ptr(dh), dh.stride(0), L.ptr(h), h.stride(0), L.ptr(d16), d16.stride(0), L.ptr(dT), dT.stride(0), L.ptr(cs)
    L.call("dmm_tanh_bwd_fused", arr, len(problems), rows, cols, rows_pad, L.stream())


def add_inplace(y: Tensor, x: Tensor, alpha: float = 1.0) -> None:
    L.call("dmm_add_inplace", L.ptr(y), L.ptr(x), alpha, y.numel(), L.stream())


def colsum(x: Tensor, rows: int, cols: int, out: Tensor, add: bool = False) -> None:
    """out[c] = sum_r x[r, c]; add=True: accumulated into out (which the caller has zeroed: no memset launch)."""
    L.call("dmm_colsum_add" if add else "dmm_colsum", L.ptr(x), x.stride(0), rows, cols, L.ptr(out), L.stream())


def _gs_dev(t: Optional[Tensor]):
    if t is not None and (t.dtype != torch.float64 or t.numel() != 1):
        raise ValueError("gscale_dev: a one-element float64 device tensor")
    return L.ptr(t)


def diff_loss(out: Tensor, x0: Tensor, w: Tensor, um: Tensor, ui: Tensor, I: int, sim_weight: float, reg_term: float, gscale: float,
              loss: Optional[Tensor], dout: Optional[Tensor], dum: Optional[Tensor], gscale_dev: Optional[Tensor] = None) -> None:
    """loss None and dout None: only dum (the gradient through the similarity term) is computed; the [B, I] rows are not read.
    gscale_dev: a float64 device scalar multiplied into gscale (no host read of the minibatch's loss total)."""
    B, d = um.shape
    if out.stride(0) != x0.stride(0) or (dout is not None and dout.stride(0) != out.stride(0)):
        raise ValueError("diff_loss: out, x0 and dout must share one leading dimension")
    if w.dtype != torch.float64 or (loss is not None and loss.dtype != torch.float64):
        raise ValueError("diff_loss: w and loss are float64")
    L.call("dmm_diff_loss", L.ptr(out), L.ptr(x0), out.stride(0), L.ptr(w), L.ptr(um), L.ptr(ui), B, I, d, sim_weight, reg_term, gscale,
                                  L.ptr(loss), L.ptr(dout), L.ptr(dum), _gs_dev(gscale_dev), L.stream())


def topk_rows(x: Tensor, row_ids: Tensor, out_ptr: Tensor, out_idx: Tensor, cols: int) -> None:
    L.call("dmm_topk_rows", L.ptr(x), x.stride(0), L.ptr(_i32c(row_ids, "row_ids")), L.ptr(_i32c(out_ptr, "out_ptr")), L.ptr(out_idx),
                                  row_ids.shape[0], cols, L.stream())


def eval_topk(score: Tensor, users: Tensor, trn_ptr: Tensor, trn_col: Tensor, k: int, cols: Optional[int] = None) -> Tensor:
    """Mask the users' train items in `score` (in place) and return the k best columns per row, best first (Main.py:403-411)."""
    if score.dtype != torch.float32 or score.stride(1) != 1:
        raise ValueError("eval_topk: score must be fp32 with unit column stride")
    if trn_ptr.dtype != torch.int64 or trn_col.dtype != torch.int64:
        raise ValueError("eval_topk: trn_ptr / trn_col are int64")
    cols = score.shape[1] if cols is None else cols
    out = torch.empty(score.shape[0], k, dtype=torch.int32, device=score.device)
    L.call("dmm_eval_topk", L.ptr(score), score.stride(0), L.ptr(_i32c(users, "users")), L.ptr(trn_ptr), L.ptr(trn_col), score.shape[0], cols, k,
           L.ptr(out), L.stream())
    return out


def eval_metrics(top: Tensor, users: Tensor, tst_ptr: Tensor, tst_col: Tensor) -> Tensor:
    """Per-row [recall, ndcg, precision] (fp64) of the recommendations `top` [rows, k] (Main.py:422-448)."""
    if top.dtype != torch.int32 or not top.is_contiguous() or tst_ptr.dtype != torch.int64 or tst_col.dtype != torch.int64:
        raise ValueError("eval_metrics: top is contiguous int32, tst_ptr / tst_col int64")
    out = torch.empty(top.shape[0], 3, dtype=torch.float64, device=top.device)
    L.call("dmm_eval_metrics", L.ptr(top), L.ptr(_i32c(users, "users")), L.ptr(tst_ptr), L.ptr(tst_col), top.shape[0], top.shape[1], L.ptr(out), L.stream())
    return out


def item_project(X: Tensor, F: Tensor, out: Tensor, I: int) -> None:
    """out[B, D] += X[B, :I] @ F[I, D] (bf16-rounded operands).  `out` must be zeroed by the caller (atomic accumulation)."""
    B, D = X.shape[0], F.shape[1]
    for t in (X, F, out):
        if t.dtype != torch.float32 or t.stride(1) != 1:
            raise ValueError("item_project: fp32 tensors with unit inner stride")
    if F.shape[0] < I or out.shape[0] < B or out.shape[1] < D or X.shape[1] < ((I + 31) & ~31):
        raise ValueError("item_project: shape mismatch (X needs round_up(I, 32) readable columns)")
    L.call("dmm_item_project", L.ptr(X), X.stride(0), L.ptr(F), F.stride(0), L.ptr(out), out.stride(0), B, I, D, L.stream())


def item_project_batched_x16(problems: Sequence[dict], D: int) -> None:
    """out[c, r] += sum_k X[r, k] F[k, c] with X in bf16 [rows, >= round_up(K, 32)] (the out_transposed form of item_project_batched)."""
    arr = (L.ProjectProblem * len(problems))()
    for a, p in zip(arr, problems):
        X, F, out, K = p["X"], p["F"], p["out"], int(p["K"])
        rows = int(p.get("rows", X.shape[0]))
        if X.dtype != torch.bfloat16 or F.dtype != torch.float32 or out.dtype != torch.float32 or any(t.dim() != 2 or t.stride(1) != 1 for t in (X, F, out)) or \
                F.shape[0] < K or F.shape[1] < D or out.shape[0] < D or out.shape[1] < rows or X.shape[0] < rows or X.shape[1] < ((K + 31) & ~31):
            raise ValueError("item_project_batched_x16: bad operands")
        a.X, a.ldx, a.F, a.ldf, a.out, a.ldo, a.rows, a.K = L.ptr(X), X.stride(0), L.ptr(F), F.stride(0), L.ptr(out), out.stride(0), rows, K
    L.call("dmm_item_project_batched_x16", arr, len(problems), D, L.stream())


def item_project_batched(problems: Sequence[dict], D: int, f_transposed: bool = False, out_transposed: bool = False) -> None:
    """Up to 4 problems  out[r, c] += sum_k X[r, k] Fm(k, c) (+ bias[c])  in one launch (dicts: X [rows, >= round_up(K, 32)], F, out, optional bias,
    K).  f_transposed: F is [D, K] (an nn.Linear weight); out_transposed: the result goes to out[c, r].  `out` holds the value to add to."""
    arr = (L.ProjectProblem * len(problems))()
    xb = all(p["X"].dtype == torch.bfloat16 for p in problems)  # a constant X cast once (dmm_item_project_batched_xb: same result, half the bytes)
    for a, p in zip(arr, problems):
        X, F, out, K = p["X"], p["F"], p["out"], int(p["K"])
        rows = int(p.get("rows", X.shape[0]))
        for t in (X, F, out):
            if t.dtype != (torch.bfloat16 if (xb and t is X) else torch.float32) or t.dim() != 2 or t.stride(1) != 1:
                raise ValueError("item_project_batched: 2-D fp32 tensors with unit inner stride (X: all fp32 or all bf16)")
        fk, fd = (F.shape[1], F.shape[0]) if f_transposed else (F.shape[0], F.shape[1])
        orr, oc = (out.shape[1], out.shape[0]) if out_transposed else (out.shape[0], out.shape[1])
        if fk < K or fd < D or orr < rows or oc < D or X.shape[0] < rows or X.shape[1] < ((K + 31) & ~31):
            raise ValueError("item_project_batched: shape mismatch (X needs round_up(K, 32) readable columns)")
        a.X, a.ldx, a.F, a.ldf, a.out, a.ldo, a.rows, a.K = L.ptr(X), X.stride(0), L.ptr(F), F.stride(0), L.ptr(out), out.stride(0), rows, K
        if p.get("bias") is not None:
            a.bias = L.ptr(_f32c(p["bias"], "bias"))
    if xb:
        L.call("dmm_item_project_batched_xb", arr, len(problems), D, int(f_transposed), int(out_transposed), L.stream())
    else:
        L.call("dmm_item_project_batched", arr, len(problems), D, int(f_transposed), int(out_transposed), L.stream())


def item_expand(P: Tensor, F: Tensor, R: Tensor, Y: Tensor, I: int) -> None:
    """Y[B, :I] = R[B, :I] + P[B, D] @ F[I, D]^T (bf16-rounded operands); Y may be R."""
    B, D = P.shape[0], P.shape[1]
    for t in (P, F, R, Y):
        if t.dtype != torch.float32 or t.stride(1) != 1:
            raise ValueError("item_expand: fp32 tensors with unit inner stride")
    if F.shape[0] < I or F.shape[1] != D or R.shape[0] < B or Y.shape[0] < B or R.shape[1] < I or Y.shape[1] < I:
        raise ValueError("item_expand: shape mismatch")
    L.call("dmm_item_expand", L.ptr(P), P.stride(0), L.ptr(F), F.stride(0), L.ptr(R), R.stride(0), L.ptr(Y), Y.stride(0), B, I, D, L.stream())


def item_expand_fused(P: Tensor, F: Tensor, R: Tensor, Y16: Tensor, YT: Tensor, I: int, Kc: int, Kt: int, x0: Optional[Tensor] = None,
                      w: Optional[Tensor] = None, gscale: float = 0.0, csum: Optional[Tensor] = None, x0bits: Optional[Tensor] = None,
                      gscale_dev: Optional[Tensor] = None, F16: Optional[Tensor] = None) -> None:
    """item_expand whose result goes straight into the bf16 operands of the next GEMMs (Y16 [B, Kc] row-major, YT [Kt, round_up(B, 64)]
    transposed).  x0 / w / csum given: the loss-gradient form v = gscale w[b] 2/I (R - x0) + P F^T with its column sums added to csum.
    F16: a bf16 copy of F (cast_bf16) for the kernel to stage instead of F."""
    B, D = P.shape[0], P.shape[1]
    mode = 0 if (x0 is None and x0bits is None) else 1
    for t in (P, F, R) + ((x0,) if x0 is not None else ()):
        if t.dtype != torch.float32 or t.stride(1) != 1:
            raise ValueError("item_expand_fused: fp32 tensors with unit inner stride")
    Bp = (B + 63) & ~63
    if Y16.dtype != torch.bfloat16 or YT.dtype != torch.bfloat16 or Y16.stride(1) != 1 or YT.stride(1) != 1 or Y16.shape[0] < B or Y16.shape[1] < Kc or \
            YT.shape[0] < Kt or YT.shape[1] < Bp or F.shape[0] < I or F.shape[1] != D or R.shape[0] < B or R.shape[1] < (I if mode else Kc):
        raise ValueError("item_expand_fused: shape mismatch")
    a = L.ExpandFusedArgs()
    a.P, a.ldp, a.F, a.ldf, a.R, a.ldr = L.ptr(P), P.stride(0), L.ptr(F), F.stride(0), L.ptr(R), R.stride(0)
    a.Y16, a.ldy, a.YT, a.ldt, a.B, a.I, a.Kc, a.Kt = L.ptr(Y16), Y16.stride(0), L.ptr(YT), YT.stride(0), B, I, Kc, Kt
    if F16 is not None:
        if F16.dtype != torch.bfloat16 or F16.stride(1) != 1 or F16.shape[0] < I or F16.shape[1] != D:
            raise ValueError("item_expand_fused: F16 must be the bfloat16 copy of F")
        a.F16, a.ldf16 = L.ptr(F16), F16.stride(0)
    if mode:
        if w.dtype != torch.float64 or csum is None or csum.numel() < I:
            raise ValueError("item_expand_fused: w is float64 [B], csum fp32 [I]")
        a.w, a.gscale, a.csum = L.ptr(w), float(gscale), L.ptr(_f32c(csum, "csum"))
        a.gscale_dev = _gs_dev(gscale_dev)
        if x0 is not None:
            a.X0, a.ldx0 = L.ptr(x0), x0.stride(0)
        else:  # x_0 as bit masks (csr_row_bits)
            if x0bits.dtype != torch.int32 or x0bits.stride(1) != 1 or x0bits.shape[0] < B or x0bits.shape[1] * 32 < I:
                raise ValueError("item_expand_fused: x0bits must be int32 [B, >= ceil(I / 32)]")
            a.X0bits, a.ldbits = L.ptr(x0bits), x0bits.stride(0)
    L.call("dmm_item_expand_fused", C.byref(a), mode, D, L.stream())


def noised_start(x0: Tensor, randn: Optional[Tensor], sqrt_ac: float, sqrt_1mac: float, xT: Tensor, I: int, seed: int = 0, offset: int = 0) -> None:
    """xT[:, :I] = sqrt_ac * x0 + sqrt_1mac * sign(x0) * normalize(randn) (Model.py:336-341); randn None = in-kernel Philox draw."""
    if x0.dtype != torch.float32 or xT.dtype != torch.float32 or x0.stride(1) != 1 or xT.stride(1) != 1 or x0.shape[1] < I or xT.shape[1] < I:
        raise ValueError("noised_start: fp32 operands with at least I columns")
    if randn is not None and (randn.dtype != torch.float32 or not randn.is_contiguous() or tuple(randn.shape) != (x0.shape[0], I)):
        raise ValueError("noised_start: randn must be a contiguous fp32 [B, I] tensor")
    L.call("dmm_noised_start", L.ptr(x0), x0.stride(0), L.ptr(randn), sqrt_ac, sqrt_1mac, L.ptr(xT), xT.stride(0), x0.shape[0], I, seed, offset, L.stream())

"""ctypes binding of libdiffmm_hip.so (include/diffmm_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this module raises.
Build the library with `python __graft_entry__.py` (or `make -C diffmm_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMM_LIB (tools only): another build of the same library, e.g. the -DP8_TIMING build tools/latent_stamps.py reads per-workgroup timestamps from
LIB_PATH = os.environ.get("DMM_LIB") or os.path.join(_HERE, "libdiffmm_hip.so")

SPMM_MAX_BATCH = 4
ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2


class SpmmProblem(C.Structure):
    _fields_ = [("tasks", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("x_lo", C.c_void_p), ("x_hi", C.c_void_p),
                ("y", C.c_void_p), ("c1", C.c_void_p), ("c2", C.c_void_p), ("y_pert", C.c_void_p), ("noise", C.c_void_p),
                ("y_mean", C.c_void_p), ("m1", C.c_void_p), ("m2", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("partial", C.c_void_p), ("fix", C.c_void_p), ("n_partial_blocks", C.c_int32), ("n_fix", C.c_int32), ("n_rows", C.c_int32), ("n_tasks", C.c_int32), ("n_head_tasks", C.c_int32), ("split", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float),
                ("noise_degree", C.c_float), ("raw", C.c_void_p), ("raw_from", C.c_int32), ("rng_row_split", C.c_int32), ("rng_row_add_lo", C.c_int32),
                ("rng_row_add_hi", C.c_int32), ("y16", C.c_void_p), ("y_pert16", C.c_void_p), ("fix_count", C.c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("sam", C.c_int64), ("sak", C.c_int64), ("B", C.c_void_p), ("sbk", C.c_int64), ("sbn", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("alpha", C.c_float), ("e1", C.c_float), ("e2", C.c_float),
                ("act", C.c_int32), ("split_k", C.c_int32), ("precision", C.c_int32), ("accumulate", C.c_int32)]


class GemmBf16Args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("Cb", C.c_void_p), ("ldcb", C.c_int64), ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", C.c_int64), ("Rb", C.c_void_p), ("ldrb", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("alpha", C.c_float), ("e1", C.c_float), ("e2", C.c_float),
                ("act", C.c_int32), ("bias2", C.c_void_p)]


class LatentChainArgs(C.Structure):
    _fields_ = [("G", C.c_void_p), ("ldg", C.c_int64), ("g", C.c_void_p), ("z", C.c_void_p), ("ldz", C.c_int64), ("h_a", C.c_void_p), ("h_b", C.c_void_p),
                ("ldh", C.c_int64), ("bias_t", C.c_void_p), ("ldbt", C.c_int64)]


class CsrGatherProblem(C.Structure):
    _fields_ = [("Wt", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("z", C.c_void_p), ("ldz", C.c_int64), ("out16", C.c_void_p), ("ldo", C.c_int64)]


class TanhBwdFusedArgs(C.Structure):
    _fields_ = [("dh", C.c_void_p), ("ld_dh", C.c_int64), ("h", C.c_void_p), ("ld_h", C.c_int64), ("d16", C.c_void_p), ("ld16", C.c_int64),
                ("dT16", C.c_void_p), ("ldT", C.c_int64), ("csum", C.c_void_p)]


class ProjectProblem(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("F", C.c_void_p), ("ldf", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64), ("bias", C.c_void_p),
                ("rows", C.c_int32), ("K", C.c_int32)]


class ExpandFusedArgs(C.Structure):
    _fields_ = [("P", C.c_void_p), ("ldp", C.c_int64), ("F", C.c_void_p), ("ldf", C.c_int64), ("R", C.c_void_p), ("ldr", C.c_int64), ("X0", C.c_void_p),
                ("ldx0", C.c_int64), ("w", C.c_void_p), ("gscale", C.c_double), ("Y16", C.c_void_p), ("ldy", C.c_int64), ("YT", C.c_void_p), ("ldt", C.c_int64),
                ("csum", C.c_void_p), ("B", C.c_int32), ("I", C.c_int32), ("Kc", C.c_int32), ("Kt", C.c_int32), ("X0bits", C.c_void_p), ("ldbits", C.c_int64),
                ("gscale_dev", C.c_void_p), ("F16", C.c_void_p), ("ldf16", C.c_int64)]


class RownormBwdProblem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("y", C.c_void_p), ("norm", C.c_void_p), ("dx", C.c_void_p), ("csum", C.c_void_p), ("rows", C.c_int32)]


class GatherDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("idx", C.c_void_p), ("grad", C.c_void_p)]


class InfoNCETerm(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("temp", C.c_float), ("weight", C.c_float)]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_U64 = C.c_uint64

_SIGS = {
    "dmm_version": (C.c_int, []),
    "dmm_last_error_string": (C.c_char_p, []),
    "dmm_spmm_tasks_per_block": (C.c_int, [_I]),
    "dmm_spmm_csr_f32": (C.c_int, [C.POINTER(SpmmProblem), _I, _I, _P]),
    "dmm_spmm_csr_bf16_hotlds": (C.c_int, [C.POINTER(SpmmProblem), _P, _I, _P, _I, _P]),
    "dmm_spmm_tasks_per_block_bf16": (C.c_int, [_I]),
    "dmm_spmm_csr_bf16": (C.c_int, [C.POINTER(SpmmProblem), _I, _I, _P]),
    "dmm_spmm_epilogue_rows": (C.c_int, [C.POINTER(SpmmProblem), _I, _I, _P]),
    "dmm_gemm": (C.c_int, [C.POINTER(GemmArgs), _P]),
    "dmm_gemm_batched": (C.c_int, [C.POINTER(GemmArgs), _I, _P]),
    "dmm_gemm_bf16_nt": (C.c_int, [C.POINTER(GemmBf16Args), _P]),
    "dmm_gemm_bf16_nt_batched": (C.c_int, [C.POINTER(GemmBf16Args), _I, _P]),
    "dmm_latent_chain": (C.c_int, [C.POINTER(LatentChainArgs), _I, _I, _I, _I, _P, C.c_float, _P, _I, _P]),
    "dmm_latent_chain_part": (C.c_int, [C.POINTER(LatentChainArgs), _I, _I, _I, _I, _P, C.c_float, _P, _I, _I, _P]),
    "dmm_latent_chain_groups_per_launch": (C.c_int, [_I]),
    "dmm_latent_chain_supported": (C.c_int, [_I]),
    "dmm_cast_bf16": (C.c_int, [_P, _L, _P, _L, _I, _I, _I, _P]),
    "dmm_transpose_cast_bf16": (C.c_int, [_P, _L, _P, _L, _I, _I, _I, _P]),
    "dmm_cast_bf16_split": (C.c_int, [_P, _L, _P, _P, _L, _I, _I, _I, _P]),
    "dmm_transpose_cast_bf16_split": (C.c_int, [_P, _L, _P, _P, _L, _I, _I, _I, _P]),
    "dmm_gemv_rows": (C.c_int, [_P, _L, _P, _P, _I, _I, _P]),
    "dmm_temb_bias_table": (C.c_int, [_P, _P, _I, _P, _L, _I, _P, _I, _I, _P, _P]),
    "dmm_rownorm_fwd": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "dmm_rownorm_fwd_x16": (C.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    "dmm_rownorm_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    "dmm_rownorm_bwd_batched": (C.c_int, [C.POINTER(RownormBwdProblem), _I, _I, _P]),
    "dmm_gcn_mix_fwd": (C.c_int, [_P, C.POINTER(_P), _P, _I, _F, _P, _L, _P]),
    "dmm_gcn_mix_partials": (C.c_int64, [_L]),
    "dmm_gcn_mix_bwd": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _I, _F, _P, _L, _P]),
    "dmm_gcn_mix_bwd_counted": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _I, _F, _P, _L, _L, _L, _L, _P]),
    "dmm_gcn_mix_fwd_x16": (C.c_int, [_P, C.POINTER(_P), _P, _I, _F, _P, _P, _L, _P]),
    "dmm_gcn_mix_bwd_x16": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _I, _F, _P, _L, _L, _L, _L, _P, C.POINTER(_P), _I, _P]),
    "dmm_zero_batch_rows": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int32), _I, _P, _P, _P, _I, _I, _P, _P]),
    "dmm_cast_batch_rows_bf16": (C.c_int, [C.POINTER(_P), C.POINTER(_P), _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "dmm_infonce_pack_bytes": (C.c_int64, [_I, _I, _I]),
    "dmm_gather_normalize": (C.c_int, [C.POINTER(GatherDesc), _I, _I, _I, _P, _P, _P, _P]),
    "dmm_gather_normalize_hi": (C.c_int, [C.POINTER(GatherDesc), _I, _I, _I, _P, _P, _P, _P]),
    "dmm_infonce_fwd": (C.c_int, [_P, _P, _I, _I, C.POINTER(InfoNCETerm), _I, _I, _I, _P, _P, _P]),
    "dmm_infonce_fwd_bwd": (C.c_int, [_P, _P, _I, _I, C.POINTER(InfoNCETerm), _I, _I, _I, _P, _P, _P, _P]),
    "dmm_infonce_bwd": (C.c_int, [_P, _P, _I, _I, _P, C.POINTER(InfoNCETerm), _I, _I, _I, _P, _P]),
    "dmm_scatter_normalize_bwd": (C.c_int, [C.POINTER(GatherDesc), _I, _I, _I, _P, _P, _P, _P]),
    "dmm_bpr_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "dmm_bpr_fwd_bwd_tables": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "dmm_gather_rows": (C.c_int, [C.POINTER(GatherDesc), _I, _I, _I, _P, _P, _P]),
    "dmm_scatter_add_rows": (C.c_int, [C.POINTER(GatherDesc), _I, _I, _I, _P, _P, _P]),
    "dmm_segment_sum": (C.c_int, [_P, _I, _L, _F, _P, _P]),
    "dmm_sumsq_scratch": (C.c_int64, [_L]),
    "dmm_sumsq": (C.c_int, [_P, _L, _F, _P, _P, _P]),
    "dmm_loss_reduce": (C.c_int, [_P, _I, _L, _F, _P, _P, _L, _F, _P, _P, _P, _I, _P]),
    "dmm_zero_regions": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int64), _I, _P]),
    "dmm_adam_step": (C.c_int, [_P, _P, _P, _P, _L, C.POINTER(_P), C.POINTER(_L), _I, _F, _F, _F, _F, _F, _I, _P]),
    "dmm_adam_step_prefix": (C.c_int, [_P, _P, _P, _P, _L, C.POINTER(_P), C.POINTER(_L), _I, _F, _F, _F, _F, _F, _I, _P, _L]),
    "dmm_adam_step_x16": (C.c_int, [_P, _P, _P, _P, _L, C.POINTER(_P), C.POINTER(_L), _I, _F, _F, _F, _F, _F, _I, _P, _L, _P, _L, _L]),
    "dmm_qsample_temb": (C.c_int, [_P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _I, _I, _U64, _U64, _P, _P]),
    "dmm_csr_dense_rows": (C.c_int, [_P, _P, _P, _I, _I, _P, _L, _P]),
    "dmm_csr_dense_rows_slice": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "dmm_bias_tanh_rows": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "dmm_csr_gather_sum_bf16": (C.c_int, [_P, _P, _P, _I, _P, _L, _P, _I, _I, _P, _L, _P]),
    "dmm_csr_gather_sum_latent": (C.c_int, [_P, _P, _P, _I, _P, _L, _P, _I, _P, _L, _P, _L, _P]),
    "dmm_csr_gather_sum_latent_batched": (C.c_int, [_P, _P, _P, _I, C.POINTER(CsrGatherProblem), _I, _I, _P]),
    "dmm_csr_add_rows": (C.c_int, [_P, _P, _P, _I, _I, _F, _P, _L, _P, _L, _P]),
    "dmm_rows_axpby": (C.c_int, [_P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _P]),
    "dmm_temb_bwd": (C.c_int, [_P, _P, _L, _I, _I, _I, _P, _P, _P]),
    "dmm_mul_sigmoid_fwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "dmm_gate_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _P]),
    "dmm_tanh_bwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "dmm_neg_sample": (C.c_int, [_P, _P, _P, _I, _I, _U64, _U64, _P, _P]),
    "dmm_tanh_bwd_fused": (C.c_int, [_P, _I, _I, _I, _I, _P]),
    "dmm_add_inplace": (C.c_int, [_P, _P, _F, _L, _P]),
    "dmm_colsum": (C.c_int, [_P, _L, _I, _I, _P, _P]),
    "dmm_colsum_add": (C.c_int, [_P, _L, _I, _I, _P, _P]),
    "dmm_diff_loss": (C.c_int, [_P, _P, _L, _P, _P, _P, _I, _I, _I, _F, _D, _D, _P, _P, _P, _P, _P]),
    "dmm_diff_loss_bits": (C.c_int, [_P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _F, _D, _P, _P]),
    "dmm_csr_row_bits": (C.c_int, [_P, _P, _P, _I, _I, _P, _L, _P]),
    "dmm_csr_row_bits_slice": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "dmm_qsample_temb_bits": (C.c_int, [_P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _I, _I, _U64, _U64, _P, _P]),
    "dmm_topk_rows": (C.c_int, [_P, _L, _P, _P, _P, _I, _I, _P]),
    "dmm_noised_start": (C.c_int, [_P, _L, _P, _F, _F, _P, _L, _I, _I, _U64, _U64, _P]),
    "dmm_item_project": (C.c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P]),
    "dmm_item_project_batched": (C.c_int, [C.POINTER(ProjectProblem), _I, _I, _I, _I, _P]),
    "dmm_item_project_batched_x16": (C.c_int, [C.POINTER(ProjectProblem), _I, _I, _P]),
    "dmm_item_project_batched_xb": (C.c_int, [C.POINTER(ProjectProblem), _I, _I, _I, _I, _P]),
    "dmm_item_expand": (C.c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P]),
    "dmm_item_expand_fused": (C.c_int, [C.POINTER(ExpandFusedArgs), _I, _I, _P]),
    "dmm_graph_build_scratch_bytes": (C.c_int, [_I, _I, _L, _I, _P]),
    "dmm_graph_build_task_capacity": (C.c_int, [_I, _I, _L, _I, _I, _P, _P]),
    "dmm_graph_build": (C.c_int, [_P, _P, _P, _I, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmm_count_items": (C.c_int, [_P, _L, _I, _P, _P]),
    "dmm_graph_build_shard": (C.c_int, [_P, _P, _P, _I, _I, _L, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmm_eval_topk": (C.c_int, [_P, _L, _P, _P, _P, _I, _I, _I, _P, _P]),
    "dmm_eval_metrics": (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "dmm_fill_zero": (C.c_int, [_P, _L, _P]),
}

_lib: Optional[C.CDLL] = None


def exported_symbols():
    return sorted(_SIGS.keys())


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` -- diffmm_amd has no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().dmm_last_error_string().decode(errors="replace")
        raise RuntimeError(f"libdiffmm_hip {what} failed with code {code}: {msg}")


# ---- launch plans -----------------------------------------------------------------------------------
# A training step is a fixed sequence of C calls on fixed HBM buffers.  Recording it once and replaying the prebuilt
# (function, ctypes-arguments) list removes the Python-side argument marshalling from the steady state (the joint step
# is ~45 launches of 5-15 us each, so per-launch host cost is first order).
_recording: Optional[list] = None


def call(name: str, *args) -> None:
    """Launch (or, inside `Recorder`, only record) one C-ABI call."""
    fn = getattr(lib(), name)
    if _recording is not None:
        _recording.append([fn, list(args), name])
        return
    rc = fn(*args)
    if rc:
        check(rc, name)


ADAM_CALLS = ("dmm_adam_step", "dmm_adam_step_prefix", "dmm_adam_step_x16")  # (lr, step) sit at argument positions 9 and 13 in all of them


def is_recording() -> bool:
    """True inside a `Recorder` block (calls are being recorded into a launch plan, not launched)."""
    return _recording is not None


class Recorder:
    """`with Recorder() as plan:` -- calls made through `call()` are appended to `plan` instead of being launched."""

    def __enter__(self):
        global _recording
        if _recording is not None:
            raise RuntimeError("nested launch recording")
        _recording = []
        self.plan = _recording
        return self.plan

    def __exit__(self, *exc):
        global _recording
        _recording = None
        return False


def call_py(fn, *args) -> None:
    """Run (or, inside `Recorder`, record) a host-side step of a launch sequence -- a torch.distributed collective between two
    kernel launches.  Recorded entries are replayed in order with the C calls (fn must return None / 0)."""
    if _recording is not None:
        _recording.append([fn, list(args), "py:" + getattr(fn, "__name__", "fn")])
        return
    fn(*args)


def replay(plan) -> None:
    for fn, args, name in plan:
        rc = fn(*args)
        if rc:
            check(rc, name)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("diffmm_amd kernels take device tensors only (there is no CPU path)")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream

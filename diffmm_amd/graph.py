"""Adjacency construction and SpMM work planning (host side, numpy).

Mirrors DataHandler.normalizeAdj / DataHandler.makeTorchAdj of the reference (DataHandler.py:52-93): the
bipartite block matrix [[0, R], [R^T, 0]], binarised, plus self loops, scaled by D^-1/2 on both sides in
float64 and cast to fp32 once.  Instead of an uncoalesced torch COO tensor the result is CSR (int32 row
pointers / column ids, fp32 values) plus the task list the HIP SpMM kernel consumes.
"""
from __future__ import annotations

import itertools
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
import torch


def normalized_bipartite_csr(rows: np.ndarray, cols: np.ndarray, users: int, items: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(user, item) edge list -> CSR of D^-1/2 (A + I) D^-1/2 over N = users + items nodes.
    Duplicate edges collapse (the reference binarises, DataHandler.py:83).  Columns ascend inside a row."""
    n = users + items
    r = np.asarray(rows, dtype=np.int64)
    c = np.asarray(cols, dtype=np.int64)
    if r.size and (r.min() < 0 or r.max() >= users or c.min() < 0 or c.max() >= items):
        raise ValueError("edge list out of range")
    # de-duplicate user->item edges once, then mirror
    e = np.unique(r * items + c)
    eu, ei = e // items, e % items + users
    diag = np.arange(n, dtype=np.int64)
    src = np.concatenate([eu, ei, diag])
    dst = np.concatenate([ei, eu, diag])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    counts = np.bincount(src, minlength=n)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    scale = np.zeros(n, dtype=np.float64)
    nz = counts > 0
    scale[nz] = counts[nz].astype(np.float64) ** -0.5
    val = ((scale[src] * 1.0) * scale[dst]).astype(np.float32)
    if rowptr[-1] >= 2 ** 31:
        raise ValueError("more than 2^31 non-zeros: shard the graph")
    return rowptr.astype(np.int32), dst.astype(np.int32), val


@dataclass
class Plan:
    """SpMM work list: int32 [T, 4] tasks (row, nz_begin, nz_end, run_pos << 16 | run_len); the first `n_partial_blocks` workgroups are
    segments of giant rows (their partial rows are summed by the fix-up pass described by `fix`: int32 [n_giant, 4] =
    row, first_slot, n_slots, 0); the first `head` tasks (segments + split rows) keep round-robin XCD dispatch."""
    tasks: "np.ndarray"
    head: int
    n_partial_blocks: int
    fix: "np.ndarray"


def hub_keys(rowptr: np.ndarray, col: np.ndarray) -> np.ndarray:
    """Locality key of every row: the id of its highest-degree neighbour (its "hub"; the self loop does not count, rows without neighbours keep
    their own id).  Rows that share a hub -- users around one popular item, items around one heavy user -- gather overlapping sets of X rows;
    plan_tasks(row_key=...) schedules them next to each other so that those rows are still in the XCD's L2."""
    rp = np.asarray(rowptr, dtype=np.int64)
    n = rp.shape[0] - 1
    c = np.asarray(col, dtype=np.int64)
    deg = rp[1:] - rp[:-1]
    nzrow = np.repeat(np.arange(n, dtype=np.int64), deg)
    w = deg[c].copy()
    w[c == nzrow] = -1
    idx = np.lexsort((c, w, nzrow))  # per row: ascending neighbour degree (ties: larger id last)
    last = rp[1:] - 1
    key = np.arange(n, dtype=np.int64)
    has = deg > 0
    key[has] = c[idx[last[has]]]
    lone = has & (w[idx[np.maximum(last, 0)]] < 0)  # only the self loop
    key[lone] = np.flatnonzero(lone)
    return key


BASE_SEG_MULT = 4  # the base graph's in-workgroup row limit (plan_tasks: seg_mult)


def plan_tasks(rowptr: np.ndarray, D: int, task_len: int = 32, return_head: bool = False, full: bool = False, row_key: Optional[np.ndarray] = None,
               seg_mult: int = 2):
    """Work list for dmm_spmm_csr_f32.

    A workgroup of the kernel holds G = 1024 / D task groups.  Rows with at most `task_len` non-zeros are one task.  Longer
    rows (up to 2 G task_len non-zeros) are cut into k <= G tasks that sit in consecutive groups of ONE workgroup (combined
    through LDS in task order): k is rounded up to a power of two and runs are placed largest first, so every run is aligned to
    its own size and never straddles a workgroup.  Still longer ("giant") rows are cut into segments of 2 G task_len non-zeros,
    one full workgroup each; the segment partials are added in order by a second pass.  Padding tasks have row = -1.
    row_key (optional, one integer per row): rows are scheduled in ascending key order (ties: row order) instead of row order -- any order
    is valid, a task names its row; see hub_keys.
    seg_mult: a row stays inside ONE workgroup up to seg_mult G task_len non-zeros (default 2, the device planner's constant: dmm_graph_build
    plans the rebuilt graphs bit-identically).  The BASE graph is planned on the host only and takes 4: at TikTok shape its longest rows (users of
    degree 513-521 incl. the self loop) then need no segments -- and its launches no second (fix-up) launch, 4 of the joint step's 6.
    """
    G = 1024 // D
    if G * D != 1024 or D % 4:
        raise ValueError(f"D={D} unsupported")
    rp = np.asarray(rowptr, dtype=np.int64)
    n = rp.shape[0] - 1
    length = rp[1:] - rp[:-1]
    seg_len = int(seg_mult) * G * task_len
    k = np.maximum(1, -(-length // task_len))
    k = np.minimum(k, G)
    giant = np.flatnonzero(length > seg_len)
    multi = np.flatnonzero((k > 1) & (length <= seg_len))
    single = np.flatnonzero(k == 1)
    pad_row = np.array([[-1, 0, 0, 1]], dtype=np.int64)
    parts = []
    n_partial_blocks = 0
    fix = np.zeros((0, 4), dtype=np.int32)
    if giant.size:
        nseg = -(-length[giant] // seg_len)
        first = np.cumsum(nseg) - nseg
        total_seg = int(nseg.sum())
        seg_owner = np.repeat(np.arange(giant.size), nseg)
        seg_j = np.arange(total_seg) - first[seg_owner]
        sbeg = rp[giant][seg_owner] + seg_j * seg_len
        send = np.minimum(sbeg + seg_len, rp[giant + 1][seg_owner])
        chunk = np.maximum(1, -(-(send - sbeg) // G))
        sidx = np.repeat(np.arange(total_seg), G)
        gidx = np.tile(np.arange(G), total_seg)
        beg = sbeg[sidx] + gidx * chunk[sidx]
        end = np.minimum(beg + chunk[sidx], send[sidx])
        live = beg < end
        nlive = np.bincount(sidx[live], minlength=total_seg)
        t = np.full((total_seg * G, 4), -1, dtype=np.int64)
        t[:, 1:] = 0
        t[live, 0] = giant[seg_owner[sidx[live]]]
        t[live, 1] = beg[live]
        t[live, 2] = end[live]
        t[live, 3] = (gidx[live] << 16) | nlive[sidx[live]]
        t[~live, 3] = 1
        parts.append(t)
        n_partial_blocks = total_seg
        fix = np.stack([giant, first, nseg, np.zeros_like(giant)], axis=1).astype(np.int32)
    if multi.size:
        kp = 1 << np.ceil(np.log2(k[multi])).astype(np.int64)  # slots reserved per run (power of two <= G)
        order = np.argsort(-kp, kind="stable") if row_key is None else np.lexsort((np.asarray(row_key)[multi], -kp))
        multi, kp = multi[order], kp[order]
        km = np.minimum(kp, G)
        start = np.concatenate([[0], np.cumsum(km)[:-1]])
        total = int(km.sum())
        t = np.full((total, 4), -1, dtype=np.int64)
        t[:, 1:] = 0
        run_of = np.repeat(np.arange(multi.size), km)
        j = np.arange(total) - np.repeat(start, km)
        chunk = -(-length[multi] // km)  # all reserved slots are used: shorter tasks, still one run
        beg = rp[multi][run_of] + j * chunk[run_of]
        end = np.minimum(beg + chunk[run_of], rp[multi + 1][run_of])
        live = beg < end
        nlive = np.bincount(run_of[live], minlength=multi.size)  # live tasks are a prefix of the run (equal chunks)
        t[live, 0] = multi[run_of[live]]
        t[live, 1] = beg[live]
        t[live, 2] = end[live]
        t[live, 3] = (j[live] << 16) | nlive[run_of[live]]
        t[~live, 3] = 1
        pad = (-total) % G
        if pad:
            t = np.concatenate([t, np.tile(pad_row, (pad, 1))])
        parts.append(t)
    head = int(sum(p.shape[0] for p in parts))  # segments + split rows come first
    if single.size:
        if row_key is not None:
            single = single[np.argsort(np.asarray(row_key)[single], kind="stable")]
        t = np.stack([single, rp[single], rp[single + 1], np.ones_like(single)], axis=1)
        pad = (-t.shape[0]) % G
        if pad:
            t = np.concatenate([t, np.tile(pad_row, (pad, 1))])
        parts.append(t)
    out = np.ascontiguousarray(np.concatenate(parts).astype(np.int32)) if parts else np.tile(pad_row.astype(np.int32), (G, 1))
    if full:
        return Plan(out, head, n_partial_blocks, fix)
    return (out, head) if return_head else out


def check_plan(rowptr: np.ndarray, tasks: np.ndarray, D: int) -> None:
    """Host-side validation of a task list (run before any launch: a bad plan would read out of bounds)."""
    G = 1024 // D
    rp = np.asarray(rowptr, dtype=np.int64)
    n = rp.shape[0] - 1
    t = np.asarray(tasks, dtype=np.int64)
    assert t.ndim == 2 and t.shape[1] == 4 and t.shape[0] % G == 0 and t.shape[0] > 0, "task list shape"
    live = t[:, 0] >= 0
    r, b, e = t[live, 0], t[live, 1], t[live, 2]
    assert (r < n).all() and (b >= rp[r]).all() and (e <= rp[r + 1]).all() and (b <= e).all(), "task range outside its row"
    covered = np.zeros(n, dtype=np.int64)
    np.add.at(covered, r, e - b)
    assert np.array_equal(covered, rp[1:] - rp[:-1]), "tasks do not tile every row exactly"
    pos, ln = t[:, 3] >> 16, t[:, 3] & 0xFFFF
    slot = np.arange(t.shape[0]) % G
    assert (ln[live] >= 1).all() and (pos[live] < ln[live]).all(), "run fields"
    assert ((slot - pos + ln)[live] <= G).all(), "a run straddles a workgroup"
    heads = np.flatnonzero(live & (pos == 0))
    assert np.array_equal(np.unique(t[heads, 0]), np.arange(n)), "every row needs a run head"
    seg = 2 * G * 32
    multi_head = np.bincount(t[heads, 0], minlength=n) > 1
    assert not (multi_head & ((rp[1:] - rp[:-1]) <= G)).any(), "only giant rows may have several run heads (one per segment)"
    for h in heads[ln[heads] > 1][:64]:  # spot-check run contiguity
        m = ln[h]
        assert (t[h:h + m, 0] == t[h, 0]).all() and np.array_equal(pos[h:h + m], np.arange(m)), "run not contiguous"


_adjacency_uid = itertools.count(1)
TASK_LEN = int(os.environ.get("DMM_TASK_LEN", "16"))  # non-zeros per SpMM task (DMM_TASK_LEN: tuning knob).  At TikTok size the joint
# iteration takes 15.9 / 15.8 / 16.0 / 16.5 / 16.9 ms per epoch at 8 / 12 / 16 / 24 / 32: shorter tasks = more waves to hide the gather latency


@dataclass
class Adjacency:
    """A normalised adjacency resident in HBM: CSR arrays + the SpMM task list for one embedding width.
    `uid` is unique for the life of the process: recorded launch plans (which hold raw device pointers into these arrays) are keyed
    by it -- never by id(), which CPython reuses as soon as an object is collected."""
    n: int
    nnz: int
    D: int
    rowptr: torch.Tensor  # int32 [n+1]
    col: torch.Tensor  # int32 [nnz]
    val: torch.Tensor  # fp32 [nnz]
    tasks: torch.Tensor  # int32 [T, 4]
    n_head_tasks: int = 0  # leading tasks that belong to split rows / giant-row segments
    n_partial_blocks: int = 0  # leading workgroups that are segments of giant rows
    fix: Optional[torch.Tensor] = None  # int32 [n_giant, 4]: row, first_slot, n_slots, 0
    host: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None
    uid: int = field(default_factory=lambda: next(_adjacency_uid))
    vec: int = 4  # columns per lane the task list was planned for: 4 = fp32 operands (dmm_spmm_csr_f32), 8 = bf16 operands (dmm_spmm_csr_bf16)
    # A PERSISTENT adjacency (DeviceGraphBuilder(reuse=True): the rebuilt modality graphs) is rebuilt into the same device buffers epoch after epoch:
    # recorded launch plans stay valid -- their pointers do not move -- and only the SIZES they carry (n_tasks, n_head_tasks, n_partial_blocks, n_fix)
    # are patched when `meta_version` has moved (engine.patch_spmm_sizes); `version` moves when a buffer had to be reallocated (plans are dropped).
    persistent: bool = False
    version: int = 0
    meta_version: int = 0
    partial_cap: int = 0  # rows of the giant-row scratch buffers (persistent adjacencies allocate with headroom)

    @property
    def n_tasks(self) -> int:
        return int(self.tasks.shape[0])

    def partial_buffer(self, i: int) -> Optional[torch.Tensor]:
        """Scratch for the giant-row segments of the i-th concurrent use of this adjacency in one launch (owned here so that
        recorded launch plans keep valid pointers)."""
        if not self.n_partial_blocks and not self.persistent:
            return None
        pool = self.__dict__.setdefault("_partial_pool", [])
        self.ensure_capacity()
        rows = max(self.partial_cap, 64) if self.persistent else self.n_partial_blocks
        while len(pool) <= i:
            pool.append(torch.empty((rows, self.D), dtype=torch.float32, device=self.col.device))
        return pool[i]

    def ensure_capacity(self) -> None:
        """Persistent adjacencies: the giant-row scratch buffers hold max(partial_cap, 64) rows; a rebuild that needs more gets new, larger
        buffers and a new `version` (recorded plans, which hold the old pointers, are dropped by their owners).  Called by the builder after
        every in-place rebuild and before a buffer is handed out."""
        if self.persistent and self.n_partial_blocks > max(self.partial_cap, 64):
            self.partial_cap = 2 * self.n_partial_blocks
            self.__dict__.setdefault("_partial_pool", []).clear()
            self.version += 1

    def fix_counter(self, i: int) -> Optional[torch.Tensor]:
        """Arrival counters of the giant rows for the i-th concurrent use of this adjacency in one launch (dmm_spmm_problem.fix_count: zero
        between launches -- the kernel leaves them zero)."""
        if self.fix is None or (not self.n_partial_blocks and not self.persistent):
            return None
        pool = self.__dict__.setdefault("_fix_count_pool", [])
        cap = int(self.fix.shape[0]) if not self.persistent else max(1, self.fix.untyped_storage().nbytes() // 16)  # (persistent: the fix buffer's capacity)
        while len(pool) <= i:
            pool.append(torch.zeros(cap, dtype=torch.int32, device=self.col.device))
        return pool[i]

    def algorithmic_bytes(self, elem_bytes: int = 4) -> int:
        """Compulsory HBM traffic of one SpMM: indices + values + row pointers + X + Y."""
        return self.nnz * 8 + (self.n + 1) * 4 + 2 * self.n * self.D * elem_bytes


def spmm_operand_vec(n_rows: int, D: int) -> int:
    """Operand type of the joint phase's SpMMs: 8 = the gathered operand X is stored in bf16 (dmm_spmm_csr_bf16: fp32 accumulation, fp32
    epilogue operands and outputs), 4 = fp32.  bf16 gathers pay where the kernel is bound by the bytes that miss the caches: DMM_SPMM_BF16=auto
    (default) switches at an fp32 operand of >= 256 MiB (the Infinity Cache; 1M users x 100k items at d = 256: 1.83 -> 0.99 ms per
    SpMM, profiles/README.md) and never at the shipped datasets' sizes, where the launch is latency-bound; =1 / =0 force it."""
    env = os.environ.get("DMM_SPMM_BF16", "auto")
    if env in ("0", "1"):
        return 8 if env == "1" and D >= 16 else 4
    return 8 if D >= 16 and n_rows * D * 4 >= (1 << 28) else 4


def spmm_task_len(vec: int) -> int:
    """Non-zeros per SpMM task: TASK_LEN (16: the latency-bound launches of the shipped datasets' sizes), 64 on the graphs planned for bf16
    gather operands (vec = 8, i.e. operands beyond the caches): at 1M users x 100k items, d = 256 that cuts the rows longer than one workgroup's
    share ("giant" rows: partial sums + fix-up launch) from 14341 to 686 (SpMM 981 -> 968 us, the fix-up launch 197 -> 30 us)."""
    return TASK_LEN if vec == 4 or "DMM_TASK_LEN" in os.environ else 64


def make_adjacency(rows: np.ndarray, cols: np.ndarray, users: int, items: int, D: int, device, task_len: int = TASK_LEN,
                   keep_host: bool = False, vec: int = 4, hub_order: bool = False, seg_mult: int = 2) -> Adjacency:
    """hub_order: schedule the rows by their hub (hub_keys) instead of by row id.  seg_mult: plan_tasks."""
    rowptr, col, val = normalized_bipartite_csr(rows, cols, users, items)
    return adjacency_from_csr(rowptr, col, val, D, device, task_len, keep_host, vec, row_key=hub_keys(rowptr, col) if hub_order else None, seg_mult=seg_mult)


def adjacency_from_csr(rowptr, col, val, D: int, device, task_len: int = TASK_LEN, keep_host: bool = False, vec: int = 4,
                       row_key: Optional[np.ndarray] = None, seg_mult: int = 2) -> Adjacency:
    """vec = 8 plans the work list for bf16 operands: a lane owns 8 columns, so a workgroup holds 2048 / D task groups -- the plan of
    width D / 2.  row_key: schedule the rows in key order (plan_tasks)."""
    if vec not in (4, 8):
        raise ValueError("vec must be 4 (fp32 operands) or 8 (bf16 operands)")
    Dp = D * 4 // vec
    plan = plan_tasks(rowptr, Dp, task_len, full=True, row_key=row_key, seg_mult=seg_mult)
    tasks, head = plan.tasks, plan.head
    check_plan(rowptr, tasks, Dp)
    assert col.size == 0 or (col.min() >= 0 and col.max() < rowptr.shape[0] - 1), "column id out of range"
    dev = torch.device(device)
    return Adjacency(n=int(rowptr.shape[0] - 1), nnz=int(col.shape[0]), D=D,
                     rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev),
                     val=torch.from_numpy(val).to(dev), tasks=torch.from_numpy(tasks).to(dev), n_head_tasks=head,
                     n_partial_blocks=plan.n_partial_blocks, fix=torch.from_numpy(plan.fix).to(dev) if plan.n_partial_blocks else None,
                     host=(rowptr, col, val) if keep_host else None, vec=vec)


# ======================================================================================================
# Device-side construction (no host round trip): used for the modality graphs rebuilt every epoch.
# Index arithmetic runs as torch ops on the device the edge list already lives on (plumbing: sort / cumsum / scatter of
# ~1e5 integers); the values follow the same float64 formula as normalized_bipartite_csr.
# ======================================================================================================
class DeviceGraphBuilder:
    """Builds D^-1/2 (A + I) D^-1/2 CSR + task list on the device (csrc/graph_build.hip: dmm_graph_build) for edge lists that
    share the base graph's user degrees (the rebuild keeps k_u = deg_u, Main.py:215-230): `item_idx[user_ptr[u] : user_ptr[u+1]]`
    are user u's items, distinct and ascending.  Bit-identical to `make_adjacency` on the same edges (tests/test_host_logic.py);
    the only host read-back is the 32-byte size record."""

    def __init__(self, user_ptr: torch.Tensor, users: int, items: int, D: int, task_len: int = TASK_LEN, vec: int = 4, reuse: bool = False):
        """reuse: build_all() rebuilds graph i into the SAME device buffers and Adjacency object as its previous call (Adjacency.persistent): the
        engines' recorded launch plans survive the rebuild (re-recording the joint step's plan cost ~0.5 ms of host time per epoch at TikTok
        shape, with the GPU idle).  The previous build's graphs are overwritten: callers that need both generations build without reuse."""
        import ctypes as C
        from . import _lib as L
        if vec not in (4, 8):
            raise ValueError("vec must be 4 (fp32 operands) or 8 (bf16 operands)")
        self.reuse, self._slots = bool(reuse), []
        self.D_full, self.vec = D, vec
        D = D * 4 // vec  # (bf16 operands: a lane owns 8 columns -- the task plan of width D / 2, as in adjacency_from_csr)
        self.U, self.I, self.N, self.D, self.task_len = users, items, users + items, D, task_len
        self.dev = user_ptr.device
        up = user_ptr.long()
        self.E = int(up[-1])
        self.user_ptr32 = up.to(torch.int32).contiguous()
        self.edge_user = torch.repeat_interleave(torch.arange(users, device=self.dev, dtype=torch.int32), (up[1:] - up[:-1]), output_size=self.E).contiguous()
        # d^-0.5 in fp64, computed by numpy exactly like normalized_bipartite_csr does (index = row length incl. the self loop)
        with np.errstate(divide="ignore"):
            self.dinv = torch.from_numpy(np.power(np.arange(self.N + 2, dtype=np.float64), -0.5)).to(self.dev)
        sb, tc, fc = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().dmm_graph_build_scratch_bytes(users, items, self.E, D, C.byref(sb)), "dmm_graph_build_scratch_bytes")
        L.check(L.lib().dmm_graph_build_task_capacity(users, items, self.E, D, task_len, C.byref(tc), C.byref(fc)), "dmm_graph_build_task_capacity")
        self.scratch = torch.empty(sb.value, dtype=torch.uint8, device=self.dev)
        self._scratches, self._streams = [self.scratch], []  # one scratch / stream per graph of a build_all (grown on demand)
        self.task_cap, self.fix_cap = int(tc.value), int(fc.value)
        self.nnz = 2 * self.E + self.N

    def build(self, item_idx: torch.Tensor, validate: bool = False) -> Adjacency:
        return self.build_all([item_idx], validate)[0]

    def build_all(self, item_lists, validate: bool = False, ready=None):
        """One graph per edge list (the M modality graphs of a rebuild): all launches first, then ONE read-back of the size records.
        ready: per list an event (or None) behind the kernels that fill it on ANOTHER stream (DiffusionEngine.rebuild_edges(defer_join=True)): graph i's
        stream waits for its own list only, so a modality whose edges are ready early is built early."""
        from . import _lib as L
        dev, N, nnz = self.dev, self.N, self.nnz
        metas = torch.zeros((len(item_lists), 8), dtype=torch.int32, device=dev)
        outs = []
        for i, item_idx in enumerate(item_lists):
            if item_idx.dtype != torch.int32 or not item_idx.is_contiguous() or item_idx.numel() < self.E:
                raise ValueError("DeviceGraphBuilder.build: item_idx must be a contiguous int32 tensor with one entry per edge")
            if self.reuse and i < len(self._slots):
                outs.append(self._slots[i]["bufs"])
                continue
            outs.append((torch.empty(N + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev),
                         torch.empty(nnz, dtype=torch.float32, device=dev), torch.empty((self.task_cap, 4), dtype=torch.int32, device=dev),
                         torch.empty((max(self.fix_cap, 1), 4), dtype=torch.int32, device=dev)))
        # the graphs are independent and their 7 launches each are latency-bound (scans, per-row sorts): one HIP stream and one scratch
        # buffer per graph, forked from and joined to the caller's stream (all outputs were allocated on it above)
        n = len(item_lists)
        while len(self._scratches) < n:
            self._scratches.append(torch.empty_like(self.scratch))
        if dev.type == "cuda" and n > 1:
            while len(self._streams) < n:
                self._streams.append(torch.cuda.Stream(device=dev))
            main = torch.cuda.current_stream(dev)
            fork = torch.cuda.Event(); fork.record(main)
            for i, item_idx in enumerate(item_lists):
                st = self._streams[i]
                st.wait_event(fork)
                if ready is not None and ready[i] is not None:
                    st.wait_event(ready[i])
                rowptr, col, val, tasks, fix = outs[i]
                with torch.cuda.stream(st):
                    L.call("dmm_graph_build", L.ptr(self.user_ptr32), L.ptr(self.edge_user), L.ptr(item_idx), self.U, self.I, self.E, self.D, self.task_len,
                           L.ptr(self.dinv), L.ptr(rowptr), L.ptr(col), L.ptr(val), L.ptr(tasks), L.ptr(fix), L.ptr(metas[i]), L.ptr(self._scratches[i]),
                           L.stream())
                    done = torch.cuda.Event(); done.record(st)
                main.wait_event(done)
        else:
            if ready is not None:
                for e in ready:
                    if e is not None:
                        torch.cuda.current_stream(dev).wait_event(e)
            for i, item_idx in enumerate(item_lists):
                rowptr, col, val, tasks, fix = outs[i]
                L.call("dmm_graph_build", L.ptr(self.user_ptr32), L.ptr(self.edge_user), L.ptr(item_idx), self.U, self.I, self.E, self.D, self.task_len,
                       L.ptr(self.dinv), L.ptr(rowptr), L.ptr(col), L.ptr(val), L.ptr(tasks), L.ptr(fix), L.ptr(metas[i]), L.ptr(self._scratches[i]), L.stream())
        adjs = []
        for i, ((rowptr, col, val, tasks, fix), m) in enumerate(zip(outs, metas.tolist())):
            n_tasks, head, n_partial, n_fix, nnz_d = m[:5]
            if nnz_d != nnz or n_tasks > self.task_cap or n_fix > self.fix_cap:
                raise RuntimeError(f"dmm_graph_build: inconsistent sizes (nnz {nnz_d} vs {nnz}, tasks {n_tasks}/{self.task_cap}, fix {n_fix}/{self.fix_cap})")
            if validate:
                check_plan(rowptr.cpu().numpy().astype(np.int64), tasks[:n_tasks].cpu().numpy(), self.D)
            if self.reuse:
                if i < len(self._slots):  # the same Adjacency object, rebuilt in place: new sizes, same pointers
                    a = self._slots[i]["adj"]
                    a.tasks, a.n_head_tasks, a.n_partial_blocks, a.fix = tasks[:n_tasks], head, n_partial, fix[:n_fix]
                    a.meta_version += 1
                    a.ensure_capacity()
                else:
                    a = Adjacency(n=N, nnz=nnz, D=self.D_full, rowptr=rowptr, col=col, val=val, tasks=tasks[:n_tasks], n_head_tasks=head,
                                  n_partial_blocks=n_partial, fix=fix[:n_fix], vec=self.vec, persistent=True)
                    self._slots.append(dict(bufs=(rowptr, col, val, tasks, fix), adj=a))
                adjs.append(a)
            else:
                adjs.append(Adjacency(n=N, nnz=nnz, D=self.D_full, rowptr=rowptr, col=col, val=val, tasks=tasks[:n_tasks], n_head_tasks=head,
                                      n_partial_blocks=n_partial, fix=fix[:n_fix] if n_partial else None, vec=self.vec))
        return adjs


class ShardedGraphBuilder:
    """DeviceGraphBuilder for ONE replica of the user-sharded layout (csrc/graph_build.hip: dmm_graph_build_shard): builds the
    replica's local block over [own users ; all items] from the top-k item lists of its own users.  The item degrees that scale the values
    are global -- every replica counts how often its users picked each item (dmm_count_items), ONE integer all-reduce of the [M, I] counts
    sums them ("item-side degrees need one I-length integer all-reduce per modality") -- so the blocks carry the full graph's values and
    add up to it, each item self loop on exactly one replica (i % world == rank)."""

    def __init__(self, user_ptr_local: torch.Tensor, users_local: int, users_global: int, items: int, D: int, rank: int, world: int, group=None,
                 task_len: int = TASK_LEN, vec: int = 4):
        import ctypes as C
        from . import _lib as L
        if vec not in (4, 8):
            raise ValueError("vec must be 4 (fp32 operands) or 8 (bf16 operands)")
        self.D_full, self.vec = D, vec
        D = D * 4 // vec  # (bf16 operands: the task plan of width D / 2, as in adjacency_from_csr)
        self.nu, self.U, self.I, self.D, self.task_len = users_local, users_global, items, D, task_len
        self.rank, self.world, self.group = rank, world, group
        self.n = users_local + items
        self.dev = user_ptr_local.device
        up = user_ptr_local.long()
        self.E = int(up[-1])
        self.user_ptr32 = up.to(torch.int32).contiguous()
        self.edge_user = torch.repeat_interleave(torch.arange(users_local, device=self.dev, dtype=torch.int32), (up[1:] - up[:-1]),
                                                 output_size=self.E).contiguous()
        with np.errstate(divide="ignore"):  # an item can be picked by every user of the FULL graph: the table covers the global degrees
            self.dinv = torch.from_numpy(np.power(np.arange(users_global + items + 2, dtype=np.float64), -0.5)).to(self.dev)
        nu_q = max(1, users_local)  # the C ABI wants U > 0; a replica without users still holds the item rows
        sb, tc, fc = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().dmm_graph_build_scratch_bytes(nu_q, items, self.E, D, C.byref(sb)), "dmm_graph_build_scratch_bytes")
        L.check(L.lib().dmm_graph_build_task_capacity(nu_q, items, self.E, D, task_len, C.byref(tc), C.byref(fc)), "dmm_graph_build_task_capacity")
        self.scratch = torch.empty(sb.value, dtype=torch.uint8, device=self.dev)
        self.task_cap, self.fix_cap = int(tc.value), int(fc.value)
        self.nnz_cap = 2 * self.E + self.n

    def build_all(self, item_lists, validate: bool = False):
        from . import _lib as L
        if self.nu == 0:
            raise NotImplementedError("a replica without users: use fewer replicas than users")
        dev, n = self.dev, self.n
        M = len(item_lists)
        counts = torch.zeros((M, self.I), dtype=torch.int32, device=dev)
        for i, item_idx in enumerate(item_lists):
            if item_idx.dtype != torch.int32 or not item_idx.is_contiguous() or item_idx.numel() < self.E:
                raise ValueError("ShardedGraphBuilder.build_all: item_idx must be a contiguous int32 tensor with one entry per local edge")
            L.call("dmm_count_items", L.ptr(item_idx), self.E, self.I, L.ptr(counts[i]), L.stream())
        from . import comm
        if comm.active(self.world):
            comm.all_reduce(counts, group=self.group)
        metas = torch.zeros((M, 8), dtype=torch.int32, device=dev)
        outs = []
        for i, item_idx in enumerate(item_lists):
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            col = torch.empty(self.nnz_cap, dtype=torch.int32, device=dev)
            val = torch.empty(self.nnz_cap, dtype=torch.float32, device=dev)
            tasks = torch.empty((self.task_cap, 4), dtype=torch.int32, device=dev)
            fix = torch.empty((self.fix_cap, 4), dtype=torch.int32, device=dev)
            L.call("dmm_graph_build_shard", L.ptr(self.user_ptr32), L.ptr(self.edge_user), L.ptr(item_idx), self.nu, self.I, self.E, self.D, self.task_len,
                   L.ptr(self.dinv), L.ptr(counts[i]), self.world, self.rank, L.ptr(rowptr), L.ptr(col), L.ptr(val), L.ptr(tasks), L.ptr(fix),
                   L.ptr(metas[i]), L.ptr(self.scratch), L.stream())
            outs.append((rowptr, col, val, tasks, fix))
        adjs = []
        for (rowptr, col, val, tasks, fix), m in zip(outs, metas.tolist()):
            n_tasks, head, n_partial, n_fix, nnz_d = m[:5]
            if nnz_d > self.nnz_cap or n_tasks > self.task_cap or n_fix > self.fix_cap:
                raise RuntimeError(f"dmm_graph_build_shard: inconsistent sizes (nnz {nnz_d}/{self.nnz_cap}, tasks {n_tasks}/{self.task_cap}, fix {n_fix}/{self.fix_cap})")
            tasks = tasks[:n_tasks]
            if validate:
                check_plan(rowptr.cpu().numpy().astype(np.int64), tasks.cpu().numpy(), self.D)
            adjs.append(Adjacency(n=n, nnz=nnz_d, D=self.D_full, rowptr=rowptr, col=col[:nnz_d], val=val[:nnz_d], tasks=tasks, n_head_tasks=head,
                                  n_partial_blocks=n_partial, fix=fix[:n_fix] if n_partial else None, vec=self.vec))
        return adjs

"""Joint training step with users AND items sharded ("2D"; DESIGN 6, SURVEY 8e: "one all-reduce (or reduce-scatter + all-gather) of an I x d
partial per SpMM").  The user-sharded engine (sharded.py) replicates the item side: every propagation all-reduces a [I, d] fp32 partial
(2 (W-1)/W of it out of every GPU) and every replica repeats the item-side dense passes (epilogues, mixing, normalisation, Adam over I rows) --
at BASELINE config 5 (10M users x 1M items on 8 GPUs) that exchange costs as much as the step's compute.  Here replica g also OWNS the item slice
I_g = [g S, (g+1) S), S = ceil(I / W) -- the partition the item-sharded Denoise uses (engine.ItemShard) -- :

  * state: emb rows [own users ; own items] + their Adam state (1 / W of i_embs); every node tensor of a step is [own users ; own items];
  * a propagation Y = A X: the local launch (same kernel, same local adjacency block over rows [own users ; ALL items]) finishes the user rows in
    place and parks raw partials of all item rows; ONE reduce-scatter hands every replica the summed rows of its slice (half an all-reduce's
    bytes); the epilogue (scale / residual / perturbation / mean: dmm_spmm_epilogue_rows) runs on the slice only;
  * gather operands: a propagation reads X at ALL items (the user rows' neighbours), so an item-side tensor that feeds a later propagation is
    all-gathered into a full [W S, d] table -- in the operand's dtype: bf16 where the graphs are planned for bf16 gather operands (a quarter of
    the all-reduce's bytes), fp32 otherwise (parity mode);
  * minibatch: user rows as in sharded.py (own segment + one all-reduce of the batch table); the item rows of the contrasted views at pos / neg come
    from their owners the same way (masked gather into a zeroed table + one all-reduce); InfoNCE / BPR run on the tables on every replica (the
    reference's global in-batch negatives), the table gradients are scattered to the rows a replica owns and, for the two tensors that are
    propagated next (dFin, dCl1), into full sparse gather tables directly (every replica holds the whole table gradient);
  * dense item-side work (mixing, feature projection + normalisation and their backward, Adam) covers 1 / W of the items; replicated parameters
    are the projections and modal_weight only (their partial gradients ride one small all-reduce, as before).

Reference call sites: Model.py:60-134 (getItemEmbs / gcn_MM), Main.py:292-377 (the joint step), Utils/Utils.py:45-98 (losses).
Same arithmetic per row as JointEngine / ShardedJointEngine (the partials are summed in another order: parity to fp32 round-off).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import comm, graph, ops
from .engine import FlatParams, JointEngine, modality_names
from .sharded import ShardLayout, sort_batch_by_owner

Tensor = torch.Tensor


def _ar(t: Tensor, group) -> None:
    comm.all_reduce(t, group=group)


def _rs(out: Tensor, inp: Tensor, rank: int, world: int, group) -> None:
    comm.reduce_scatter(out, inp, rank, world, group=group)


def _ag(out: Tensor, t: Tensor, rank: int, group) -> None:
    comm.all_gather_into(out, t, rank, group=group)


class Sharded2DJointEngine:
    """JointEngine on the 2D layout.  world == 1 runs the same code (the collectives become copies)."""

    replicated_items = False  # (Coach: i_embs are sharded; the replicated slice of the flat parameters starts at `replicated_lo`)

    def __init__(self, layout: ShardLayout, D: int, feats: Sequence[Tensor], hp: dict, device, B_max: int, group=None):
        self.layout, self.group = layout, group
        self.world, self.rank = layout.world, layout.rank
        W, g = self.world, self.rank
        self.U, self.I, self.N = layout.users, layout.items, layout.users + layout.items
        self.S = -(-self.I // W)
        self.i0, self.i1 = min(g * self.S, self.I), min((g + 1) * self.S, self.I)
        self.ni = self.i1 - self.i0
        if (W - 1) * self.S >= self.I:
            raise ValueError(f"Sharded2DJointEngine: {self.I} items leave the last of {W} slices of {self.S} empty")
        self.nu = layout.nu
        self.n_adj = self.nu + self.I          # rows of the local adjacency blocks: [own users ; all items]
        self.n = self.nu + self.S              # rows of every local node tensor: [own users ; own item slice, padded to S rows]
        self.D, self.M = D, len(feats)
        self.hp = dict(hp)
        self.dev = torch.device(device)
        nu, S, ni, n, M, I = self.nu, self.S, self.ni, self.n, self.M, self.I
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        # the replica's slice of the raw modality features (constant), padded to S rows
        self.feats = []
        for f in feats:
            t = torch.zeros(S, f.shape[1], dtype=torch.float32, device=self.dev)
            t[:ni].copy_(f[self.i0:self.i1])
            self.feats.append(t)
        names = modality_names(M)
        shapes = [("emb", (n, D))]
        for nm, f in zip(names, self.feats):
            shapes += [(f"{nm}_layer.weight", (D, f.shape[1])), (f"{nm}_layer.bias", (D,))]
        shapes += [("modal_weight", (M,))]
        self.P = FlatParams(shapes, self.dev)
        self._feat_block, self._nf_block, self._fnorm_block = z(M, S, D), z(M, S, D), z(M, S)
        self.feat = [self._feat_block[m] for m in range(M)]
        self.nf = [self._nf_block[m] for m in range(M)]
        self.fnorm = [self._fnorm_block[m] for m in range(M)]
        self.Z = [z(n, D) for _ in range(M)]
        self.Bx, self.X1p, self.Mx, self.Fin, self.X2p, self.X3p, self.CL1 = (z(n, D) for _ in range(7))
        self.gblock = z(M + 3, n, D)
        self.dFin, self.dCl1, self.dCl2 = self.gblock[0], self.gblock[1], self.gblock[2]
        self.dZ = [self.gblock[3 + m] for m in range(M)]
        self.dMx, self.dX2p, self.dX1p, self.dBx = (z(n, D) for _ in range(4))
        self.T = [z(n, D) for _ in range(M)]
        self.dfeat = [z(S, D) for _ in range(M)]
        self.mix_partials = z(int(L.lib().dmm_gcn_mix_partials(n * D)))
        self.sumsq_scratch = z(int(L.lib().dmm_sumsq_scratch(n * D)))
        self.xraw = z(L.SPMM_MAX_BATCH, W * S, D)   # raw item partials of one launch (rows I .. W S stay zero): the reduce-scatter's input
        self.xown = z(L.SPMM_MAX_BATCH, S, D)       # ... and its output for the problems that still need an epilogue
        self.B_max = B_max
        self.nV = 3 + M                             # contrasted views: Fin, Z_m, CL1, X1p
        self.n_slots = 2 * self.nV
        self.ub, self.dub = z(self.nV, B_max, D), z(self.nV, B_max, D)          # user rows of the views (batch table) and its gradient
        self.ib, self.dib = z(self.nV + 1, B_max, D), z(self.nV + 1, B_max, D)  # item rows: [Fin@pos, Fin@neg, CL1@pos, Z_m@pos ..., X1p@pos]
        self.zhat, self.znorm, self.dzhat = z(self.n_slots, B_max, D), z(self.n_slots, B_max), z(self.n_slots, B_max, D)
        self.mfma_cl = D in (32, 64, 128, 256)
        self.pack = ops.infonce_pack(self.n_slots, B_max, D, self.dev) if self.mfma_cl else None
        pairs = M * (M - 1) // 2
        self.n_terms = 2 + (2 * pairs if self.hp["cl_method"] == 1 else 2 * M)
        self.lse = z(self.n_terms, B_max)
        self.row_loss = z(self.n_terms + 1, B_max)
        self.loss_parts = z(self.n_terms + 3)  # per-term InfoNCE, bpr, reg over the own users, reg over the own items (both partial sums)
        self.lr = float(hp["lr"])
        self.adj: Optional[graph.Adjacency] = None
        self.modal_adj: List[graph.Adjacency] = []
        self.rng_step = 0
        zi = lambda k: torch.zeros(k, dtype=torch.int32, device=self.dev)
        # static index buffers of the recorded plans: users as LOCAL ids (valid in the own segment), items as global ids and as local rows (-1: not owned)
        self.users_s, self.pos_s, self.neg_s, self.seg_s = zi(B_max), zi(B_max), zi(B_max), zi(2)
        self.pos_loc, self.neg_loc = zi(B_max), zi(B_max)
        self.arange = torch.arange(2 * B_max, dtype=torch.int32, device=self.dev)
        self._plans: Dict[tuple, dict] = {}
        self._philox_refs: Optional[list] = None
        self._tables: Dict[tuple, Tensor] = {}
        self._featsT = None
        self._emb16_step = -1  # (interface of the other engines: nothing to refresh here, the gather tables are rebuilt per step)

    # -- parameters ------------------------------------------------------------------------------------
    @property
    def u_embs(self) -> Tensor:
        return self.P.view("emb")[: self.nu]

    @property
    def i_embs_local(self) -> Tensor:
        """The replica's own rows of i_embs (items i0 .. i1)."""
        return self.P.view("emb")[self.nu:self.nu + self.ni]

    @property
    def i_embs(self) -> Tensor:
        """ALL item embeddings [I, d] (an all-gather of the replicas' slices: the diffusion phase and the evaluation read them)."""
        return self.gather_items(self.P.view("emb"), "i_embs_full", torch.float32)[: self.I]

    @property
    def replicated_lo(self) -> int:
        """Flat parameters from here on are replicated (projections, modal_weight); everything in front is this replica's own rows."""
        return self.n * self.D

    def load_reference_state(self, sd: Dict[str, Tensor]):
        t = {k: torch.as_tensor(v) for k, v in sd.items()}
        emb = torch.zeros(self.n, self.D)
        emb[: self.nu] = t.pop("u_embs")[self.layout.u0:self.layout.u1]
        emb[self.nu:self.nu + self.ni] = t.pop("i_embs")[self.i0:self.i1]
        t["emb"] = emb
        self.P.load(t)

    def state_dict(self) -> Dict[str, Tensor]:
        """Local view: u_embs are the replica's own users, i_embs its own item slice."""
        sd = self.P.state_dict()
        emb = sd.pop("emb")
        sd["u_embs"], sd["i_embs"] = emb[: self.nu].clone(), emb[self.nu:self.nu + self.ni].clone()
        return sd

    def local_rows(self, full: Tensor) -> Tensor:
        """[N, d] node tensor -> this replica's [own users ; own items (padded to S rows)] rows."""
        out = torch.zeros(self.n, full.shape[1], dtype=full.dtype, device=full.device)
        out[: self.nu] = full[self.layout.u0:self.layout.u1]
        out[self.nu:self.nu + self.ni] = full[self.U + self.i0:self.U + self.i1]
        return out

    def set_adjacency(self, adj: graph.Adjacency, modal_adj: Sequence[graph.Adjacency]):
        assert adj.n == self.n_adj and adj.D == self.D and len(modal_adj) == self.M and all(a.n == self.n_adj for a in modal_adj)
        if any(a.vec != adj.vec for a in modal_adj):
            raise ValueError("Sharded2DJointEngine.set_adjacency: all graphs of a step must be planned for the same operand type (Adjacency.vec)")
        self.adj, self.modal_adj = adj, list(modal_adj)
        live = (adj.uid, tuple(a.uid for a in self.modal_adj))
        for k in [k for k in self._plans if (k[2], k[3]) != live]:
            del self._plans[k]

    # -- exchange helpers (recorded into the launch plans as Python calls) ------------------------------------
    def _op_dtype(self):
        return torch.bfloat16 if (self.adj is not None and self.adj.vec == 8) else torch.float32

    def _table(self, name, dtype) -> Tensor:
        """A full [W S, d] gather table (rows I .. W S are padding)."""
        t = self._tables.get((name, dtype))
        if t is None:
            t = self._tables[(name, dtype)] = torch.zeros(self.world * self.S, self.D, dtype=dtype, device=self.dev)
        return t

    def gather_items(self, local: Tensor, name: str, dtype) -> Tensor:
        """The item rows of a local node tensor [n, d] (or an [S, d] item tensor) from every replica -> the table `name` [W S, d] in `dtype`."""
        rows = local[self.nu:self.nu + self.S] if local.shape[0] == self.n else local
        if dtype != torch.float32:
            tmp = self._tables.get(("_cast", dtype))
            if tmp is None:
                tmp = self._tables[("_cast", dtype)] = torch.zeros(self.S, self.D, dtype=dtype, device=self.dev)
            ops.cast_bf16(rows, tmp)
            rows = tmp
        tab = self._table(name, dtype)
        if comm.active(self.world):
            L.call_py(_ag, tab, rows, self.rank, self.group)
        else:
            L.call_py(_copy_rows, tab, rows)
        return tab

    def _user_operand(self, local: Tensor, slot: int, dtype) -> Tensor:
        """The own user rows of a local node tensor as a gather operand (a bf16 copy where the graphs want one)."""
        if dtype == torch.float32:
            return local[: self.nu]
        key = ("_u16", slot)
        t = self._tables.get(key)
        if t is None:
            t = self._tables[key] = torch.zeros(max(self.nu, 1), self.D, dtype=dtype, device=self.dev)
        if self.nu:
            ops.cast_bf16(local[: self.nu], t[: self.nu])
        return t[: self.nu]

    def _spmm(self, batch: List[dict], refs_key=None):
        """batch: dicts with adj, x_lo ([nu, d]), x_hi (a [W S, d] table), `plain` (y = A x and nothing else) and the epilogue keywords of ops.SpmmOp
        (outputs / operands are LOCAL tensors [n, d]).  Local launch -> reduce-scatter of the raw item partials -> epilogue on the own slice."""
        nu, S, ni, I = self.nu, self.S, self.ni, self.I
        ops_, plain = [], []
        rng = (nu, self.layout.u0, self.U - nu + self.i0)  # global row ids of the Philox draws: users u0 + r, items U + i0 + (r - nu)
        for i, b in enumerate(batch):
            b = dict(b)
            plain.append(bool(b.pop("plain", False)))
            x_hi = b.pop("x_hi")
            kw = dict(b)
            if "seed" in kw or "noise" in kw:
                kw.setdefault("rng_rows", rng)
            ops_.append(ops.SpmmOp(kw.pop("adj"), kw.pop("x_lo"), x_hi=x_hi[:I], split=nu, raw=self.xraw[i][:I], raw_from=nu, **kw))
        arr = ops.spmm_batched(ops_)
        # the own slice's sums: straight into y's item rows for plain problems, into xown otherwise
        outs = []
        for i, b in enumerate(batch):
            out = b["y"][nu:nu + S] if plain[i] else self.xown[i]
            outs.append(out)
            if comm.active(self.world):
                L.call_py(_rs, out, self.xraw[i], self.rank, self.world, self.group)
            else:
                L.call_py(_copy_rows, out, self.xraw[i][:S])
        if not all(plain):
            arr2 = (L.SpmmProblem * len(batch))()
            for i in range(len(batch)):
                C.memmove(C.byref(arr2[i]), C.byref(arr[i]), C.sizeof(L.SpmmProblem))
                arr2[i].raw, arr2[i].raw_from, arr2[i].n_rows = L.ptr(outs[i]), nu, nu + ni
            ops.spmm_epilogue_rows(arr2, len(batch), self.D)
            return arr, arr2, ops_
        return arr, None, ops_

    def _rng(self, k: int, noise) -> dict:
        lanes = self.D // 4
        if noise is not None:
            return dict(noise=noise[k])
        seed, base = self.hp.get("seed", 1818), self.rng_step * 3 * self.N * lanes
        return dict(noise=None, seed=seed, offset=base + k * self.N * lanes)

    _feat_kernels = JointEngine._feat_kernels

    # -- forward -----------------------------------------------------------------------------------------
    def _project(self, precision: int):
        """Linear_m(E_m) and its row normalisation on the OWN item slice; the normalised rows (gather operands of the modal graphs) are all-gathered."""
        ni, D, M = self.ni, self.D, self.M
        names = modality_names(M)
        if self._feat_kernels(precision):
            ops.zero_(self._feat_block)
            ops.item_project_batched([dict(X=self.feats[m], F=self.P.view(f"{nm}_layer.weight"), out=self.feat[m], bias=self.P.view(f"{nm}_layer.bias"),
                                           K=self.feats[m].shape[1], rows=ni) for m, nm in enumerate(names)], D, f_transposed=True)
        else:
            ops.gemm_batched([dict(A=self.feats[m][:ni], B=self.P.view(f"{nm}_layer.weight"), out=self.feat[m][:ni], trans_b=True,
                                   bias=self.P.view(f"{nm}_layer.bias"), precision=precision) for m, nm in enumerate(names)])
        ops.rownorm_fwd(self._feat_block.view(-1, D), self._nf_block.view(-1, D), self._fnorm_block.view(-1))
        dt = self._op_dtype()
        return [self.gather_items(self.nf[m], ("nf", m), dt) for m in range(M)]

    def all_feats(self) -> List[Tensor]:
        """The projected features of ALL items per modality, fp32 [I, d] (the row-split diffusion phase reads them; call after _project)."""
        return [self.gather_items(self.feat[m], ("feat_full", m), torch.float32)[: self.I] for m in range(self.M)]

    def forward(self, noise: Optional[Sequence[Tensor]] = None, with_cl: bool = True, precision: int = 3):
        """noise (parity tests): 3 tensors in the LOCAL row order [own users ; own items (S rows)]."""
        hp, A, M = self.hp, self.adj, self.M
        nd = hp["noise_degree"]
        dt = self._op_dtype()
        refs = self._philox_refs
        T_nf = self._project(precision)
        emb = self.P.view("emb")
        xu, T_emb = self._user_operand(emb, 0, dt), self.gather_items(emb, "emb", dt)
        self._spmm([dict(adj=self.modal_adj[m], x_lo=xu, x_hi=T_nf[m], y=self.Z[m], plain=True) for m in range(M)])
        base = dict(adj=A, x_lo=xu, x_hi=T_emb, y=self.Bx)
        if with_cl:
            base.update(y_pert=self.X1p, noise_degree=nd, **self._rng(0, noise))
        r = self._spmm([base])
        if refs is not None and with_cl:
            refs.append((r, 0, 0))
        ops.gcn_mix_fwd(self.Bx, self.Z, self.P.view("modal_weight"), hp["modal_adj_weight"], self.Mx)
        s = 1.0 + hp["residual_weight"]
        batch = [dict(adj=A, x_lo=self._user_operand(self.Mx, 0, dt), x_hi=self.gather_items(self.Mx, "Mx", dt), y=self.Fin, alpha=s, c1=self.Mx, beta=s)]
        if with_cl:
            batch.append(dict(adj=A, x_lo=self._user_operand(self.X1p, 1, dt), x_hi=self.gather_items(self.X1p, "X1p", dt), y_pert=self.X2p, noise_degree=nd,
                              **self._rng(1, noise)))
        r = self._spmm(batch)
        if refs is not None and with_cl:
            refs.append((r, 1, 1))
        if with_cl:
            r = self._spmm([dict(adj=A, x_lo=self._user_operand(self.X2p, 0, dt), x_hi=self.gather_items(self.X2p, "X2p", dt), y_pert=self.X3p, noise_degree=nd,
                                 y_mean=self.CL1, m1=self.X1p, m2=self.X2p, **self._rng(2, noise))])
            if refs is not None:
                refs.append((r, 0, 2))

    def final_item_embs(self) -> Tensor:
        """Fin's item rows of ALL items [I, d] (evaluation: Main.Coach.testEpoch)."""
        return self.gather_items(self.Fin, "Fin_full", torch.float32)[: self.I]

    def gcn_output(self) -> Dict[str, Tensor]:
        nu, ni = self.nu, self.ni
        out = {"u_final_embs": self.Fin[:nu], "i_final_embs": self.Fin[nu:nu + ni]}
        for m, nm in enumerate(modality_names(self.M)):
            out[f"u_{nm}_embs"], out[f"i_{nm}_embs"] = self.Z[m][:nu], self.Z[m][nu:nu + ni]
        return out

    # -- one optimiser step ------------------------------------------------------------------------------
    def _views(self):
        return [(self.Fin, self.dFin)] + [(self.Z[m], self.dZ[m]) for m in range(self.M)] + [(self.CL1, self.dCl1), (self.X1p, self.dCl2)]

    def _terms(self):
        """Slot 2 j = user rows of view j, slot 2 j + 1 = its item rows at pos (JointEngine's slot order: views Fin, Z_m, CL1, X1p)."""
        M, hp = self.M, self.hp
        c = 2 + 2 * M
        terms = [(c, c + 2, hp["cross_cl_temp"], hp["cross_cl_rate"]), (c + 1, c + 3, hp["cross_cl_temp"], hp["cross_cl_rate"])]
        tm, rm = hp["modal_cl_temp"], hp["modal_cl_rate"]
        if hp["cl_method"] == 1:
            for a in range(M):
                for b in range(a + 1, M):
                    terms += [(2 + 2 * a, 2 + 2 * b, tm, rm), (3 + 2 * a, 3 + 2 * b, tm, rm)]
        else:
            for m in range(M):
                terms += [(0, 2 + 2 * m, tm, rm), (1, 3 + 2 * m, tm, rm)]
        assert len(terms) == self.n_terms
        return ops.make_terms(terms)

    def set_batch(self, users: Tensor, pos: Tensor, neg: Tensor, bounds: Optional[Tensor] = None):
        if bounds is None:
            users, pos, neg, bounds = sort_batch_by_owner(self.layout, users, pos, neg)
        B = int(users.shape[0])
        assert B <= self.B_max
        self.users_s[:B].copy_(users.to(torch.int32) - self.layout.u0)
        p, q = pos.to(torch.int32), neg.to(torch.int32)
        self.pos_s[:B].copy_(p)
        self.neg_s[:B].copy_(q)
        own = lambda t: torch.where((t >= self.i0) & (t < self.i1), t - self.i0 + self.nu, torch.full_like(t, -1))
        self.pos_loc[:B].copy_(own(p))
        self.neg_loc[:B].copy_(own(q))
        self.seg_s.copy_(bounds[self.rank:self.rank + 2])
        return B

    def step(self, users: Tensor, pos: Tensor, neg: Tensor, noise: Optional[Sequence[Tensor]] = None, precision: int = 3, update: bool = True,
             bounds: Optional[Tensor] = None) -> Tensor:
        B = self.set_batch(users, pos, neg, bounds)
        if noise is None and update:
            return self._step_planned(B, precision)
        return self._step_body(B, noise, precision, update)

    def _step_body(self, B: int, noise, precision: int, update: bool) -> Tensor:
        D, nu, ni, S, M, hp, I = self.D, self.nu, self.ni, self.S, self.M, self.hp, self.I
        nS, nV, nT = self.n_slots, self.nV, self.n_terms
        dt = self._op_dtype()
        cview = lambda t, lead, *rest: t.view(-1)[: lead * B * int(np.prod(rest) if rest else 1)].view(lead, B, *rest)
        zhat, dzhat, znorm = cview(self.zhat, nS, D), cview(self.dzhat, nS, D), cview(self.znorm, nS)
        ub, dub = cview(self.ub, nV, D), cview(self.dub, nV, D)
        ib, dib = cview(self.ib, nV + 1, D), cview(self.dib, nV + 1, D)
        lse, row_loss = cview(self.lse, nT), cview(self.row_loss, nT + 1)
        users, pos, neg, ar = self.users_s[:B], self.pos_s[:B], self.neg_s[:B], self.arange[:B]
        pos_loc, neg_loc = self.pos_loc[:B], self.neg_loc[:B]
        ops.zero_(self.gblock)
        for t in (dzhat, ub, dub, ib, dib):
            ops.zero_(t)
        self.forward(noise=noise, with_cl=True, precision=precision)
        views = self._views()
        # ---- the minibatch tables: user rows from the own segment, item rows from the own slice, one all-reduce each ----
        ops.gather_rows(ops.Slots([(v[0], users, None) for v in views]), B, D, self.seg_s, ub)
        # item-table rows: [Fin@pos, Fin@neg, CL1@pos, Z_m@pos ..., X1p@pos] (the three that are propagated next lead, contiguous)
        item_rows = [(self.Fin, pos_loc, self.dFin), (self.Fin, neg_loc, self.dFin), (self.CL1, pos_loc, self.dCl1)] + \
                    [(self.Z[m], pos_loc, self.dZ[m]) for m in range(M)] + [(self.X1p, pos_loc, self.dCl2)]
        ops.gather_rows(ops.Slots([(t, ix, None) for t, ix, _ in item_rows]), B, D, None, ib)
        if comm.active(self.world):
            L.call_py(_ar, ub, self.group)
            L.call_py(_ar, ib, self.group)
        irow = lambda j: 0 if j == 0 else (2 if j == nV - 2 else (2 + j if j <= M else nV))  # table row of view j's item rows at pos
        entries = []
        for j in range(nV):
            entries += [(ub[j], ar, dub[j]), (ib[irow(j)], ar, dib[irow(j)])]
        slots = ops.Slots(entries)
        terms = self._terms()
        pk = dict(pack=self.pack, n_slots=nS, precision=precision) if self.mfma_cl else {}
        ops.gather_normalize(slots, B, D, zhat, znorm, self.pack)
        ops.infonce_fwd_bwd(zhat, terms, nT, B, D, lse, row_loss, dzhat, **pk)
        fin_i, dfin_i = ib[0:2].reshape(2 * B, D), dib[0:2].reshape(2 * B, D)  # [Fin@pos ; Fin@neg]
        ops.bpr_fwd_bwd_tables(ub[0], fin_i, ar, ar, self.arange[B:2 * B], row_loss[nT], dub[0], dfin_i)
        ops.segment_sum(row_loss, nT + 1, B, 1.0, self.loss_parts)
        emb = self.P.view("emb")
        if nu:
            ops.sumsq(emb[:nu], hp["reg"], self.loss_parts[nT + 1:], self.sumsq_scratch)
        else:
            ops.zero_(self.loss_parts[nT + 1:nT + 2])
        ops.sumsq(emb[nu:nu + ni], hp["reg"], self.loss_parts[nT + 2:], self.sumsq_scratch)
        # ---- backward: table gradients -> owned rows (+ the sparse gather tables of the two tensors that are propagated next) ----
        ops.scatter_normalize_bwd(slots, B, D, zhat, znorm, dzhat)
        ops.scatter_add_rows(ops.Slots([(v[0], users, v[1]) for v in views]), B, D, self.seg_s, dub)
        ops.scatter_add_rows(ops.Slots(item_rows), B, D, None, dib)
        # full sparse gather tables of dFin / dCl1 item rows (every replica holds the whole dib): fp32 scatter, cast where the graphs want bf16
        gF, gC = self._table("gFin32", torch.float32), self._table("gCl1_32", torch.float32)
        ops.zero_(gF); ops.zero_(gC)
        ops.scatter_add_rows(ops.Slots([(gF, pos, gF), (gF, neg, gF), (gC, pos, gC)]), B, D, None, dib[0:3])
        if dt != torch.float32:
            gF16, gC16 = self._table("gFin", dt), self._table("gCl1", dt)
            ops.cast_bf16(gF, gF16); ops.cast_bf16(gC, gC16)
            gF, gC = gF16, gC16
        A = self.adj
        s = 1.0 + hp["residual_weight"]
        self._spmm([dict(adj=A, x_lo=self._user_operand(self.dFin, 0, dt), x_hi=gF, y=self.dMx, alpha=s, c1=self.dFin, beta=s),
                    dict(adj=A, x_lo=self._user_operand(self.dCl1, 1, dt), x_hi=gC, y=self.dX2p, alpha=1.0 / 3.0, c1=self.dCl1, beta=1.0 / 3.0)])
        self._spmm([dict(adj=A, x_lo=self._user_operand(self.dX2p, 0, dt), x_hi=self.gather_items(self.dX2p, "dX2p", dt), y=self.dX1p, alpha=1.0, c1=self.dCl1,
                         beta=1.0 / 3.0, c2=self.dCl2, gamma=1.0)])
        ops.gcn_mix_bwd(self.dMx, self.Bx, self.Z, self.dZ, self.dX1p, self.P.view("modal_weight"), hp["modal_adj_weight"], dbx=self.dBx,
                        dw=self.P.g("modal_weight"), partials=self.mix_partials, count=(nu * D, nu * D, (nu + ni) * D))
        self._spmm([dict(adj=self.modal_adj[m], x_lo=self._user_operand(self.dZ[m], m, dt), x_hi=self.gather_items(self.dZ[m], ("dZ", m), dt),
                         y=self.T[m], plain=True) for m in range(M)])
        self._spmm([dict(adj=A, x_lo=self._user_operand(self.dBx, 0, dt), x_hi=self.gather_items(self.dBx, "dBx", dt), y=self.P.g("emb"), plain=True)])
        # ---- projection backward on the own slice; partial gradients of the replicated parameters: one small all-reduce ----
        ne = self.n * D
        names = modality_names(M)
        mw = self.P.offsets["modal_weight"]
        ops.zero_(self.P.grad[ne:mw])
        ops.rownorm_bwd_batched([dict(dy=self.T[m][nu:nu + ni], y=self.nf[m][:ni], norm=self.fnorm[m][:ni], dx=self.dfeat[m][:ni],
                                      csum=self.P.g(f"{nm}_layer.bias")) for m, nm in enumerate(names)])
        if self._feat_kernels(precision):
            FT = self._features_T_slice()
            ops.item_project_batched([dict(X=FT[m], F=self.dfeat[m][:ni], out=self.P.g(f"{nm}_layer.weight"), K=ni) for m, nm in enumerate(names)], D,
                                     out_transposed=True)
        else:
            ops.gemm_batched([dict(A=self.dfeat[m][:ni], B=self.feats[m][:ni], out=self.P.g(f"{nm}_layer.weight"), trans_a=True, precision=precision,
                                   split_k=128) for m, nm in enumerate(names)])
        if comm.active(self.world):
            L.call_py(_ar, self.P.grad[ne:], self.group)
        if update:
            self.apply_adam()
        self.rng_step += 1
        return self.loss_parts

    def _features_T_slice(self) -> List[Tensor]:
        if self._featsT is None:
            Kp = (max(self.ni, 1) + 31) & ~31
            self._featsT = []
            for f in self.feats:
                t = torch.zeros(f.shape[1], Kp, dtype=torch.float32, device=self.dev)
                t[:, : self.ni].copy_(f[: self.ni].t())
                self._featsT.append(t)
        return self._featsT

    def _step_planned(self, B: int, precision: int) -> Tensor:
        key = (B, precision, self.adj.uid, tuple(a.uid for a in self.modal_adj))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 8:
                self._plans.clear()
            self._philox_refs = []
            try:
                with L.Recorder() as calls:
                    self._step_body(B, None, precision, True)
                plan = dict(calls=calls, philox=self._philox_refs, adam=[c for c in calls if c[2] in L.ADAM_CALLS])
            finally:
                self._philox_refs = None
            self.rng_step -= 1
            self.P.step -= 1
            self._plans[key] = plan
        lanes = self.D // 4
        base = self.rng_step * 3 * self.N * lanes
        for (arr, arr2, _ops), i, k in plan["philox"]:
            arr[i].offset = base + k * self.N * lanes
            if arr2 is not None:
                arr2[i].offset = base + k * self.N * lanes
        self.P.step += 1
        for c in plan["adam"]:
            c[1][9], c[1][13] = self.lr, self.P.step
        L.replay(plan["calls"])
        self.rng_step += 1
        return self.loss_parts

    def apply_adam(self):
        P, nu, D = self.P, self.nu, self.D
        P.step += 1
        ops.adam_step(P.data, P.grad, P.m, P.v, self.lr, P.step, extra=list(self.T), extra_n=[nu * D] * self.M, l2_coef=2.0 * self.hp["reg"], l2_n=self.n * D)

    def full_grads(self) -> Dict[str, Tensor]:
        """Gradients as autograd reports them on the unsharded model: u_embs = own users' rows, i_embs = own item slice."""
        nu, ni = self.nu, self.ni
        g = {k: self.P.g(k).clone() for k in self.P.names}
        emb = g.pop("emb") + 2.0 * self.hp["reg"] * self.P.view("emb")
        for t in self.T:
            emb[:nu] += t[:nu]
        g["u_embs"], g["i_embs"] = emb[:nu], emb[nu:nu + ni]
        return g

    def losses(self) -> Dict[str, float]:
        lp = self.loss_parts.double().clone()
        nT = self.n_terms
        if comm.active(self.world):
            _ar(lp[nT + 1:nT + 3], self.group)
        lp = lp.cpu()
        cl, bpr, reg = float(lp[:nT].sum()), float(lp[nT]), float(lp[nT + 1] + lp[nT + 2])
        return {"loss": bpr + reg + cl, "bpr": bpr, "reg": reg, "cl": cl}


def _copy_rows(dst: Tensor, src: Tensor) -> None:
    dst[: src.shape[0]].copy_(src)

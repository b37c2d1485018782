"""User-sharded training engines (BASELINE.json north_star: "DataHandler shards users across the 8 GPUs ... gradient
all-reduce on RCCL over xGMI overlapped with the backward SpMM").  The reference is single-process; the call sites that get
sharded are the SpMMs of Model.py:90,93,105,111,114,123,130 and Main.py:319, the minibatch gathers of Main.py:305-307 and the
per-user loops of Main.py:145-253.

Layout.  Replica g owns a contiguous user range [u0, u1) (balanced by edges, shard.user_ranges): the rows u_embs[u0:u1], their
Adam state, the user rows of every adjacency and the matching columns of the item rows, as ONE local CSR over the node order
[own users ; all items].  Items -- i_embs, features, projections, modal_weight -- are replicated and stay bit-identical: every
replicated parameter only ever receives all-reduced gradients.

One joint step (same arithmetic, same number of optimiser steps as the single-GPU engine; the minibatch is the reference's
B shuffled interactions, every replica holds the user rows of the triples whose user it owns):
  * every SpMM launch  = local dmm_spmm_csr_f32 (own user rows finished in place, item rows parked raw) -> ONE all-reduce of the
    [problems, I, d] raw item partials -> dmm_spmm_epilogue_rows (scale / residual / perturbation / mean on the summed item rows);
    forward and backward alike (A is symmetric), 6 launches per step;
  * the minibatch: each replica gathers its segment of the B user rows of the 3 + M user-side views (dmm_gather_rows), ONE all-reduce
    assembles the [views, B, d] batch table everywhere; InfoNCE (global in-batch negatives, exactly the reference's B x B logits) and
    BPR then run on every replica against the replicated item rows -- their item-row gradients are complete without communication,
    the user-row gradients of the own segment are scattered back (dmm_scatter_add_rows);
  * replicated-parameter gradients: i_embs = item rows of the last backward SpMM (already all-reduced); each replica runs the
    modality projections' backward on ITS slice of the items and counts its own users + that slice towards d(modal_weight);
    ONE all-reduce of the small tail of the flat gradient buffer sums them;
  * Adam: own user rows + replicated rows, identical on every replica.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import comm, graph, ops, shard
from .engine import FlatParams, JointEngine, modality_names

Tensor = torch.Tensor


# ======================================================================================================
@dataclass
class ShardLayout:
    """Who owns what: contiguous user ranges balanced by edges; a contiguous slice of the (replicated) items per replica for the
    work that is split rather than repeated (projection backward, d(modal_weight))."""
    world: int
    rank: int
    users: int
    items: int
    cuts: np.ndarray  # int64 [world + 1]: replica g owns users cuts[g] .. cuts[g+1]

    @staticmethod
    def make(trn_rows: np.ndarray, users: int, items: int, world: int, rank: int) -> "ShardLayout":
        r = shard.user_ranges(trn_rows, users, world)
        cuts = np.asarray([a for a, _ in r] + [r[-1][1]], dtype=np.int64)
        return ShardLayout(world, rank, users, items, cuts)

    @property
    def u0(self) -> int:
        return int(self.cuts[self.rank])

    @property
    def u1(self) -> int:
        return int(self.cuts[self.rank + 1])

    @property
    def nu(self) -> int:
        return self.u1 - self.u0

    @property
    def n_local(self) -> int:
        return self.nu + self.items

    def item_slice(self, rank: Optional[int] = None):
        g = self.rank if rank is None else rank
        return (self.items * g) // self.world, (self.items * (g + 1)) // self.world

    def owner_of(self, users: Tensor) -> Tensor:
        """Replica that owns each user id (device op)."""
        inner = torch.as_tensor(self.cuts[1:-1], device=users.device, dtype=users.dtype)
        return torch.bucketize(users, inner, right=True)

    def owner_of_np(self, users: np.ndarray) -> np.ndarray:
        return np.searchsorted(self.cuts[1:-1], users, side="right")

    def local_rows(self, full: Tensor) -> Tensor:
        """[N, d] node tensor -> this replica's [own users ; all items] rows."""
        return torch.cat([full[self.u0:self.u1], full[self.users:]], dim=0).contiguous()


def sort_batch_by_owner(layout: ShardLayout, users: Tensor, pos: Tensor, neg: Tensor):
    """Stable reorder of one minibatch so that every replica's triples are contiguous (all replicas compute the same order from the same
    batch).  Returns (users, pos, neg, bounds) with bounds int32 [world + 1]: replica g owns positions bounds[g] .. bounds[g+1]."""
    owner = layout.owner_of(users.long())
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=layout.world)
    bounds = torch.zeros(layout.world + 1, dtype=torch.int32, device=users.device)
    bounds[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return users[order].contiguous(), pos[order].contiguous(), neg[order].contiguous(), bounds


def _all_reduce(t: Tensor, group=None) -> None:
    comm.all_reduce(t, group=group)


# ======================================================================================================
class ShardedJointEngine:
    """JointEngine (engine.py) on the user-sharded layout.  world == 1 runs the same code without communication."""

    def __init__(self, layout: ShardLayout, D: int, feats: Sequence[Tensor], hp: dict, device, B_max: int, group=None):
        self.layout, self.group = layout, group
        self.world, self.rank = layout.world, layout.rank
        self.U, self.I, self.N = layout.users, layout.items, layout.users + layout.items  # global sizes
        self.nu, self.n = layout.nu, layout.n_local
        self.D, self.M = D, len(feats)
        self.hp = dict(hp)
        self.dev = torch.device(device)
        self.feats = [f.to(self.dev, torch.float32).contiguous() for f in feats]
        names = modality_names(self.M)
        shapes = [("emb", (self.n, D))]
        for nm, f in zip(names, self.feats):
            shapes += [(f"{nm}_layer.weight", (D, f.shape[1])), (f"{nm}_layer.bias", (D,))]
        shapes += [("modal_weight", (self.M,))]
        self.P = FlatParams(shapes, self.dev)
        n, M, I = self.n, self.M, self.I
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self._feat_block = z(M, I, D)
        self.feat = [self._feat_block[m] for m in range(M)]
        self._nf_block, self._fnorm_block = z(M, I, D), z(M, I)  # blocks: the modalities' row normalisation is one launch
        self.nf = [self._nf_block[m] for m in range(M)]
        self.fnorm = [self._fnorm_block[m] for m in range(M)]
        self.Z = [z(n, D) for _ in range(M)]
        self.Bx, self.X1p, self.Mx, self.Fin, self.X2p, self.X3p, self.CL1 = (z(n, D) for _ in range(7))
        self.gblock = z(M + 3, n, D)
        self.dFin, self.dCl1, self.dCl2 = self.gblock[0], self.gblock[1], self.gblock[2]
        self.dZ = [self.gblock[3 + m] for m in range(M)]
        self.dMx, self.dX2p, self.dX1p, self.dBx = (z(n, D) for _ in range(4))
        self.T = [z(n, D) for _ in range(M)]
        self.dfeat = [z(I, D) for _ in range(M)]
        self.mix_partials = z(int(L.lib().dmm_gcn_mix_partials(n * D)))
        self.sumsq_scratch = z(int(L.lib().dmm_sumsq_scratch(n * D)))
        self.xraw = z(L.SPMM_MAX_BATCH, I, D)  # raw item partials of one SpMM launch: the all-reduce buffer
        self.B_max = B_max
        self.n_slots = 2 + 2 * M + 4
        self.n_uslots = self.n_slots // 2
        self.ub, self.dub = z(self.n_uslots, B_max, D), z(self.n_uslots, B_max, D)  # batch table of the user-side views and its gradient
        self.zhat, self.znorm, self.dzhat = z(self.n_slots, B_max, D), z(self.n_slots, B_max), z(self.n_slots, B_max, D)
        self.mfma_cl = D in (32, 64, 128, 256)
        self.pack = ops.infonce_pack(self.n_slots, B_max, D, self.dev) if self.mfma_cl else None
        pairs = M * (M - 1) // 2
        self.n_terms = 2 + (2 * pairs if self.hp["cl_method"] == 1 else 2 * M)
        self.lse = z(self.n_terms, B_max)
        self.row_loss = z(self.n_terms + 1, B_max)
        self.loss_parts = z(self.n_terms + 3)  # per-term InfoNCE, bpr, reg over the own users (a partial sum), reg over the items
        self.lr = float(hp["lr"])
        self.adj: Optional[graph.Adjacency] = None
        self.modal_adj: List[graph.Adjacency] = []
        self.rng_step = 0
        zi = lambda k: torch.zeros(k, dtype=torch.int32, device=self.dev)
        # static index buffers of the recorded launch plans: users as LOCAL row ids (valid inside the own segment), items global
        self.users_s, self.pos_s, self.neg_s, self.seg_s = zi(B_max), zi(B_max), zi(B_max), zi(2)
        self.arange = torch.arange(B_max, dtype=torch.int32, device=self.dev)
        self._plans: Dict[tuple, dict] = {}
        self._philox_refs: Optional[list] = None
        self._x16_bufs: Optional[Dict[str, Tensor]] = None
        self._emb16_step = -1

    # -- parameters ------------------------------------------------------------------------------------
    @property
    def u_embs(self) -> Tensor:
        return self.P.view("emb")[: self.nu]

    @property
    def i_embs(self) -> Tensor:
        return self.P.view("emb")[self.nu:]

    def load_reference_state(self, sd: Dict[str, Tensor]):
        """`sd` holds the FULL model (as the reference names it); the replica keeps its own user rows."""
        t = {k: torch.as_tensor(v) for k, v in sd.items()}
        u = t.pop("u_embs")[self.layout.u0:self.layout.u1]
        t["emb"] = torch.cat([u, t.pop("i_embs")], dim=0)
        self.P.load(t)
        self._emb16_step = -1

    def state_dict(self) -> Dict[str, Tensor]:
        """Local view: u_embs are the replica's OWN rows (users u0..u1)."""
        sd = self.P.state_dict()
        emb = sd.pop("emb")
        sd["u_embs"], sd["i_embs"] = emb[: self.nu].clone(), emb[self.nu:].clone()
        return sd

    def set_adjacency(self, adj: graph.Adjacency, modal_adj: Sequence[graph.Adjacency]):
        assert adj.n == self.n and adj.D == self.D and len(modal_adj) == self.M and all(a.n == self.n for a in modal_adj)
        if any(a.vec != adj.vec for a in modal_adj):
            raise ValueError("ShardedJointEngine.set_adjacency: all graphs of a step must be planned for the same operand type (Adjacency.vec)")
        self.adj, self.modal_adj = adj, list(modal_adj)
        live = (adj.uid, tuple(a.uid for a in self.modal_adj))
        for k in [k for k in self._plans if (k[2], k[3]) != live]:
            del self._plans[k]

    # -- one SpMM launch on the sharded layout -------------------------------------------------------------
    def _x16(self) -> Optional[Dict[str, Tensor]]:
        """bf16 copies of the SpMMs' gathered operands when the local blocks are planned for them (Adjacency.vec == 8; JointEngine._x16): the item rows'
        copies are finished by dmm_spmm_epilogue_rows together with their fp32 rows, so the replicas' copies stay bit-identical."""
        if self.adj is None or self.adj.vec != 8:
            return None
        if self._x16_bufs is None:
            zb = lambda *s_: torch.zeros(s_, dtype=torch.bfloat16, device=self.dev)
            n, M, D = self.n, self.M, self.D
            self._x16_bufs = dict(emb=zb(n, D), nf=zb(M, self.I, D), Mx=zb(n, D), X1p=zb(n, D), X2p=zb(n, D), g=zb(M + 3, n, D), dX2p=zb(n, D), dBx=zb(n, D))
        return self._x16_bufs

    def _ar(self, t: Tensor):
        if comm.active(self.world):
            L.call_py(_all_reduce, t, self.group)

    _PLAIN = frozenset(("adj", "x_lo", "x_hi", "split", "y"))

    def _in_place(self) -> bool:
        """Large blocks (those planned for bf16 gather operands; DMM_SHARD_INPLACE=0/1 forces it): a PLAIN problem (y = A x and nothing else: the
        modal graphs' forward and backward, the base graph's last backward -- 7 of a step's 14) parks its raw item partials in y's own item rows,
        they are all-reduced there and need no epilogue pass: no [I, d] scratch round trip (2 GB of traffic per problem at 1M items, d = 256).
        Small blocks keep ONE scratch buffer per launch: one collective instead of one per problem."""
        env = os.environ.get("DMM_SHARD_INPLACE")
        return (env == "1") if env in ("0", "1") else (self.adj is not None and self.adj.vec == 8)

    def _spmm(self, batch: List[dict]):
        """batch: SpmmOp keyword dicts (with adj, x_lo).  Local launch -> all-reduce of the raw item partials -> item-row epilogue."""
        arr, bufs, epi = self._spmm_local(batch, 0)
        for t in bufs:
            self._ar(t)
        if epi:
            ops.spmm_epilogue_rows(arr, len(batch), self.D)
        return arr

    def _spmm_local(self, batch: List[dict], slot0: int):
        """The local launch alone: own user rows finished, raw item partials parked in xraw[slot0 : slot0 + len(batch)] -- or, for plain problems in
        in-place mode, in the item rows of their own output.  Returns (argument block, the buffers to all-reduce, whether any problem needs
        dmm_spmm_epilogue_rows afterwards)."""
        nu, ops_, bufs, run = self.nu, [], [], None
        inplace = self._in_place()
        for i, b in enumerate(batch):
            if inplace and set(b) <= self._PLAIN:
                raw = b["y"][nu:]
                bufs.append(raw)
                run = None
            else:
                raw = self.xraw[slot0 + i]
                if run is not None and run[1] == slot0 + i:  # consecutive scratch slots travel as one collective
                    run[1] += 1
                else:
                    run = [slot0 + i, slot0 + i + 1]
                    bufs.append(run)
            ops_.append(ops.SpmmOp(raw=raw, raw_from=nu, **b))
        epi = any(isinstance(t, list) for t in bufs)
        bufs = [self.xraw[t[0]:t[1]] if isinstance(t, list) else t for t in bufs]
        return ops.spmm_batched(ops_), bufs, epi

    # The exchange of one launch's partials can run on a side stream UNDER the next launch (RCCL uses its own stream anyway; the side
    # stream only orders it behind the producing launch without blocking the main stream): the gradient all-reduces of the backward --
    # the item rows of the modal SpMMs that feed the projection gradients, then the item rows of the base SpMM = d(i_embs) -- overlap
    # with the remaining backward SpMM and with the projection backward.
    def _exchange_begin(self, bufs: List[Tensor]):
        if comm.active(self.world):
            L.call_py(self._exchange_begin_now, bufs)

    def _exchange_end(self):
        if comm.active(self.world):
            L.call_py(self._exchange_end_now)

    def _exchange_begin_now(self, bufs: List[Tensor]):
        if not hasattr(self, "_comm_stream"):
            self._comm_stream, self._comm_event = torch.cuda.Stream(device=self.dev), torch.cuda.Event()
        self._comm_event.record(torch.cuda.current_stream())
        self._comm_stream.wait_event(self._comm_event)
        with torch.cuda.stream(self._comm_stream):
            self._work = [comm.all_reduce(t, group=self.group, async_op=True) for t in bufs]

    def _exchange_end_now(self):
        for w in self._work:
            w.wait()
        torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._work = None

    def _rng(self, k: int, noise) -> dict:
        lanes = self.D // 4
        if noise is not None:
            return dict(noise=noise[k])
        seed, base = self.hp.get("seed", 1818), self.rng_step * 3 * self.N * lanes
        return dict(noise=None, seed=seed, offset=base + k * self.N * lanes, rng_rows=(self.nu, self.layout.u0, self.U - self.nu))

    # -- forward of gcn_MM (+ the 3 perturbed CL layers) ---------------------------------------------------
    def _project(self, precision: int):
        # replicated: every replica needs all normalised item features as SpMM operand; one launch for the modalities' projections
        if self._feat_kernels(precision):  # (at most two K slices: order-independent, so the replicas' copies stay bit-identical)
            ops.zero_(self._feat_block)
            ops.item_project_batched([dict(X=self.feats[m], F=self.P.view(f"{nm}_layer.weight"), out=self.feat[m], bias=self.P.view(f"{nm}_layer.bias"),
                                           K=self.feats[m].shape[1]) for m, nm in enumerate(modality_names(self.M))], self.D, f_transposed=True)
        else:
            ops.gemm_batched([dict(A=self.feats[m], B=self.P.view(f"{nm}_layer.weight"), out=self.feat[m], trans_b=True, bias=self.P.view(f"{nm}_layer.bias"),
                                   precision=precision) for m, nm in enumerate(modality_names(self.M))])
        x16 = self._x16()
        ops.rownorm_fwd(self._feat_block.view(-1, self.D), self._nf_block.view(-1, self.D), self._fnorm_block.view(-1),
                        y16=x16["nf"].view(-1, self.D) if x16 is not None else None)  # all modalities' rows (+ their bf16 gather copy)

    _feat_kernels = JointEngine._feat_kernels
    _emb16_refresh = JointEngine._emb16_refresh

    def _features_T_slice(self) -> List[Tensor]:
        """E_m^T restricted to this replica's item slice, [F_m, round_up(i1 - i0, 32)] (constant)."""
        if getattr(self, "_featsT", None) is None:
            i0, i1 = self.layout.item_slice()
            Kp = (max(i1 - i0, 1) + 31) & ~31
            self._featsT = []
            for f in self.feats:
                t = torch.zeros(f.shape[1], Kp, dtype=torch.float32, device=self.dev)
                t[:, : i1 - i0].copy_(f[i0:i1].t())
                self._featsT.append(t)
        return self._featsT

    def forward(self, noise: Optional[Sequence[Tensor]] = None, with_cl: bool = True, precision: int = 3):
        """noise (parity tests): 3 tensors already in the LOCAL row order [own users ; items]."""
        hp, A, emb = self.hp, self.adj, self.P.view("emb")
        nd = hp["noise_degree"]
        self._project(precision)
        refs = self._philox_refs
        M = self.M
        x16 = self._x16()
        if x16 is None:
            xe, xu, xnf, xMx, xX1, xX2, o1, o2, mo = emb, self.u_embs, self.nf, self.Mx, self.X1p, self.X2p, {}, {}, {}
        else:  # bf16 gather operands (JointEngine._forward_x16): casts of the embedding table and the normalised features, producer-made copies behind them
            self._emb16_refresh(x16)  # (a recorded step relies on Adam's copy: dmm_adam_step_x16)
            xe, xu, xnf, xMx, xX1, xX2 = x16["emb"], x16["emb"][: self.nu], [x16["nf"][m] for m in range(M)], x16["Mx"], x16["X1p"], x16["X2p"]
            o1, o2, mo = dict(y_pert16=x16["X1p"]), dict(y_pert16=x16["X2p"]), dict(mx16=x16["Mx"])
        modal = [dict(adj=self.modal_adj[m], x_lo=xu, y=self.Z[m], x_hi=xnf[m], split=self.nu) for m in range(M)]
        base = dict(adj=A, x_lo=xe, y=self.Bx, y_pert=self.X1p, noise_degree=nd, **o1, **self._rng(0, noise)) if with_cl else dict(adj=A, x_lo=xe, y=self.Bx)
        # the modal graphs' partials are exchanged under the base graph's launch
        arr_m, bufs_m, epi_m = self._spmm_local(modal, 0)
        self._exchange_begin(bufs_m)
        arr, bufs_b, _ = self._spmm_local([base], M)
        self._exchange_end()
        if epi_m:
            ops.spmm_epilogue_rows(arr_m, M, self.D)
        for t in bufs_b:
            self._ar(t)
        ops.spmm_epilogue_rows(arr, 1, self.D)
        if refs is not None and with_cl:
            refs.append((arr, 0, 0))
        ops.gcn_mix_fwd(self.Bx, self.Z, self.P.view("modal_weight"), hp["modal_adj_weight"], self.Mx, **mo)
        s = 1.0 + hp["residual_weight"]
        batch = [dict(adj=A, x_lo=xMx, y=self.Fin, alpha=s, c1=self.Mx, beta=s)]
        if with_cl:
            batch.append(dict(adj=A, x_lo=xX1, y_pert=self.X2p, noise_degree=nd, **o2, **self._rng(1, noise)))
        arr = self._spmm(batch)
        if refs is not None and with_cl:
            refs.append((arr, 1, 1))
        if with_cl:
            arr = self._spmm([dict(adj=A, x_lo=xX2, y_pert=self.X3p, noise_degree=nd, y_mean=self.CL1, m1=self.X1p, m2=self.X2p, **self._rng(2, noise))])
            if refs is not None:
                refs.append((arr, 0, 2))

    def gcn_output(self) -> Dict[str, Tensor]:
        nu = self.nu
        out = {"u_final_embs": self.Fin[:nu], "i_final_embs": self.Fin[nu:]}
        for m, nm in enumerate(modality_names(self.M)):
            out[f"u_{nm}_embs"], out[f"i_{nm}_embs"] = self.Z[m][:nu], self.Z[m][nu:]
        return out

    # -- one optimiser step ------------------------------------------------------------------------------
    def _views(self):
        """(user-side source, user-side gradient, item-side source, item-side gradient) of every contrasted view, in JointEngine's slot order."""
        nu = self.nu
        v = [(self.Fin, self.dFin)] + [(self.Z[m], self.dZ[m]) for m in range(self.M)] + [(self.CL1, self.dCl1), (self.X1p, self.dCl2)]
        return [(s[:nu], g[:nu], s[nu:], g[nu:]) for s, g in v]

    def _terms(self):
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
        """Install one minibatch (global user / item ids).  Without `bounds` the batch is first sorted by owner (sort_batch_by_owner)."""
        if bounds is None:
            users, pos, neg, bounds = sort_batch_by_owner(self.layout, users, pos, neg)
        B = int(users.shape[0])
        assert B <= self.B_max
        self.users_s[:B].copy_(users.to(torch.int32) - self.layout.u0)  # local ids; only the own segment is ever dereferenced
        self.pos_s[:B].copy_(pos)
        self.neg_s[:B].copy_(neg)
        self.seg_s.copy_(bounds[self.rank:self.rank + 2])
        return B

    def step(self, users: Tensor, pos: Tensor, neg: Tensor, noise: Optional[Sequence[Tensor]] = None, precision: int = 3, update: bool = True,
             bounds: Optional[Tensor] = None) -> Tensor:
        """Forward + backward (+ Adam) of one minibatch of GLOBAL (user, pos, neg) ids -- the same call on every replica.
        Returns loss_parts = [InfoNCE terms..., bpr, reg(own users), reg(items)] (device tensor; no host synchronisation)."""
        B = self.set_batch(users, pos, neg, bounds)
        if noise is None and update:
            return self._step_planned(B, precision)
        return self._step_body(B, noise, precision, update)

    def _step_body(self, B: int, noise, precision: int, update: bool) -> Tensor:
        D, nu, M, hp, I = self.D, self.nu, self.M, self.hp, self.I
        nS, nU, nT = self.n_slots, self.n_uslots, self.n_terms
        cview = lambda t, lead, *rest: t.view(-1)[: lead * B * int(np.prod(rest) if rest else 1)].view(lead, B, *rest)
        zhat, dzhat, znorm = cview(self.zhat, nS, D), cview(self.dzhat, nS, D), cview(self.znorm, nS)
        ub, dub = cview(self.ub, nU, D), cview(self.dub, nU, D)
        lse, row_loss = cview(self.lse, nT), cview(self.row_loss, nT + 1)
        users, pos, neg, ar = self.users_s[:B], self.pos_s[:B], self.neg_s[:B], self.arange[:B]
        x16 = self._x16()
        if x16 is None:
            ops.zero_(self.gblock)
        # (bf16 gather operands, i.e. blocks beyond the caches: the (M + 3) n d scatter targets are NOT cleared by a pass over all rows -- 13.8 GB per
        # step on a replica of the 10M-user configuration.  dFin / dCl1 / dCl2 are zero outside the minibatch's rows and get those rows cleared again
        # once they are consumed; the dZ_m are written, not accumulated into, by the mixing backward, the CL gradients of the modality views are
        # scattered on top afterwards -- JointEngine.step does the same on one GPU)
        ops.zero_(dzhat)
        ops.zero_(ub)
        ops.zero_(dub)
        self.forward(noise=noise, with_cl=True, precision=precision)
        views = self._views()
        # ---- the minibatch: own segment of the user rows -> batch table on every replica ----
        ops.gather_rows(ops.Slots([(v[0], users, None) for v in views]), B, D, self.seg_s, ub)
        self._ar(ub)
        entries = []
        for j, v in enumerate(views):
            entries += [(ub[j], ar, dub[j]), (v[2], pos, v[3])]
        slots = ops.Slots(entries)
        terms = self._terms()
        pk = dict(pack=self.pack, n_slots=nS, precision=precision) if self.mfma_cl else {}
        ops.gather_normalize(slots, B, D, zhat, znorm, self.pack)
        ops.infonce_fwd_bwd(zhat, terms, nT, B, D, lse, row_loss, dzhat, **pk)  # (rows are unit-norm: forward + both gradient passes in one call)
        ops.bpr_fwd_bwd_tables(ub[0], views[0][2], ar, pos, neg, row_loss[nT], dub[0], views[0][3])
        ops.segment_sum(row_loss, nT + 1, B, 1.0, self.loss_parts)
        emb = self.P.view("emb")
        if nu:
            ops.sumsq(emb[:nu], hp["reg"], self.loss_parts[nT + 1:], self.sumsq_scratch)
        else:
            ops.zero_(self.loss_parts[nT + 1:nT + 2])
        ops.sumsq(emb[nu:], hp["reg"], self.loss_parts[nT + 2:], self.sumsq_scratch)
        # ---- backward ----
        A = self.adj
        s = 1.0 + hp["residual_weight"]
        modal_v = range(1, 1 + M)  # the modality views (their gradients go into the dZ_m, i.e. behind the mixing backward in bf16-operand mode)
        if x16 is None:
            ops.scatter_normalize_bwd(slots, B, D, zhat, znorm, dzhat)
            ops.scatter_add_rows(ops.Slots([(v[0], users, v[1]) for v in views]), B, D, self.seg_s, dub)
            gFin, gCl1, gX2p, gBx, gZ, o16, mo = self.dFin, self.dCl1, self.dX2p, self.dBx, self.dZ, {}, {}
        else:
            # the user-side slots scatter into the batch table dub (all of them now); the item-side slots of the modality views wait
            ops.scatter_normalize_bwd(ops.Slots([(e_[0], e_[1], None if (i_ % 2 == 1 and i_ // 2 in modal_v) else e_[2]) for i_, e_ in enumerate(slots.entries)]),
                                      B, D, zhat, znorm, dzhat)
            ops.scatter_add_rows(ops.Slots([(v[0], users, None if j in modal_v else v[1]) for j, v in enumerate(views)]), B, D, self.seg_s, dub)
            # dFin and dCl1 are zero outside the minibatch's rows (own users of the segment, U + pos[b], U + neg[b]): their bf16 gather copies get
            # just those rows; the mixing backward writes the copies of its dense outputs itself
            g16 = x16["g"]
            ops.cast_batch_rows_bf16([self.gblock[0], self.gblock[1]], [g16[0], g16[1]], users, pos, neg, nu, seg=self.seg_s)
            gFin, gCl1, gX2p, gBx, gZ = g16[0], g16[1], x16["dX2p"], x16["dBx"], [g16[3 + m] for m in range(M)]
            o16, mo = dict(y16=x16["dX2p"]), dict(dbx16=x16["dBx"], dz16=gZ, dz_overwrite=True)
        self._spmm([dict(adj=A, x_lo=gFin, y=self.dMx, alpha=s, c1=self.dFin, beta=s),
                    dict(adj=A, x_lo=gCl1, y=self.dX2p, alpha=1.0 / 3.0, c1=self.dCl1, beta=1.0 / 3.0, **o16)])
        self._spmm([dict(adj=A, x_lo=gX2p, y=self.dX1p, alpha=1.0, c1=self.dCl1, beta=1.0 / 3.0, c2=self.dCl2, gamma=1.0)])
        i0, i1 = self.layout.item_slice()
        if x16 is not None:  # dFin, dCl1, dCl2 and the two bf16 copies are consumed: their minibatch rows are cleared (they stay zero for the next step)
            ops.zero_batch_rows([self.gblock[0], self.gblock[1], self.gblock[2], x16["g"][0], x16["g"][1]], users, pos, neg, nu, seg=self.seg_s)
        ops.gcn_mix_bwd(self.dMx, self.Bx, self.Z, self.dZ, self.dX1p, self.P.view("modal_weight"), hp["modal_adj_weight"], dbx=self.dBx,
                        dw=self.P.g("modal_weight"), partials=self.mix_partials, count=(nu * D, (nu + i0) * D, (nu + i1) * D), **mo)
        if x16 is not None:  # the CL gradients of the modality views on top of the written dZ_m, then the bf16 copies of the rows they touched
            ops.scatter_normalize_bwd(ops.Slots([(e_[0], e_[1], e_[2] if (i_ % 2 == 1 and i_ // 2 in modal_v) else None) for i_, e_ in enumerate(slots.entries)]),
                                      B, D, zhat, znorm, dzhat)
            ops.scatter_add_rows(ops.Slots([(v[0], users, v[1] if j in modal_v else None) for j, v in enumerate(views)]), B, D, self.seg_s, dub)
            ops.cast_batch_rows_bf16(self.dZ, gZ, users, pos, neg, nu, seg=self.seg_s)
        # last backward SpMMs.  Gradient exchange overlapped with the remaining backward work: the modal graphs' item partials (they feed
        # the projection gradients) travel under the base graph's SpMM; the base graph's item partials (= d i_embs) travel under the
        # projection backward.
        arr_m, bufs_m, epi_m = self._spmm_local([dict(adj=self.modal_adj[m], x_lo=gZ[m], y=self.T[m]) for m in range(M)], 0)
        self._exchange_begin(bufs_m)
        arr_b, bufs_b, epi_b = self._spmm_local([dict(adj=A, x_lo=gBx, y=self.P.g("emb"))], M)
        self._exchange_end()
        if epi_m:
            ops.spmm_epilogue_rows(arr_m, M, D)
        self._exchange_begin(bufs_b)
        ne = self.n * D
        names = modality_names(M)
        mw = self.P.offsets["modal_weight"]
        ops.zero_(self.P.grad[ne:mw])  # projection weights (accumulated by K slices) and biases; d(modal_weight) behind them was just written
        if i1 > i0:  # projection backward: this replica's slice of the items, the modalities' weight gradients in one launch
            ops.rownorm_bwd_batched([dict(dy=self.T[m][nu + i0:nu + i1], y=self.nf[m][i0:i1], norm=self.fnorm[m][i0:i1], dx=self.dfeat[m][i0:i1],
                                          csum=self.P.g(f"{nm}_layer.bias")) for m, nm in enumerate(names)])  # (+ the biases' gradients, region zeroed above)
            if self._feat_kernels(precision):
                FT = self._features_T_slice()
                ops.item_project_batched([dict(X=FT[m], F=self.dfeat[m][i0:i1], out=self.P.g(f"{nm}_layer.weight"), K=i1 - i0) for m, nm in enumerate(names)],
                                         D, out_transposed=True)
            else:
                ops.gemm_batched([dict(A=self.dfeat[m][i0:i1], B=self.feats[m][i0:i1], out=self.P.g(f"{nm}_layer.weight"), trans_a=True, precision=precision,
                                       split_k=128) for m, nm in enumerate(names)])
        self._exchange_end()
        if epi_b:
            ops.spmm_epilogue_rows(arr_b, 1, D)
        self._ar(self.P.grad[ne:])  # projections + modal_weight: partial sums over disjoint node sets
        if update:
            self.apply_adam()
        self.rng_step += 1
        return self.loss_parts

    def _step_planned(self, B: int, precision: int) -> Tensor:
        """The step as a recorded launch plan (kernel launches and collectives in order), replayed with fresh Philox offsets and Adam
        (lr, step).  The batch lives in the static index buffers set_batch() fills."""
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
            self.rng_step -= 1  # recording ran the bookkeeping once without launching anything
            self.P.step -= 1
            self._plans[key] = plan
        x16 = self._x16()
        if x16 is not None and self._emb16_step != self.P.step:
            self._emb16_refresh(x16)
        lanes = self.D // 4
        base = self.rng_step * 3 * self.N * lanes
        for arr, i, k in plan["philox"]:
            arr[i].offset = base + k * self.N * lanes
        self.P.step += 1
        for c in plan["adam"]:
            c[1][9], c[1][13] = self.lr, self.P.step
        L.replay(plan["calls"])
        if x16 is not None:
            self._emb16_step = self.P.step
        self.rng_step += 1
        return self.loss_parts

    def apply_adam(self):
        P, nu, D = self.P, self.nu, self.D
        P.step += 1
        ne = self.n * D
        x16 = self._x16()
        ops.adam_step(P.data, P.grad, P.m, P.v, self.lr, P.step, extra=list(self.T), extra_n=[nu * D] * self.M, l2_coef=2.0 * self.hp["reg"], l2_n=ne,
                      p16=x16["emb"].view(-1) if x16 is not None else None)
        if x16 is not None:
            self._emb16_step = P.step

    def full_grads(self) -> Dict[str, Tensor]:
        """Gradients as autograd reports them on the unsharded model (parity tests): u_embs = the replica's OWN rows."""
        nu = self.nu
        g = {k: self.P.g(k).clone() for k in self.P.names}
        emb = g.pop("emb") + 2.0 * self.hp["reg"] * self.P.view("emb")
        for t in self.T:
            emb[:nu] += t[:nu]
        g["u_embs"], g["i_embs"] = emb[:nu], emb[nu:]
        return g

    def losses(self) -> Dict[str, float]:
        """Loss values of the last step (global: the partial user-side L2 term is summed over the replicas here -- a host sync)."""
        lp = self.loss_parts.double().clone()
        nT = self.n_terms
        if comm.active(self.world):
            _all_reduce(lp[nT + 1:nT + 2], self.group)
        lp = lp.cpu()
        cl, bpr, reg = float(lp[:nT].sum()), float(lp[nT]), float(lp[nT + 1] + lp[nT + 2])
        return {"loss": bpr + reg + cl, "bpr": bpr, "reg": reg, "cl": cl}


def make_sharded_adjacency(layout: ShardLayout, trn_rows: np.ndarray, trn_cols: np.ndarray, D: int, device, task_len: int = graph.TASK_LEN,
                           vec: int = 4) -> graph.Adjacency:
    """The replica's local block of the normalised bipartite adjacency of (trn_rows, trn_cols): host build (base graph, once per run)."""
    rowptr, col, val = shard.local_block_csr(trn_rows, trn_cols, layout.users, layout.items, layout.u0, layout.u1, layout.rank, layout.world)
    return graph.adjacency_from_csr(rowptr, col, val, D, device, task_len, vec=vec)

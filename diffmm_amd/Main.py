"""Training driver with the reference's surface (Main.py:18-487): Coach(handler, config).run() /.prepareModel() /
.trainEpoch() / .testEpoch() / .calcRes() / .makePrint() / .save_max(), seed_it(seed), CLI `-c/--config`.

The three phases of an epoch (diffusion training, reverse-chain rebuild of the modality graphs, joint BPR +
contrastive training) run on diffmm_amd.engine (HIP kernels only); with several GPUs on diffmm_amd.sharded (users sharded over
the replicas, one process per GPU).  Host synchronisation points per epoch: one per diffusion PHASE in bf16 mode (the steps leave
their loss means on the device; the loss normalisation of Main.py:175-176 reaches the backward kernels as a device scalar -- the ~fp32
parity mode reads the means back per iteration, as the reference does), one per rebuilt graph set (which also reads the reverse chain's
error word, before the graphs are used), one at the end of the joint phase for the logged losses.
"""
from __future__ import annotations

import argparse
import math
import os
import random
from typing import Dict, List, Optional

import numpy as np
import torch

from . import comm, graph, ops
from .Conf import Config, load_config
from .DataHandler import DataHandler
from .Model import Denoise, GaussianDiffusion, Model, denoise_dims, hyper_dict
from .engine import DiffusionEngine, modality_names

main_log = None  # set by the CLI; any object with .info(str)


def _log(msg: str) -> None:
    if main_log is not None:
        main_log.info(msg)


class Coach:
    def __init__(self, handler: DataHandler, config: Config, precision: int = 1, rank: int = 0, world: int = 1, group=None, layout=None,
                 emulate_replica: bool = False):
        """precision: 1 = single bf16 MFMA product in the dense GEMMs (fast path), 3 = hi/lo split (~fp32, parity tests).
        world > 1: one process per GPU (torch.distributed initialised by the caller; backend 'nccl' == RCCL); the users are sharded over
        the replicas (diffmm_amd/sharded.py), every epoch makes exactly the reference's optimiser steps on the reference's minibatches.
        layout: a sharded.ShardLayout to use instead of the edge-balanced one computed from the handler's graph (the replica emulation's graph holds only the
        own users' interactions).  emulate_replica (tools/scale_epoch.py): ONE process runs replica `rank` of `world` with every collective
        reduced to its local part (no process group is needed) -- the per-GPU work and memory of a job this box has no GPUs for."""
        self.handler, self.config, self.precision = handler, config, precision
        self.device = torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")
        _log(f"USER: {config.data.user_num}, ITEM: {config.data.item_num}")
        _log(f"NUM OF INTERACTIONS: {len(handler.trainData)}")
        self.epoch_idx = 0
        self.world, self.rank, self.group = int(world), int(rank), group
        self.layout = layout
        if emulate_replica:
            self.group = comm.LOCAL
        if comm.active(self.world) and layout is None:
            from .sharded import ShardLayout
            self.layout = ShardLayout.make(handler.graph.trn_rows, int(config.data.user_num), int(config.data.item_num), self.world, self.rank)
        self.phase_seconds = [0.0, 0.0, 0.0]
        # bounded samples of an epoch (bench.py --workload scale, tests/test_configs_gpu.py; None = the reference's epoch): the reverse chain + top-k
        # run for the first `rebuild_limit` (own) users only, the others keep their observed interactions as their rebuilt edges
        self.rebuild_limit: Optional[int] = None
        self.best = dict(Recall=0.0, NDCG=0.0, Precision=0.0, epoch=0)
        self.his_max = [0.0, 0.0, 0.0]

    def makePrint(self, name, epoch, results: dict) -> str:
        body = ", ".join(f"{k}={v:.5f}" for k, v in results.items())
        return f"Epoch {epoch}/{self.config.train.epoch}, {name}: {body}  "

    def save_max(self, new: list, old: list) -> list:
        return [n if n > o else o for n, o in zip(new, old)]

    # ---------------------------------------------------------------------------------------------
    def prepareModel(self) -> None:
        cfg, h = self.config, self.handler
        M = h.modalities()
        feats = [h.image_feats, h.text_feats] + ([h.audio_feats] if M == 3 else [])
        self.M = M
        lay = self.layout
        self.model = Model(cfg, *feats, precision=self.precision, layout=lay, group=self.group)
        self.diffusion_model = GaussianDiffusion(cfg)
        in_dims, out_dims = denoise_dims(cfg)
        own_users = lay.nu if lay is not None else int(cfg.data.user_num)
        self.item_shard = self._denoise_item_shard(int(in_dims[1]), [int(x) for x in out_dims[:-1]])
        self.diff_engine = DiffusionEngine(int(cfg.data.item_num), int(cfg.base.latdim), int(in_dims[1]), int(cfg.base.d_emb_size), M,
                                           int(cfg.hyper.steps), hyper_dict(cfg), self.device, int(cfg.train.batch),
                                           chain_rows=min(max(own_users, 1), 16384), hidden=[int(x) for x in out_dims[:-1]], item_shard=self.item_shard)
        dens = [Denoise(in_dims, out_dims, cfg, _engine=self.diff_engine, _slot=m, precision=self.precision) for m in range(M)]
        self.image_denoise_model, self.text_denoise_model = dens[0], dens[1]
        if M == 3:
            self.audio_denoise_model = dens[2]
        self.base_lr = float(cfg.train.lr)
        self.image_adj = self.text_adj = self.audio_adj = None
        self.base_adj = h.torchBiAdj
        if lay is not None:
            from .sharded import make_sharded_adjacency
            DE, dd = self.diff_engine, h.diffusionData
            DE.hp["seed"] = int(DE.hp.get("seed", 1818)) + 7919 * self.rank  # the replicas' rows / item slices draw from different Philox streams
            DE.t_gen = torch.Generator(device=self.device)
            if self.item_shard is None:  # row split: every replica trains the Denoise on the minibatch rows of its own users
                DE.shard_world, DE.shard_rank, DE.shard_group, DE.shard_active = self.world, self.rank, self.group, True
                DE.t_gen.manual_seed(int(cfg.base.seed) + 101 * (self.rank + 1))
            else:  # item split: every replica sees all rows of the minibatch (over its item slice), so the rows' timesteps must agree everywhere
                DE.t_gen.manual_seed(int(cfg.base.seed) + 101)
            vec = graph.spmm_operand_vec(lay.n_local, int(cfg.base.latdim))  # by the size of the replica's own operand [own users ; items]
            self.base_adj = make_sharded_adjacency(lay, h.graph.trn_rows, h.graph.trn_cols, int(cfg.base.latdim), self.device,
                                                   task_len=graph.spmm_task_len(vec), vec=vec)
            self.local_ptr = (dd.ptr[lay.u0:lay.u1 + 1] - dd.ptr[lay.u0]).contiguous()  # CSR of the own users' train items
            self.local_ptr32 = self.local_ptr.to(torch.int32)

    def _denoise_item_shard(self, H: int, hidden: List[int]):
        """How the Denoise MLPs are split over the replicas of a user-sharded run.  "rows": weights replicated, every replica trains on
        the minibatch rows of its own users, gradients all-reduced (2 I H fp32 per modality and step).  "items" (SURVEY 8e): replica g holds the W1
        columns / W2 rows of its item slice -- weights, gradients and Adam state are 1 / world each and the step exchanges [B, H]-sized partial
        activations instead of weight gradients.  DMM_DENOISE_SHARD = rows | items | auto (default: items once a modality's weight gradient
        reaches 256 MiB -- I >= 32768 at H = 1024 -- and the step's kernels cover the shape)."""
        from .engine import ItemShard
        if not comm.active(self.world):
            return None
        cfg = self.config
        mode = os.environ.get("DMM_DENOISE_SHARD", "auto")
        I, D = int(cfg.data.item_num), int(cfg.base.latdim)
        latent = os.environ.get("DMM_CHAIN_MODE", "latent") == "latent"  # (the item-sharded rebuild exists for the latent chain only: engine.rebuild_edges)
        fits = (len(hidden) == 1 and D in (64, 128, 256) and self.precision == 1 and int(cfg.hyper.sampling_step) == 0 and latent
                and (self.world - 1) * -(-I // self.world) < I)  # (the last replica's slice is not empty)
        if mode == "items" and not fits:
            raise ValueError("DMM_DENOISE_SHARD=items needs one hidden width, d in 64 / 128 / 256, bf16 mode, sampling_step = 0, the latent chain and a "
                             "non-empty item slice per replica")
        if mode == "items" or (mode == "auto" and fits and 2 * I * H * 4 >= (256 << 20)):
            return ItemShard(self.rank, self.world, I, group=self.group)
        return None

    def _set_lr(self) -> None:
        """CosineAnnealingLR(T_max=train.epoch, eta_min=1e-4), stepped once per epoch (Main.py:59-65,93,101,105,110)."""
        lr = self.base_lr
        if self.config.train.use_lr_scheduler:
            lr = 1e-4 + (self.base_lr - 1e-4) * (1.0 + math.cos(math.pi * self.epoch_idx / self.config.train.epoch)) / 2.0
        self.model.engine.lr = lr
        self.diff_engine.lr = lr

    def run(self, resume: Optional[str] = None, save: Optional[str] = None) -> None:
        self.prepareModel()
        _log("Model Initialized")
        if resume:
            self.load_checkpoint(resume)
            _log(f"Resumed from {resume} at epoch {self.epoch_idx}")
        try:
            for epoch in range(self.epoch_idx, int(self.config.train.epoch)):
                result = self.trainEpoch()
                _log(self.makePrint("Train", epoch, result))
                if epoch % int(self.config.train.tstEpoch) == 0:
                    result = self.testEpoch()
                    self.his_max = self.save_max([result["Recall"], result["NDCG"], result["Precision"]], self.his_max)
                    if result["Recall"] > self.best["Recall"]:
                        self.best = dict(result, epoch=epoch)
                    _log(self.makePrint("Test", epoch, result))
                if save:
                    self.save_checkpoint(save)
                _log("Current best: " + self._best_line())
            _log("Best epoch: " + self._best_line(prefix=""))  # the reference's closing line (Main.py:80)
        except KeyboardInterrupt:
            _log("Training interrupted by user!")
            _log("Current best: " + self._best_line())

    def _best_line(self, prefix: str = "Epoch: ") -> str:
        best, his_max = self.best, self.his_max
        return (f"{prefix}{best['epoch']}, Recall: {best['Recall']:.5f}({his_max[0]:.5f}), NDCG: {best['NDCG']:.5f}({his_max[1]:.5f}), "
                f"Precision: {best['Precision']:.5f}({his_max[2]:.5f})")

    # -- checkpoint / resume (absent in the reference) ----------------------------------------------------------------
    def _flat_models(self):
        return [("model", self.model.engine.P)] + [(f"denoise{m}", p) for m, p in enumerate(self.diff_engine.P)]

    def _ckpt_path(self, path: str) -> str:
        """User-sharded runs write one file per replica (each holds its own user rows and their Adam state)."""
        return path if self.world == 1 else f"{path}.rank{self.rank}of{self.world}"

    def save_checkpoint(self, path: str) -> None:
        """Everything the next epoch depends on: weights and Adam state of the model and of every Denoise, the step / Philox counters,
        the sampler and loader generators, the epoch index, the best metrics so far.  Tensors and plain numbers only (loads with
        weights_only=True)."""
        h, J, DE = self.handler, self.model.engine, self.diff_engine
        sd = {"epoch_idx": int(self.epoch_idx), "rng_step_joint": int(J.rng_step), "rng_step_diffusion": int(DE.rng_step),
              "xT_calls": int(DE._xT_calls), "world": int(self.world),
              "best": {k: float(v) for k, v in self.best.items()}, "his_max": [float(v) for v in self.his_max],
              "gen": {"train_loader": h.trainLoader.gen.get_state(), "diffusion_loader": h.diffusionLoader.gen.get_state(),
                      "neg_sampler": h.trainData.gen.get_state(), "neg_calls": int(h.trainData.neg_calls), "torch_cpu": torch.get_rng_state(),
                      "torch_device": torch.cuda.get_rng_state(self.device)}}  # the timestep draws use the global device generator
        if DE.t_gen is not None:
            sd["gen"]["timesteps"] = DE.t_gen.get_state()
        for name, P in self._flat_models():
            sd[name] = {"data": P.data.clone(), "m": P.m.clone(), "v": P.v.clone(), "step": int(P.step), "layout": P.signature()}
        sd["format"] = {"dmm_version": int(ops.L.lib().dmm_version()), "denoise_shard": "items" if self.item_shard is not None else "rows",
                        "rank": int(self.rank)}
        torch.save(sd, self._ckpt_path(path))

    def load_checkpoint(self, path: str) -> None:
        """Call after prepareModel() on a Coach built from the same config, data and number of replicas."""
        h, J, DE = self.handler, self.model.engine, self.diff_engine
        sd = torch.load(self._ckpt_path(path), map_location=self.device, weights_only=True)
        if int(sd.get("world", 1)) != self.world:
            raise ValueError(f"checkpoint {path} was written by {sd.get('world', 1)} replica(s), this run has {self.world}")
        fmt = sd.get("format", {})
        shard_now = "items" if self.item_shard is not None else "rows"
        if fmt and (fmt.get("denoise_shard", shard_now) != shard_now or int(fmt.get("rank", self.rank)) != self.rank):
            raise ValueError(f"checkpoint {path} was written by replica {fmt.get('rank')} with the Denoise split by {fmt.get('denoise_shard')}; "
                             f"this is replica {self.rank} with the split by {shard_now}")
        for name, P in self._flat_models():
            restore_flat(P, sd[name], f"checkpoint {path}: {name}")
        self.epoch_idx, J.rng_step, DE.rng_step = int(sd["epoch_idx"]), int(sd["rng_step_joint"]), int(sd["rng_step_diffusion"])
        DE._xT_calls = int(sd["xT_calls"])
        if "best" in sd:
            self.best = {k: (int(v) if k == "epoch" else float(v)) for k, v in sd["best"].items()}
            self.his_max = [float(v) for v in sd["his_max"]]
        DE.w16_step = [-1] * DE.M  # bf16 weight copies of the reverse chain are stale
        DE.w16_train_step = [-1] * DE.M  # ... and the training step's
        J._plans.clear()
        J._emb16_step = -1  # ... and so is the embedding table's bf16 gather copy
        self._reg_cache = None  # ... and the cached sum(i_embs^2)
        h.trainLoader.gen.set_state(sd["gen"]["train_loader"].cpu()); h.diffusionLoader.gen.set_state(sd["gen"]["diffusion_loader"].cpu())
        h.trainData.gen.set_state(sd["gen"]["neg_sampler"].cpu())
        h.trainData.neg_calls = int(sd["gen"].get("neg_calls", 0))
        torch.set_rng_state(sd["gen"]["torch_cpu"].cpu()); torch.cuda.set_rng_state(sd["gen"]["torch_device"].cpu(), self.device)
        if DE.t_gen is not None and "timesteps" in sd["gen"]:
            DE.t_gen.set_state(sd["gen"]["timesteps"].cpu())

    def makeTorchAdj(self, u_list, i_list, edge_list=None) -> graph.Adjacency:
        cfg = self.config
        return graph.make_adjacency(np.asarray(u_list), np.asarray(i_list), int(cfg.data.user_num), int(cfg.data.item_num), int(cfg.base.latdim),
                                    self.device, vec=self.base_adj.vec, task_len=graph.spmm_task_len(self.base_adj.vec))

    # ---------------------------------------------------------------------------------------------
    def trainEpoch(self, draws=None, modal_adj_override: Optional[List[graph.Adjacency]] = None) -> Dict[str, float]:
        """One epoch (Main.py:136-388).  `draws` (tests only, single replica) supplies the random draws in the reference's call order."""
        cfg, h, M, dev = self.config, self.handler, self.M, self.device
        J, DE, prec, lay = self.model.engine, self.diff_engine, self.precision, self.layout
        if draws is not None and (lay is not None or self.item_shard is not None):
            raise ValueError("recorded draws drive the single-replica path only")
        self._set_lr()
        if draws is None:
            h.trainData.negSampling()  # own generator, identically seeded on every replica: the same negatives everywhere
        B = int(cfg.train.batch)
        train_steps = max(1, len(h.trainData) // B)
        diffusion_steps = max(1, len(h.diffusionData) // B)
        T = int(cfg.hyper.steps)
        dd = h.diffusionData

        import time
        # phase boundaries as events on the stream (the host no longer waits at them: the diffusion phase's results are read with the epoch's last
        # transfer), turned into phase_seconds after that transfer
        cuda = dev.type == "cuda"
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if cuda else None
        mark = (lambda i: ev[i].record()) if cuda else (lambda i: None)
        t_start = time.perf_counter()
        mark(0)
        # ---- phase 1: diffusion training (Main.py:145-192) ----
        J._project(prec)  # Linear_m(E_m): constant during this phase (the encoder is not stepped here)
        ish = self.item_shard
        two_d = not getattr(J, "replicated_items", True)  # items sharded as well (sharded2d.py): J holds the features / embeddings of its own item slice
        if two_d:  # (the slices are engine.ItemShard's: an item-sharded Denoise reads exactly the replica's own rows; the row split reads all items)
            assert ish is None or (ish.i0, ish.i1) == (J.i0, J.i1)
            feats = [J.feat[m][:J.ni] for m in range(M)] if ish is not None else J.all_feats()
        else:
            feats = [J.feat[m] if ish is None else J.feat[m][ish.i0:ish.i1] for m in range(M)]
        i_embs = J.i_embs  # (2D: one all-gather of the replicas' slices) -- does not move in this phase either
        cached = getattr(self, "_reg_cache", None)  # (step of the joint model it was computed at, value): read with the previous epoch's results
        reg_term = cached[1] if (cached is not None and cached[0] == J.P.step and draws is None) else DE.reg_term(i_embs)
        dl = [0.0] * M
        step_means = []
        diff_batches = h.diffusionLoader.index_batches()  # index tensors only; rows are densified per batch, for the users this replica owns
        DE.begin_phase()  # feats / i_embs are constant until end_phase(): what the steps derive from them alone is made once
        for idx in diff_batches:
            own = idx if (lay is None or ish is not None) else idx[(idx >= lay.u0) & (idx < lay.u1)]  # (item split: all rows, own item slice)
            x0 = dd.csr_rows(own)  # x_0 stays CSR: the engine densifies it straight into its operand buffer
            t = noise = None
            if draws is not None:
                t, noise = [], []
                for m in range(M):
                    t.append(draws.randint(T, x0.shape[0]).to(dev, torch.int32))
                    noise.append(draws.randn((x0.shape[0], int(cfg.data.item_num))).to(dev))
            means = DE.train_step(x0, feats, i_embs, t=t, noise=noise, precision=prec, reg_term=reg_term, global_rows=int(idx.shape[0]),
                                  host_means=prec != 1)
            step_means.append(means)
        DE.end_phase()
        # (bf16 mode: the steps left their loss means on the device; they are only logged, so they are read with the epoch's other results at the end
        # -- no host round trip between the diffusion phase and the rebuild)
        means_dev = torch.stack(step_means) if (step_means and isinstance(step_means[0], torch.Tensor)) else None
        t_diff = time.perf_counter()
        mark(1)

        # DMM_LOADER_PREFETCH: the joint phase's shuffled epoch (one permutation draw, three gathers: a dozen small launches) made UNDER the rebuild's
        # reverse chain instead of between the rebuild and the first joint step.  1 = on a stream of its own, forked here: measured neutral (round 4,
        # 2 x 20 epochs each way: 17.92 / 18.02 ms per epoch with it, 17.91 / 17.83 without) -- with the rebuild's own side streams the process has
        # more streams than the runtime has hardware queues, and the extra stream's launches queue up behind the chain's.  2 = behind the first
        # modality's top-k on the rebuild's own low-priority stream (below), which runs under the chain's second launch anyway: neutral as well
        # (17.69 / 17.53 against 17.78 / 17.49).  Default 0: the loop below draws the epoch, as before.
        prefetched = None
        prefetch_mode = int(os.environ.get("DMM_LOADER_PREFETCH", "0")) if cuda else 0
        main_s = torch.cuda.current_stream(dev) if cuda else None

        def prefetch_on(stream):
            with torch.cuda.stream(stream):
                if h.trainLoader.prefetch():
                    for t_ in h.trainLoader._prefetched:
                        t_.record_stream(main_s)
                    e_ = torch.cuda.Event()
                    e_.record(stream)
                    return e_
            return None

        if prefetch_mode == 1:
            if getattr(self, "_prep_stream", None) is None:
                self._prep_stream = torch.cuda.Stream(device=dev)
            e0 = torch.cuda.Event(); e0.record(main_s)
            self._prep_stream.wait_event(e0)
            prefetched = prefetch_on(self._prep_stream)

        # ---- phase 2: rebuild the modality graphs (Main.py:195-253) ----
        if modal_adj_override is None:
            sstep = int(cfg.hyper.sampling_step)
            lim = self.rebuild_limit
            if lay is None:
                if lim is None:
                    reb_batches, init = diff_batches, None
                else:
                    reb_batches, init = [torch.arange(min(int(lim), int(cfg.data.user_num)), device=dev)], dd.col32
                edges = DE.rebuild_edges(dd.rows, reb_batches, dd.deg_ptr32, int(dd.ptr[-1]), precision=prec, sampling_step=sstep,
                                         csr=(dd.ptr, dd.col32, 0), init_cols=init, defer_join=True)
                if getattr(self, "graph_builder", None) is None:  # CSR + SpMM work list are built on the device (no D2H of the edges)
                    self.graph_builder = graph.DeviceGraphBuilder(dd.ptr, dd.users, dd.items, int(cfg.base.latdim), vec=self.base_adj.vec,
                                                                  task_len=graph.spmm_task_len(self.base_adj.vec),
                                                                  reuse=os.environ.get("DMM_GRAPH_REUSE", "1") != "0")
            else:  # own users only (local ids index the local CSR; the data rows are looked up by global id)
                if getattr(self, "graph_builder", None) is None:
                    self.graph_builder = graph.ShardedGraphBuilder(self.local_ptr, lay.nu, lay.users, lay.items, int(cfg.base.latdim), self.rank, self.world,
                                                                   group=self.group, task_len=graph.spmm_task_len(self.base_adj.vec), vec=self.base_adj.vec)
                    self._own_local = torch.arange(lay.nu, device=dev)
                own_local = self._own_local if lim is None else self._own_local[:int(lim)]
                init = None if lim is None else dd.col32[int(dd.ptr[lay.u0]):int(dd.ptr[lay.u1])]
                edges = DE.rebuild_edges(lambda loc: dd.rows(loc + lay.u0), [own_local], self.local_ptr32, int(self.local_ptr[-1]), precision=prec,
                                         sampling_step=sstep, csr=(dd.ptr, dd.col32, lay.u0), init_cols=init)
            ready = getattr(DE, "edges_ready", None) if lay is None else None
            if prefetch_mode == 2 and ready is not None and ready[0] is not None and getattr(DE, "_tail_streams", None):
                prefetched = prefetch_on(DE._tail_streams[0])  # (in stream order behind modality 0's last GEMM + top-k, i.e. under the chain's second launch)
            adjs = self.graph_builder.build_all(edges, **({"ready": ready} if ready is not None else {}))  # (reads the graphs' sizes back: the host is synchronised here)
            DE.edges_ready = None
            DE.check_chain_error()  # a timed-out group barrier of dmm_latent_chain invalidates these graphs: raise BEFORE any Adam step uses them
        else:
            adjs = list(modal_adj_override)
        self.image_adj, self.text_adj = adjs[0], adjs[1]
        self.audio_adj = adjs[2] if M == 3 else None
        J.set_adjacency(self.base_adj, adjs)
        t_reb = time.perf_counter()
        mark(2)

        # ---- phase 3: joint training (Main.py:292-377) ----
        acc = torch.zeros(J.loss_parts.numel(), dtype=torch.float64, device=dev)
        in_kernel_acc = isinstance(getattr(J, "loss_acc", None), torch.Tensor) and os.environ.get("DMM_LOSS_ACC", "1") != "0"  # (the single-GPU engine: its loss reduction adds to fp64 sums itself)
        if in_kernel_acc:
            J.loss_acc.zero_()
        if prefetched is not None:
            torch.cuda.current_stream(dev).wait_event(prefetched)
        for users, pos, neg in h.trainLoader:  # the same shuffled minibatches on every replica (identically seeded loader generator)
            noise = [draws.rand((J.N, J.D)).to(dev) for _ in range(3)] if draws is not None else None
            lp = J.step(users, pos, neg, noise=noise, precision=prec)
            if not in_kernel_acc:
                acc += lp
        if in_kernel_acc:
            acc = J.loss_acc.clone()
        nT = J.n_terms
        if lay is not None:  # [terms..., bpr, reg over the own users, reg over the items]: the user part is a partial sum -- the item part too when
            part = acc[nT + 1:nT + 3] if not getattr(J, "replicated_items", True) else acc[nT + 1:nT + 2]  # the items are sharded as well
            comm.all_reduce(part, group=self.group)
            acc = torch.cat([acc[:nT + 1], acc[nT + 1:nT + 2] + acc[nT + 2:nT + 3]])
        # ONE transfer for everything the host wants of this epoch: the joint phase's loss sums, the diffusion phase's deferred loss means and
        # sum(i_embs^2), which the NEXT epoch's diffusion phase needs as a host value (i_embs does not move until then)
        parts = [acc]
        if means_dev is not None:
            parts.append(means_dev.reshape(-1).double())
        reg_next = getattr(J, "replicated_items", True) and draws is None
        if reg_next:
            parts.append(DE.reg_term_launch(J.i_embs).double())
        mark(3)
        host = torch.cat(parts).cpu().numpy()
        a = host[:acc.numel()]
        if means_dev is not None:
            step_means = host[acc.numel():acc.numel() + means_dev.numel()].reshape(-1, M).tolist()
        if reg_next:
            reg = float(DE.hp["reg"])
            self._reg_cache = (J.P.step, reg * reg * float(host[-1]))
        for means in step_means:  # logging quirk of Main.py:155,177-179: add, then divide the running sums by this batch's total
            tot = sum(means)
            for m in range(M):
                dl[m] += means[m]
            for m in range(M):
                dl[m] /= tot
        if lay is not None:
            self.check_replicas()
        if cuda:  # GPU time between the boundaries (the stream is drained: the transfer above waited for it)
            self.phase_seconds = [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(3)]
        else:
            self.phase_seconds = [t_diff - t_start, t_reb - t_diff, time.perf_counter() - t_reb]
        cl, bpr, reg = float(a[:nT].sum()), float(a[nT]), float(a[nT + 1])
        result = {"Loss": (bpr + reg + cl) / train_steps, "BPR Loss": bpr / train_steps, "reg loss": reg / train_steps, "CL loss": cl / train_steps}
        for m, n in enumerate(modality_names(M)):
            result[f"{n} loss"] = dl[m] / diffusion_steps
        self.epoch_idx += 1
        return result

    def _replicated_parts(self):
        J, DE = self.model.engine, self.diff_engine
        lo = getattr(J, "replicated_lo", J.nu * J.D)  # (items sharded as well: only the projections and modal_weight are replicated)
        nr = DE.n_replicated if self.item_shard is not None else None  # item-sharded Denoise: only the small tensors are replicated
        return [J.P.data[lo:], J.P.m[lo:], J.P.v[lo:]] + [t[:nr] for p in DE.P for t in (p.data, p.m, p.v)]

    def _replica_checksums_differ(self, parts) -> List[int]:
        sums = torch.stack([t.view(torch.int32).sum(dtype=torch.int64) for t in parts])
        mn, mx = sums.clone(), sums.clone()
        comm.all_reduce(mn, op="MIN", group=self.group)
        comm.all_reduce(mx, op="MAX", group=self.group)
        return (mn != mx).nonzero().flatten().tolist()  # (the same list on every replica: MIN / MAX are all-reduced)

    def check_replicas(self) -> None:
        """User-sharded runs: the replicated parameters (item embeddings, projections, modal_weight, every Denoise) and their Adam state
        must be bit-identical on all replicas -- they only ever receive all-reduced gradients (sharded.py).  One pass over them per epoch:
        a 64-bit sum of their bit patterns, MIN and MAX over the replicas must agree.  A first mismatch HEALS the run instead of ending it:
        every replica takes replica 0's copy of the replicated slices and of their Adam state (one broadcast each), the bf16 operand copies
        derived from them are marked stale, and the event is logged and counted (`replica_heals`); a mismatch at the very next check as well
        means the replicas diverge systematically -- that raises, on every replica."""
        parts = self._replicated_parts()
        bad = self._replica_checksums_differ(parts)
        if not bad:
            self._healed_last_check = False
            return
        what = (f"replicated state differs between the replicas (checksum groups {bad}; 0-2 = joint model data/m/v, then data/m/v per Denoise) "
                f"at epoch {self.epoch_idx}")
        if getattr(self, "_healed_last_check", False):
            raise RuntimeError(f"user-sharded run: {what} -- again, right after a re-broadcast from replica 0")
        for t in parts:
            comm.broadcast(t, src=0, group=self.group)
        J, DE = self.model.engine, self.diff_engine
        DE.w16_step = [-1] * DE.M; DE.w16_train_step = [-1] * DE.M  # bf16 copies of the weights are stale on the replicas that were overwritten
        J._emb16_step = -1
        self._healed_last_check = True
        self.replica_heals = getattr(self, "replica_heals", 0) + 1
        msg = f"user-sharded run: {what}; every replica took replica 0's copy (heal #{self.replica_heals})"
        _log(msg)
        if main_log is None and self.rank == 0:
            print(msg, flush=True)
        if self._replica_checksums_differ(parts):
            raise RuntimeError("user-sharded run: replicated state still differs after the re-broadcast from replica 0")

    # ---------------------------------------------------------------------------------------------
    def testEpoch(self) -> Dict[str, float]:
        """Full-ranking evaluation (Main.py:390-420): propagate once, then per user batch score = U_b I^T (dmm_gemm, split-bf16
        ~fp32), train items masked to -1e8 and top-k selected in one launch (dmm_eval_topk); only the [batch, topk] ids leave the GPU.
        User-sharded: every replica ranks the test users it owns; the metric sums are all-reduced."""
        h, J, lay = self.handler, self.model.engine, self.layout
        adjs = [self.image_adj, self.text_adj] + ([self.audio_adj] if self.M == 3 else [])
        J.set_adjacency(self.base_adj, adjs)
        J.forward(with_cl=False, precision=self.precision)
        nu = J.nu if lay is not None else J.U
        u0 = lay.u0 if lay is not None else 0
        uf, itf = J.Fin[:nu], (J.Fin[nu:] if getattr(J, "replicated_items", True) else J.final_item_embs())
        td, k = h.testData, int(self.config.base.topk)
        # the reference ranks test_batch users at a time (Main.py:396-411); the metrics are per-user sums, so the batch size only sets how many launches
        # the evaluation takes: up to 8192 users per launch here (score buffer <= 2 GiB) -- 37 host-paced rounds -> 2 at TikTok shape
        tb = max(int(self.config.train.test_batch), min(8192, max(1, (1 << 29) // max(1, J.I))))
        users_all, u32_all = td._users_d, td._users_i32
        if lay is not None:
            keep = (users_all >= lay.u0) & (users_all < lay.u1)
            users_all, u32_all = users_all[keep], u32_all[keep]
        n_own = int(users_all.shape[0])
        total = torch.zeros(3, dtype=torch.float64, device=uf.device)
        if n_own:
            score = torch.empty(min(tb, n_own), J.I, dtype=torch.float32, device=uf.device)
        for lo in range(0, n_own, tb):
            usr, u32 = users_all[lo:lo + tb], u32_all[lo:lo + tb]
            ub = uf.index_select(0, usr - u0)
            sc = score[:usr.shape[0]]
            ops.gemm(ub, itf, out=sc, trans_b=True, precision=3)
            top = ops.eval_topk(sc, u32, td._trn_ptr, td._trn_col, k)
            total += ops.eval_metrics(top, u32, td._tst_ptr, td._tst_col).sum(0)  # calcRes on the device; one host read at the end
        if lay is not None:
            comm.all_reduce(total, group=self.group)
        rec, ndcg, prec = (total / len(td)).tolist()
        return {"Recall": rec, "NDCG": ndcg, "Precision": prec}

    def calcRes(self, top_idxs: np.ndarray, test_u_its: list, users):
        """Sums of Recall / NDCG / Precision over the batch's users (Main.py:422-448)."""
        k = int(self.config.base.topk)
        disc = 1.0 / np.log2(np.arange(k) + 2.0)
        ideal = np.cumsum(disc)
        rec = ndcg = prec = 0.0
        assert top_idxs.shape[0] == len(users)
        for row, u in zip(top_idxs, users):
            truth = np.asarray(test_u_its[int(u)])
            hit = np.isin(row, truth)
            n_hit = int(hit.sum())
            rec += n_hit / len(truth)
            ndcg += float(disc[hit].sum()) / ideal[min(len(truth), k) - 1]
            prec += n_hit / k
        return rec, ndcg, prec


def restore_flat(P, t: dict, what: str) -> None:
    """Read a checkpointed flat buffer set (data / m / v + the layout it was written in) into P.  The flat layout is an implementation detail
    that has changed between versions (the Denoise's small tensors moved in front of b2 / W1 / W2): a checkpoint whose ordered (name, shape,
    offset) list equals this build's is copied as it is; one with the same tensors at other offsets is REMAPPED tensor by tensor; one
    without a layout record (written before the record existed) or with other names / shapes is refused -- never read positionally."""
    layout = t.get("layout")
    if layout is None:
        raise ValueError(f"{what}: the file carries no layout record (written by a build before the flat parameter layout was versioned); "
                         "its flat buffers cannot be mapped onto this build's layout safely")
    layout = [[str(k), [int(x) for x in shp], int(off)] for k, shp, off in layout]
    mine = P.signature()
    if layout == mine:
        if t["data"].numel() != P.data.numel():
            raise ValueError(f"{what}: {t['data'].numel()} parameters in the file, {P.data.numel()} in this model")
        P.data.copy_(t["data"]); P.m.copy_(t["m"]); P.v.copy_(t["v"])
    else:
        theirs = {k: (shp, off) for k, shp, off in layout}
        if set(theirs) != set(P.names) or any(tuple(theirs[k][0]) != tuple(P.shapes[k]) for k in P.names):
            diff = sorted(set(theirs) ^ set(P.names)) or [k for k in P.names if tuple(theirs[k][0]) != tuple(P.shapes[k])]
            raise ValueError(f"{what}: the file's tensors do not match this model's ({diff[:4]} ...)")
        for k in P.names:
            n, off = int(np.prod(P.shapes[k])), theirs[k][1]
            for dst, src in ((P.data, t["data"]), (P.m, t["m"]), (P.v, t["v"])):
                P.view(k, dst).copy_(src[off:off + n].view(P.shapes[k]))
    P.step = int(t["step"])


def seed_it(seed: int) -> None:
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None) -> None:
    global main_log
    ap = argparse.ArgumentParser(description="Model Configs")
    ap.add_argument("--config", "-c", default="conf/tiktok.toml", type=str, help="config file path")
    ap.add_argument("--precision", type=int, default=1, choices=(1, 3), help="1: bf16 MFMA GEMMs, 3: hi/lo split (~fp32)")
    ap.add_argument("--save", default=None, help="write a checkpoint to this path after every epoch")
    ap.add_argument("--resume", default=None, help="continue from a checkpoint written by --save")
    ap.add_argument("--tolerant", action="store_true", help="drop TOML keys Conf.py does not know (one logged line) instead of failing like the "
                    "reference does: its own conf/baby.toml and conf/test.toml need this")
    args = ap.parse_args(argv)
    try:
        config = load_config(args.config, strict=not args.tolerant)
        print(f"Load configuration ({config.data.name}) file successfully")
    except Exception as e:  # same behaviour as the reference: report and exit 1
        print(f"Error loading configuration file: {e}")
        raise SystemExit(1)
    seed_it(config.base.seed)
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if rank == 0:  # one log file and one console copy per run, whatever the number of replicas
        from .Utils.Log import Log
        main_log = Log("main", config.data.name)
        main_log.info("Start")
        for section, options in vars(config).items():
            main_log.info(f"{section}: {options}")
    if world > 1:  # python -m torch.distributed.run --nproc-per-node N -m diffmm_amd.Main -c ...: one process per GPU, users sharded
        import torch.distributed as dist
        config.base.gpu = str(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(int(config.base.gpu))
        dist.init_process_group(os.environ.get("DMM_BACKEND", "nccl"))
    handler = DataHandler(config)
    _log("Load Data")
    handler.LoadData(sharded=world > 1)
    Coach(handler, config, precision=args.precision, rank=rank, world=world).run(resume=args.resume, save=args.save)


if __name__ == "__main__":
    main()

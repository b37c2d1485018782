#!/usr/bin/env python3
"""Headline benchmark: BPR + diffusion training iterations per second on the TikTok-shape graph (d=64, T=40).

A "step" is one EPOCH of the hot path on synthetic inputs resident in HBM: n_diff diffusion-training iterations
(all modalities), the reverse-chain rebuild of the modality graphs (T Denoise forwards per user batch and modality,
per-user top-k, adjacency build) and n_joint joint BPR + contrastive iterations -- the blended rate of SURVEY.md
section 8(d):  value = (n_diff + n_joint) * K / wall_seconds, B-sized minibatches consumed per second, whole job.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts itself: the parent -- before it makes any GPU call -- launches
the line above as a child (one rank per GPU, backend nccl = RCCL), relays rank 0's single JSON line and exits with the children's status.

--workload scale --users U --items I --latdim d --T T [--batch B --scale-mode strong|weak]: the synthetic graph of SURVEY 8(d) at any size (TikTok degree
law, 3 modalities), a step = a BOUNDED SAMPLE of an epoch (--diff-iters diffusion iterations, the reverse chain + top-k over --rebuild-users users
per replica -- the other users keep their observed interactions --, --joint-iters joint iterations; all three named in config.workload).  strong: U
users in total at every N; weak: U users PER GPU (BASELINE config 5 is `--scale-mode weak --users 1250000 --items 1000000 --latdim 256 --T 1000
--batch 8192` at N = 8), `value` then counts shard-iterations (N per iteration of the N-shard job: the per-GPU work of an iteration is fixed).

N > 1: one process per GPU over RCCL, users sharded over the ranks (diffmm_amd/sharded.py): every rank owns a
contiguous user range (its embedding rows, Adam state, adjacency rows), items are replicated; every SpMM is a local launch
+ one all-reduce of the item partials, every minibatch is the reference's minibatch with its rows split by owner, the
reverse chain and the top-k run on the own users only.  The number of optimiser steps per epoch is the reference's at
every N; the data set is fixed, so this is strong scaling.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
# MI355X_MICROARCH.md, "Indexed rows": whole rows gathered by index read at 5.5-5.8 TB/s from a table far beyond the Infinity Cache (each row once)
# and at 7.4-7.9 TB/s from a 151 MB table (uniformly random rows) -- the range the memory system gives row gathers
GATHER_ROOFLINE_GBS = (5700.0, 7900.0)
PMC_CHAIN_JSON = "r4b_pmc_chain_gemm.json"  # committed PMC summary of tools/roofline_target.py (tools/final_profile.sh)


def scale_graph(users: int, items: int, chunk: int = 1_250_000):
    """The synthetic graph of the scale workload, generated one user range at a time (the generator's working set is ~2 KB per user) and identical on
    every rank (per-range draws are keyed on the range's first user id)."""
    from diffmm_amd import synth
    parts = [synth.make_graph(users, items, feats=(128, 768, 128) if lo == 0 else (), seed=1818, heavy_p=0.011, heavy=(300, 520), user_lo=lo,
                              user_hi=min(users, lo + chunk)) for lo in range(0, users, chunk)]
    cat = lambda k: np.concatenate([getattr(p, k) for p in parts])
    return synth.SynthGraph(users, items, cat("trn_rows"), cat("trn_cols"), cat("tst_rows"), cat("tst_cols"), parts[0].feats)


class _Head:
    """The first n minibatches of a loader's epoch (bounded samples of the scale workload)."""

    def __init__(self, loader, n):
        self.loader, self.n = loader, n

    def __iter__(self):
        for i, b in enumerate(self.loader):
            if i >= self.n:
                return
            yield b

    def __len__(self):
        return min(len(self.loader), self.n)


def make_coach(args, rank: int, world: int):
    from diffmm_amd import comm
    from diffmm_amd.Conf import load_config
    from diffmm_amd.DataHandler import DataHandler
    from diffmm_amd.Main import Coach, seed_it
    cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
    cfg.base.gpu = str(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
    cfg.base.latdim, cfg.hyper.steps, cfg.train.batch = args.latdim, args.T, args.batch
    cfg.data.name = args.workload
    seed_it(cfg.base.seed)
    h = DataHandler(cfg)
    if args.workload == "scale":
        total = args.users * world if args.scale_mode == "weak" else args.users
        h.LoadData(graph_data=scale_graph(total, args.items), sharded=comm.active(world))
    else:
        h.LoadData()
    coach = Coach(h, cfg, precision=args.precision, rank=rank, world=world)
    coach.prepareModel()
    if args.workload == "scale":  # a bounded sample of the epoch
        coach.rebuild_limit = args.rebuild_users
        h.trainLoader = _Head(h.trainLoader, args.joint_iters)
        full_idx = h.diffusionLoader.index_batches
        h.diffusionLoader.index_batches = lambda: full_idx()[: args.diff_iters]
        h.diffusionLoader.bounded = args.diff_iters
    return coach, h, cfg


def timed_launches(fn, n: int) -> float:
    """Average duration (seconds) of n back-to-back launches, HIP events on the stream the kernels run on."""
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def roofline(coach, args):
    """Live measurement of the kernels that bound the epoch (HIP events on the launch stream, operands as the epoch has them):
      * the latent reverse-chain step -- the dominant kernel of the epoch: (T - 1) launches per rebuild, all modalities and
        all users per launch; [rows, H] x [H, H] bf16 MFMA GEMM whose epilogue streams the fp32 state z in and out and writes the next
        activation in bf16: 12 bytes per element against 2 H flops -> HBM-bound (58.6 GFLOP = 23 us of MFMA vs 349 MB = 44 us of HBM);
      * the item-sized GEMM that remains per chain (last step, h_0 W2^T -> fp32 view for the top-k; MFMA-bound) and, for reference, the
        stepwise chain's layer 1 (what every chain step cost before the latent form; still the chain for DMM_CHAIN_MODE=stepwise);
      * the CSR SpMM (HBM bound; north_star's figure)."""
    from diffmm_amd import ops, _lib as L
    DE, J = coach.diff_engine, coach.model.engine
    B, H, I, Kp = args.batch, DE.H, DE.I, DE.Kp
    P = DE.P[0]
    A = J.adj
    emb = J.P.view("emb")
    with L.Recorder() as spmm_plan:  # pre-marshalled launch, like the engines' steady state: the timed loop is kernel time, not Python
        ops.spmm_batched([ops.SpmmOp(A, emb, y=J.Bx)])
    ts = timed_launches(lambda: L.replay(spmm_plan), 200)
    sb = A.algorithmic_bytes(4)
    spmm = {"bound": "hbm", "kernel": "spmm_kernel<64> base adjacency", "achieved": sb / ts / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": sb / ts / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": ts * 1e6, "algorithmic_bytes": sb}
    if args.precision != 1:  # ~fp32 mode: the stepwise chain on the split-precision GEMM
        x, h = DE.xa[:B], DE.dh[:B]
        x.normal_()
        t1 = timed_launches(lambda: ops.gemm(x, P.view("in_layers.0.weight"), out=h, trans_b=True, bias=P.view("in_layers.0.bias"), act=L.ACT_TANH,
                                             precision=args.precision), 20)
        flops1 = 2.0 * B * Kp * H * 3
        gemm = {"bound": "mfma", "kernel": "gemm_kernel (bf16 hi/lo split, 3 MFMA products per element): Denoise layer 1", "achieved": flops1 / t1 / 1e12,
                "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flops1 / t1 / 1e12 / MFMA_BF16_PEAK_TF, "traffic": None, "avg_launch_us": t1 * 1e6}
        return gemm, spmm, None
    M = DE.M
    rows = min(DE.chain_rows, coach.handler.graph.users)
    DE._latent_alloc()
    lat = DE._lat
    for m in range(M):
        DE._refresh_bf16_weights(m)
        lat["za"][m][:rows].normal_()
        DE.cxa[m][:rows].normal_()
        ops.cast_bf16(DE.cxa[m][:rows], DE.cxa16[m][:rows])
        ops.cast_bf16(lat["za"][m][:rows], DE.ch16[m][:rows])
    i_mid = max(1, DE.T // 2)
    tl = timed_launches(lambda: ops.gemm_bf16_nt_batched([DE._latent_step_args(m, i_mid, DE.ch16[m][:rows], lat["za"][m][:rows], lat["zb"][m][:rows],
                                                                               lat["hb"][m][:rows]) for m in range(M)]), 20)
    lat_bytes = M * (rows * H * (4 + 4 + 2 + 2) + H * H * 2)  # z in + z out (fp32), h in + h out (bf16), G (bf16); biases negligible
    lat_flops = 2.0 * rows * M * H * H
    # the item-sized GEMMs: the chain's last step as the chain issues it, and the stepwise chain's layer 1 for reference
    t2 = timed_launches(lambda: ops.gemm_bf16_nt_batched([dict(A=DE.ch16[m][:rows], B=DE.w2_16[m], out=DE.cxa[m][:rows], bias=lat["b2_last"][m],
                                                               alpha=float(DE.c1[0]), N=I) for m in range(M)]), 10)
    for m in range(M):
        DE.cxa[m][:rows].normal_()
    t1 = timed_launches(lambda: ops.gemm_bf16_nt_batched([dict(A=DE.cxa16[m][:rows], B=DE.w1_16[m], out_bf16=DE.ch16[m][:rows], bias=DE.bias_t[m][0],
                                                               act=L.ACT_TANH) for m in range(M)]), 10)
    flops1, flops2 = 2.0 * rows * M * Kp * H, 2.0 * rows * M * H * I
    # the chain as the epoch runs it: all T - 1 steps in the persistent kernel (z in registers; per step each workgroup writes its h tile and
    # re-reads the row block from its XCD's L2).  Algorithmic HBM bytes: z_T in, h in / out once -- the kernel is MFMA-bound.
    T = DE.T
    fused = DE._latent_fused()
    if fused:
        tc = timed_launches(lambda: ops.latent_chain([dict(G=lat["G16"][m], g=lat["g"][m], z=lat["za"][m][:rows], h_a=DE.ch16[m][:rows], h_b=lat["hb"][m][:rows],
                                                           bias_t=DE.bias_t[m]) for m in range(M)], rows=rows, T=T, coef=lat["coef"], s_init=lat["s_init"],
                                                     flags=lat["flags"]), 5)
        DE.check_chain_error()
        chain_flops = (T - 1) * lat_flops
        step = {"bound": "mfma", "kernel": f"latent_chain_8ph_kernel (persistent 256x256 LDS-DMA tiles, z in registers, group barrier per step): {T - 1} steps of "
                                           f"{M} modalities x ({rows} x {H}) x ({H} x {H})^T per rebuild", "achieved": chain_flops / tc / 1e12, "peak": MFMA_BF16_PEAK_TF,
                "unit": "TFLOP/s", "frac": chain_flops / tc / 1e12 / MFMA_BF16_PEAK_TF, "traffic": None, "avg_launch_us": tc * 1e6, "us_per_step": tc * 1e6 / (T - 1),
                "launches_per_call": -(-(M * -(-rows // 256)) // max(8, (256 // (H // 256)) & ~7)),
                "mfma_peak_at_held_clock_TF": MFMA_BF16_PEAK_TF * 1.95 / 2.4,
                "frac_of_peak_at_held_clock": chain_flops / tc / 1e12 / (MFMA_BF16_PEAK_TF * 1.95 / 2.4),
                "note_clock": "the 2.5 PF peak is 2.4 GHz; BF16 MFMA loops on random data hold 1.90-1.95 GHz on this part (MI355X_MICROARCH.md, DVFS give-back)",
                "per_step_launch_us": tl * 1e6, "per_step_launch_GBps": lat_bytes / tl / 1e9, "per_step_launch_hbm_frac": lat_bytes / tl / 1e9 / HBM_PEAK_GBS,
                "note": f"one launch per step (DMM_LATENT_FUSED=0, gemm_bf16_nt_8ph_kernel<9>) is HBM-bound: {lat_bytes / 1e6:.0f} MB per step, fp32 z in and out"}
    else:
        step = {"bound": "hbm", "kernel": f"gemm_bf16_nt_8ph_kernel<9> (256x256 LDS-DMA tile, latent-chain epilogue): {M} modalities x ({rows} x {H}) x ({H} x {H})^T per launch, "
                                          f"{T - 1} launches per rebuild", "achieved": lat_bytes / tl / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": lat_bytes / tl / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": tl * 1e6, "algorithmic_bytes": lat_bytes,
                "tflops": lat_flops / tl / 1e12, "mfma_frac": lat_flops / tl / 1e12 / MFMA_BF16_PEAK_TF}
    gemm = {"bound": "mfma", "kernel": f"gemm_bf16_nt_8ph_kernel<6>: last chain step h_0 W2^T -> fp32 view, {M} modalities x ({rows} x {H}) x ({I} x {H})^T, once per rebuild",
            "achieved": flops2 / t2 / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flops2 / t2 / 1e12 / MFMA_BF16_PEAK_TF, "traffic": None,
            "avg_launch_us": t2 * 1e6, "stepwise_layer1_tflops": flops1 / t1 / 1e12, "stepwise_layer1_avg_launch_us": t1 * 1e6,
            "stepwise_layer1_frac": flops1 / t1 / 1e12 / MFMA_BF16_PEAK_TF}
    try:  # fabric-side traffic per launch from the committed PMC passes (profiles/README.md); bench.py itself cannot run rocprofv3
        if args.batch == 4096 and args.workload == "tiktok" and M == 3:
            pmc = json.load(open(os.path.join(REPO, "profiles", "r1_pmc_traffic.json")))
            spmm["traffic"] = pmc["spmm_kernel_64"]["traffic_bytes"]
            spmm["traffic_source"] = "profiles/r1_pmc_traffic.json (committed rocprofv3 --pmc passes of this launch)"
            chain = json.load(open(os.path.join(REPO, "profiles", PMC_CHAIN_JSON)))
            src = f"profiles/{PMC_CHAIN_JSON} (committed rocprofv3 --pmc passes of tools/roofline_target.py; bench.py cannot run a PMC pass on itself)"
            for obj, key in ((step, "latent_chain" if fused else "latent_step"), (gemm, "last_step_fp32_out")):
                if key in chain:
                    e = chain[key]
                    per_call = 2 if key == "latent_chain" else 1  # the chain call is two launches; the PMC figures are per-launch means
                    obj["traffic"], obj["traffic_source"] = e["traffic_bytes"] * per_call, src
                    obj["traffic_note"] = (f"{e['traffic_bytes'] * per_call / 1e6:.0f} MB fabric-side per call vs {e['algorithmic_bytes'] * per_call / 1e6:.0f} MB algorithmic "
                                           f"(L2 hit rate {e.get('l2_hit_rate', float('nan')):.0%}, MFMA pipe busy {e.get('mfma_busy', float('nan')):.0%})")
    except Exception:
        pass
    return step, spmm, gemm


def roofline_spmm_shard(users: int = 1_000_000, items: int = 500_000, D: int = 64):
    """The SpMM where it is HBM-bound: one GPU's share of the 10M-user configuration (1M users x 500k items, TikTok degree law;
    X and Y are 384 MB each in fp32, far beyond L2 / Infinity Cache).  Live HIP-event timing of dmm_spmm_csr_f32 and dmm_spmm_csr_bf16
    (tools/spmm_scale.py); the compulsory bytes of SURVEY 8(d) use each variant's own element sizes."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import spmm_scale
    from diffmm_amd import synth
    g = synth.make_graph(users, items, feats=(), seed=1818, heavy_p=0.011, heavy=(300, 520))
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {}
    names = {"f32": "spmm_kernel<64, 4> (fp32 X, Y)", "bf16_x": "spmm_kernel<64, 8> (bf16 X, fp32 Y; fp32 accumulate) -- what the joint engine launches "
             "at this size (graph.spmm_operand_vec: the fp32 operand is beyond the 256 MiB Infinity Cache)", "bf16_xy": "spmm_kernel<64, 8> (bf16 X, Y; fp32 accumulate)"}
    for v in ("f32", "bf16_x", "bf16_xy"):
        m = spmm_scale.measure(g, D, dev, v)
        out[v] = {"bound": "hbm", "kernel": names[v],
                  "achieved": m["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": m["algorithmic_GBps"] / HBM_PEAK_GBS,
                  "traffic": None, "avg_launch_us": m["us_per_launch"], "algorithmic_bytes": m["algorithmic_bytes"], "gather_GBps": m["gather_GBps"],
                  # the kernel's gather rate (nnz row reads of d elements per launch) over what the memory system gives row gathers: the distance to
                  # what the hardware allows, next to `frac` (the distance to the compulsory-bytes roofline, which the ~9.5 re-reads of every row
                  # cannot reach).  >= 1 against the beyond-cache figure: part of the re-reads is served by L2 / the Infinity Cache
                  "gather_roofline_GBps": list(GATHER_ROOFLINE_GBS), "frac_of_gather_roofline": m["gather_GBps"] / GATHER_ROOFLINE_GBS[1],
                  "frac_of_gather_roofline_beyond_cache": m["gather_GBps"] / GATHER_ROOFLINE_GBS[0]}
        torch.cuda.empty_cache()
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "r2_pmc_spmm_shard.json")))
        for v in out:
            if v in pmc:
                out[v]["traffic"] = pmc[v]["traffic_bytes"]
                out[v]["traffic_source"] = "profiles/r2_pmc_spmm_shard.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/spmm_scale.py, FETCH_SIZE doubled per MI355X_MICROARCH.md)"
    except Exception:
        pass
    out["workload"] = f"synthetic TikTok-degree-law graph, U={users} I={items} E={g.nnz}, d={D}"
    return out


def _pick_cpu_threads():
    """torch-CPU oversubscribes on a many-core host: time a Denoise-layer-shaped GEMM + tanh at 8 .. 128 threads and keep the fastest."""
    cand = [t for t in (8, 16, 32, 64, 128) if t <= (os.cpu_count() or 8)] or [torch.get_num_threads()]
    x, w = torch.randn(2048, 6720), torch.randn(1024, 6720)
    best, times = None, {}
    for t in cand:
        torch.set_num_threads(t)
        torch.tanh(x @ w.t())
        t0 = time.perf_counter()
        for _ in range(3):
            torch.tanh(x @ w.t())
        times[t] = (time.perf_counter() - t0) / 3
        if best is None or times[t] < times[best]:
            best = t
    torch.set_num_threads(best)
    return best, times


def cpu_baseline(args, h, cfg, iters_per_epoch, n_diff, n_joint):
    """The CPU oracle (a validated restatement of the reference's PyTorch-CPU path) timed on this box's host cores on a
    bounded sample of the same workload, extrapolated to one epoch."""
    from oracle import diffmm_oracle as O
    g = h.graph
    U, I, D, M, B, T = g.users, g.items, args.latdim, len(g.feats), args.batch, args.T
    threads, sweep = _pick_cpu_threads()
    rng = np.random.default_rng(0)
    feats = [torch.from_numpy(f) for f in g.feats]
    model = {"u_embs": torch.nn.init.xavier_uniform_(torch.empty(U, D)), "i_embs": torch.nn.init.xavier_uniform_(torch.empty(I, D)),
             "modal_weight": torch.full((M,), 1.0 / M)}
    for n, f in zip(O.modality_names(M), feats):
        model[f"{n}_layer.weight"], model[f"{n}_layer.bias"] = torch.randn(D, f.shape[1]) * 0.05, torch.zeros(D)
    H, te = 1024, 10
    den = lambda: {"emb_layer.weight": torch.randn(te, te) * 0.3, "emb_layer.bias": torch.zeros(te), "in_layers.0.weight": torch.randn(H, I + te) * 0.02,
                   "in_layers.0.bias": torch.zeros(H), "out_layers.0.weight": torch.randn(I, H) * 0.02, "out_layers.0.bias": torch.zeros(I),
                   "gate_layer.weight": torch.randn(D, D) * 0.1, "gate_layer.bias": torch.zeros(D)}
    st = O.EpochState(model, [den() for _ in range(M)], 1e-3, 100)
    hp = dict(T=T, sim_weight=0.01, reg=1e-4, modal_adj_weight=0.2, residual_weight=0.5, noise_degree=1.5, cross_cl_temp=0.2, cross_cl_rate=0.5,
              modal_cl_temp=0.5, modal_cl_rate=0.01, cl_method=0)
    sched = O.schedule(T, 0.5, 0.0001, 0.02)
    dense = torch.zeros(U, I); dense[torch.from_numpy(g.trn_rows.astype(np.int64)), torch.from_numpy(g.trn_cols.astype(np.int64))] = 1.0
    adj = O.csr_to_torch(*O.build_adj_csr(g.trn_rows, g.trn_cols, U, I))
    draws = O.Draws(seed=1)
    idx = rng.permutation(U)[:B]
    x0 = dense[torch.from_numpy(idx)]
    budget_t0 = time.perf_counter()
    t0 = time.perf_counter(); O.diffusion_step(st, feats, x0, draws, sched, hp); t_diff = time.perf_counter() - t0
    # rebuild units: T reverse steps over one user batch + the per-user top-k of that batch, one unit per modality while the budget lasts
    units = []
    deg = np.bincount(g.trn_rows, minlength=U)
    for m in range(M):
        if units and time.perf_counter() - budget_t0 > 22.0:
            break
        t0 = time.perf_counter()
        with torch.no_grad():
            view = O.generate_view(st.dens[m], x0, sched, 0)
            O.topk_edges(view, idx, deg)
        units.append(time.perf_counter() - t0)
    t_reb_unit = float(np.mean(units))
    e = rng.integers(0, g.nnz, size=B)
    u, p = torch.from_numpy(g.trn_rows[e].astype(np.int64)), torch.from_numpy(g.trn_cols[e].astype(np.int64))
    ng = torch.from_numpy(rng.integers(0, I, size=B))
    t0 = time.perf_counter()
    O.joint_step(st, feats, adj, [adj] * M, u, p, ng, draws, hp)
    t_joint = time.perf_counter() - t0
    n_batches = -(-U // B)
    epoch = n_diff * t_diff + n_batches * M * t_reb_unit + n_joint * t_joint
    return {"value": iters_per_epoch / epoch, "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": f"1 diffusion iter ({t_diff:.2f}s) + {len(units)} of {n_batches * M} rebuild units (T={T} reverse steps on one {B}-user batch + its top-k, one "
                      f"modality each: {', '.join(f'{x:.2f}s' for x in units)}) + 1 joint iter ({t_joint:.2f}s), extrapolated to an epoch of {epoch:.1f}s; "
                      f"torch {torch.__version__} CPU at {threads} threads of {os.cpu_count()} cpus (fastest of "
                      f"{', '.join(f'{k}: {v * 1e3:.0f} ms' for k, v in sweep.items())} on a 2048 x 6720 x 1024 GEMM + tanh)"}


def thinned_tiktok_rate(args, E_target: int = 59541, epochs: int = 5):
    """iters/s of the same configuration on the bench graph thinned uniformly to TikTok's own interaction count (extra field; the headline workload
    stays the generator's graph)."""
    import dataclasses
    from diffmm_amd import synth
    from diffmm_amd.Conf import load_config
    from diffmm_amd.DataHandler import DataHandler
    from diffmm_amd.Main import Coach, seed_it
    cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
    cfg.base.gpu = str(torch.cuda.current_device())
    cfg.base.latdim, cfg.hyper.steps, cfg.train.batch = args.latdim, args.T, args.batch
    g = synth.named("tiktok", seed=int(cfg.base.seed))
    keep = np.sort(np.random.default_rng(59541).permutation(g.nnz)[:E_target])
    g = dataclasses.replace(g, trn_rows=g.trn_rows[keep], trn_cols=g.trn_cols[keep])
    seed_it(cfg.base.seed)
    h = DataHandler(cfg)
    h.LoadData(graph_data=g)
    coach = Coach(h, cfg, precision=args.precision)
    coach.prepareModel()
    n_it = len(h.diffusionLoader) + len(h.trainLoader)
    coach.trainEpoch(); coach.trainEpoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        coach.trainEpoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"E": int(g.nnz), "iters_per_epoch": n_it, "iters_per_s": n_it * epochs / dt, "ms_per_epoch": dt / epochs * 1e3}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N`, N > 1, outside torchrun: start the N ranks as children (this process has made no GPU call), relay rank 0's JSON
    line, return the children's status."""
    import subprocess
    if args.backend == "nccl":
        n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} GPUs, this node shows {n_dev} (--backend gloo lets ranks share a GPU: tests only)", file=sys.stderr)
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=REPO, env=env)
    line = None
    for out in p.stdout:  # stderr passes through; stdout is forwarded except that exactly one JSON line is printed, last
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stdout.write(out); sys.stdout.flush()
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line); sys.stdout.flush()
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="tiktok", help="a named synthetic shape (tiktok, baby, sports, mid, toy) or `scale` (--users / --items / ...)")
    ap.add_argument("--T", type=int, default=40); ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--latdim", type=int, default=64); ap.add_argument("--precision", type=int, default=1, choices=(1, 3))
    ap.add_argument("--users", type=int, default=1_000_000, help="--workload scale: users in total (strong) or per GPU (weak)")
    ap.add_argument("--items", type=int, default=100_000); ap.add_argument("--scale-mode", default="strong", choices=("strong", "weak"))
    ap.add_argument("--diff-iters", type=int, default=2); ap.add_argument("--joint-iters", type=int, default=8)
    ap.add_argument("--rebuild-users", type=int, default=4096, help="--workload scale: users per replica whose reverse chain + top-k run in a step")
    ap.add_argument("--no-cpu-baseline", action="store_true"); ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-shard-roofline", action="store_true", help="skip the 1M x 500k SpMM measurement (roofline_spmm_shard)")
    ap.add_argument("--no-fp32-like", action="store_true", help="skip the extra precision-3 (bf16 hi/lo split, ~fp32) epochs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N`, or under torch.distributed.run with "
                         f"--nproc-per-node equal to --gpus)")
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    from diffmm_amd import comm
    if comm.active(world):
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # a single forced-collectives replica started by hand
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        kw = dict(device_id=torch.device("cuda", local)) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, **kw)
    coach, h, cfg = make_coach(args, rank, world)
    n_diff, n_joint = min(len(h.diffusionLoader), getattr(h.diffusionLoader, "bounded", 1 << 60)), len(h.trainLoader)
    iters_per_epoch = n_diff + n_joint
    scale, weak = args.workload == "scale", args.workload == "scale" and args.scale_mode == "weak"
    import torch.distributed as dist  # (only used when a process group exists)
    grouped = comm.active(world)

    def sync():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        coach.trainEpoch()
    sync()
    t0 = time.perf_counter()
    phases = np.zeros(3)
    for _ in range(args.steps):
        coach.trainEpoch()
        phases += np.array(getattr(coach, "phase_seconds", [0, 0, 0]))
    sync()
    dt = time.perf_counter() - t0
    replicas_identical = None
    if grouped:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
        # the replicated parameters (items, projections, Denoise weights or their replicated small tensors) must be bit-identical: they only ever
        # receive all-reduced gradients.  (Coach.check_replicas made the same comparison after every epoch and would have healed / raised.)
        J, DE = coach.model.engine, coach.diff_engine
        nr = DE.n_replicated if coach.item_shard is not None else None
        flat = torch.cat([J.P.data[getattr(J, "replicated_lo", J.nu * J.D):]] + [p.data[:nr] for p in DE.P])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi)) and getattr(coach, "replica_heals", 0) == 0
    if rank == 0:
        report(args, coach, h, cfg, world, dt, phases, n_diff, n_joint, replicas_identical)
    if grouped:  # the other ranks wait here while rank 0 takes its roofline / baseline measurements (none of them communicates)
        dist.barrier()
        dist.destroy_process_group()


def report(args, coach, h, cfg, world, dt, phases, n_diff, n_joint, replicas_identical):
    from diffmm_amd import comm
    iters_per_epoch = n_diff + n_joint
    scale, weak = args.workload == "scale", args.workload == "scale" and args.scale_mode == "weak"
    units = world if weak else 1  # weak form: an iteration of the N-shard job counts as N shard-iterations (fixed per-GPU work per iteration)
    step_desc = (f"step = a bounded sample of an epoch = {n_diff} diffusion iters + reverse chain / top-k of {args.rebuild_users} users per replica + {n_joint} joint iters"
                 + (f"; weak scaling: {args.users} users per GPU, value counts shard-iterations (x{world})" if weak else "; strong scaling: the graph is fixed")
                 if scale else f"step = 1 epoch = {n_diff} diffusion iters + rebuild + {n_joint} joint iters")
    out = {"metric": "BPR+diffusion train iters/sec on TikTok d=64 T=40, at 1/2/4/8 GPUs", "value": units * iters_per_epoch * args.steps / dt, "unit": "iters/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "bf16" if args.precision == 1 else "bf16x3",
           "data": "synthetic",
           "config": {"workload": f"synthetic {args.workload}-shape graph: U={h.graph.users} I={h.graph.items} E={h.graph.nnz}, M={len(h.graph.feats)} modalities, "
                                  f"d={args.latdim}, T={args.T}, batch={args.batch}; {step_desc}",
                      "parallelism": (f"user-shard{world}" + ("" if getattr(coach.model.engine, "replicated_items", True) else "+item-shard")) if comm.active(world) else "single", "backend": args.backend if comm.active(world) else None,
                      "denoise_shard": ("items" if coach.item_shard is not None else "rows") if comm.active(world) else None, "embedding_dtype": "f32", "gemm_dtype": "bf16 MFMA, f32 accumulate" if args.precision == 1 else "bf16 hi/lo x3",
                      "phase_ms_per_epoch": {"diffusion": phases[0] / args.steps * 1e3, "rebuild": phases[1] / args.steps * 1e3, "joint": phases[2] / args.steps * 1e3},
                      # the two per-phase rates next to the blended headline (`value`); host-side phase boundaries of rank 0
                      "phase_iters_per_s": {"diffusion": n_diff * args.steps / max(phases[0], 1e-9), "joint": n_joint * args.steps / max(phases[2], 1e-9)}}}
    if replicas_identical is not None:
        out["replicated_params_identical"] = replicas_identical
    if getattr(coach, "replica_heals", 0):
        out["replica_heals"] = coach.replica_heals
    if not args.no_roofline and coach.item_shard is None and not scale:  # (the roofline replays need the whole Denoise on this rank)
        try:
            g, s, g2 = roofline(coach, args)
            out["roofline"], out["roofline_spmm"] = g, s
            if g2 is not None:
                out["roofline_item_gemm"] = g2
        except Exception as e:  # a replica's engines hold local blocks: never lose the result line over the side measurements
            if not comm.active(world):
                raise
            out["roofline_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_shard_roofline and args.workload == "tiktok":
            out["roofline_spmm_shard"] = roofline_spmm_shard()
    if world == 1 and not comm.active(world) and args.precision == 1 and not args.no_fp32_like and not scale:  # the same epochs in split-bf16 (~fp32) arithmetic, next to the bf16 headline
        coach.precision = 3
        coach.model.precision = 3
        coach.trainEpoch(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            coach.trainEpoch()
        torch.cuda.synchronize()
        out["config"]["fp32_like_iters_per_s"] = iters_per_epoch * 2 / (time.perf_counter() - t1)
        coach.precision = 1
        coach.model.precision = 1
    if world == 1 and not comm.active(world) and not scale and args.workload == "tiktok" and not args.no_fp32_like:
        # the same epochs on a graph of TikTok's own size: the synthetic generator gives E = 70 116 train interactions (mean degree 7.5) where TikTok has
        # 59 541 (6.4) -- 18 joint iterations per epoch instead of 15, the cheapest kind; a uniformly thinned copy of the graph has TikTok's count
        out["config"]["tiktok_E59541"] = thinned_tiktok_rate(args)
    if world == 1 and not args.no_cpu_baseline and not scale:
        out["cpu_baseline"] = cpu_baseline(args, h, cfg, iters_per_epoch, n_diff, n_joint)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One epoch of the whole hot path at a scaled-up synthetic size (default 1M users x 100k items, 3 modalities, d = 256, batch 8192:
BASELINE.json's config 5 scaled to one GPU) -- the memory / index-width / time audit of the scale path.  Prints one JSON line with the
phase times, the peak HBM allocation and the first epoch's losses.

    python tools/scale_epoch.py [--users 1000000 --items 100000 --latdim 256 --T 40 --batch 8192 --joint-iters 0]
--joint-iters N > 0 stops the joint phase after N minibatches (its cost is E / batch full-graph iterations) and extrapolates; --diff-iters likewise;
--rebuild-users N runs the reverse chain + top-k for the first N own users (the others keep their observed interactions as rebuilt edges).
--emulate-world W [--emulate-rank r]: ONE process runs replica r of a W-replica user-sharded job on --users x --items (BASELINE config 5 is
--users 10000000 --items 1000000 --latdim 256 --T 1000 --emulate-world 8): the graph rows of the replica's own users only, its
local adjacency blocks, its item slice of the Denoise MLPs, its share of the rebuild and of the evaluation -- with every collective reduced to
its local part (diffmm_amd/comm.py: LOCAL).  What one GPU of that job computes and holds; the exchanges themselves are not in the timing."""
import argparse, json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from diffmm_amd import synth
from diffmm_amd.Conf import load_config
from diffmm_amd.DataHandler import DataHandler
from diffmm_amd.Main import Coach, seed_it

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000); ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--latdim", type=int, default=256); ap.add_argument("--T", type=int, default=40); ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--joint-iters", type=int, default=0); ap.add_argument("--precision", type=int, default=1)
ap.add_argument("--no-test", action="store_true")
ap.add_argument("--diff-iters", type=int, default=0)
ap.add_argument("--rebuild-users", type=int, default=0, help="reverse chain + top-k for the first N own users only; the others keep their observed interactions")
ap.add_argument("--emulate-world", type=int, default=1); ap.add_argument("--emulate-rank", type=int, default=0)
a = ap.parse_args()
t0 = time.time()
EW, ER = a.emulate_world, a.emulate_rank
cuts = [a.users * k // EW for k in range(EW + 1)]
g = synth.make_graph(a.users, a.items, feats=(128, 768, 128), seed=1818, heavy_p=0.011, heavy=(300, 520), user_lo=cuts[ER], user_hi=cuts[ER + 1])
print(f"graph: U={g.users} I={g.items} E={g.nnz} test={len(g.tst_rows)} ({time.time() - t0:.1f}s)" +
      (f" -- the rows of users {cuts[ER]}..{cuts[ER + 1]} (replica {ER} of {EW})" if EW > 1 else ""), flush=True)
cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
cfg.base.latdim, cfg.hyper.steps, cfg.train.batch, cfg.data.name = a.latdim, a.T, a.batch, "scaled"
seed_it(cfg.base.seed)
h = DataHandler(cfg)
t0 = time.time(); h.LoadData(graph_data=g, sharded=EW > 1); print(f"LoadData {time.time() - t0:.1f}s", flush=True)
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if EW > 1:
    from diffmm_amd.sharded import ShardLayout
    lay = ShardLayout(EW, ER, a.users, a.items, np.asarray(cuts, dtype=np.int64))
    coach = Coach(h, cfg, precision=a.precision, rank=ER, world=EW, layout=lay, emulate_replica=True)
elif world > 1:  # user-sharded over `world` replicas (python -m torch.distributed.run --nproc-per-node W tools/scale_epoch.py ...; DMM_TEST_BACKEND=gloo
    import torch.distributed as dist  # lets the replicas share one GPU: a functional check, not a timing)
    dist.init_process_group(os.environ.get("DMM_TEST_BACKEND", "nccl"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    cfg.base.gpu = str(torch.cuda.current_device())
    coach = Coach(h, cfg, precision=a.precision, rank=rank, world=world)
else:
    coach = Coach(h, cfg, precision=a.precision)
t0 = time.time(); coach.prepareModel(); torch.cuda.synchronize(); print(f"prepareModel {time.time() - t0:.1f}s, {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
if a.joint_iters:
    full = h.trainLoader
    class Head:  # the first N minibatches of the epoch's shuffle
        def __iter__(self_):
            for i, b in enumerate(full):
                if i >= a.joint_iters:
                    return
                yield b
        def __len__(self_):
            return min(len(full), a.joint_iters)
    h.trainLoader = Head()
n_diff_full = len(h.diffusionLoader)
if a.diff_iters:
    full_idx = h.diffusionLoader.index_batches
    h.diffusionLoader.index_batches = lambda: full_idx()[: a.diff_iters]
if coach.item_shard is not None:
    print(f"Denoise item-sharded: items {coach.item_shard.i0}..{coach.item_shard.i1} of {a.items} on this replica; lean chain views: {coach.diff_engine.view_rows} rows", flush=True)
if a.rebuild_users:
    coach.rebuild_limit = a.rebuild_users
torch.cuda.synchronize(); t0 = time.time()
res = coach.trainEpoch()
torch.cuda.synchronize(); t_epoch = time.time() - t0
t0 = time.time(); tst = {} if a.no_test else coach.testEpoch(); torch.cuda.synchronize(); t_test = time.time() - t0
n_joint_full = -(-(g.nnz * EW) // a.batch)  # (an emulated replica holds 1 / EW of the interactions; the job's epoch walks all of them)
out = {"emulated_replica": [ER, EW] if EW > 1 else None, "diff_iters_run": a.diff_iters or n_diff_full, "diff_iters_full_epoch": n_diff_full,
       "denoise_shard": "items" if coach.item_shard is not None else ("rows" if max(world, EW) > 1 else None), "users": g.users, "items": g.items, "E": g.nnz, "d": a.latdim, "T": a.T, "batch": a.batch, "epoch_s": t_epoch, "phase_s": dict(zip(("diffusion", "rebuild", "joint"), coach.phase_seconds)),
       "joint_iters_run": a.joint_iters or n_joint_full, "joint_iters_full_epoch": n_joint_full, "test_s": t_test,
       "peak_alloc_GiB": torch.cuda.max_memory_allocated() / 2**30, "train": {k: float(v) for k, v in res.items()}, "test": tst}
adjs = [coach.image_adj, coach.text_adj] + ([coach.audio_adj] if coach.M == 3 else [])
out.update(world=world, rebuild_users=a.rebuild_users or None, base_block_nnz=int(coach.base_adj.nnz), modal_block_nnz=[int(x.nnz) for x in adjs],
           own_users=int(coach.layout.nu) if coach.layout is not None else g.users, chain_error_word_clear=True)  # (trainEpoch raises on a set error word)
if coach.rebuild_limit:  # the rebuilt rows: every user keeps its degree with distinct items
    nu_chk = min(int(coach.rebuild_limit), out["own_users"])
    rp = adjs[0].rowptr[:nu_chk + 1].long().cpu().numpy(); col = adjs[0].col[:int(rp[-1])].cpu().numpy()
    base_rp = coach.base_adj.rowptr[:nu_chk + 1].long().cpu().numpy()
    out["rebuilt_rows_keep_degree"] = bool((np.diff(rp) == np.diff(base_rp)).all())
    out["rebuilt_rows_distinct"] = bool(all(len(set(col[rp[u]:rp[u + 1]].tolist())) == rp[u + 1] - rp[u] for u in range(0, nu_chk, max(1, nu_chk // 64))))
if rank == 0:
    print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()

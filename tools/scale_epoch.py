#!/usr/bin/env python3
"""One epoch of the whole hot path at a scaled-up synthetic size (default 1M users x 100k items, 3 modalities, d = 256, batch 8192:
BASELINE.json's config 5 scaled to one GPU) -- the memory / index-width / time audit of the scale path.  Prints one JSON line with the
phase times, the peak HBM allocation and the first epoch's losses.

    python tools/scale_epoch.py [--users 1000000 --items 100000 --latdim 256 --T 40 --batch 8192 --joint-iters 0]
--joint-iters N > 0 stops the joint phase after N minibatches (its cost is E / batch full-graph iterations) and extrapolates."""
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
a = ap.parse_args()
t0 = time.time()
g = synth.make_graph(a.users, a.items, feats=(128, 768, 128), seed=1818, heavy_p=0.011, heavy=(300, 520))
print(f"graph: U={g.users} I={g.items} E={g.nnz} test={len(g.tst_rows)} ({time.time() - t0:.1f}s)", flush=True)
cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
cfg.base.latdim, cfg.hyper.steps, cfg.train.batch, cfg.data.name = a.latdim, a.T, a.batch, "scaled"
seed_it(cfg.base.seed)
h = DataHandler(cfg)
t0 = time.time(); h.LoadData(graph_data=g); print(f"LoadData {time.time() - t0:.1f}s", flush=True)
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world > 1:  # user-sharded over `world` replicas (python -m torch.distributed.run --nproc-per-node W tools/scale_epoch.py ...; DMM_TEST_BACKEND=gloo
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
torch.cuda.synchronize(); t0 = time.time()
res = coach.trainEpoch()
torch.cuda.synchronize(); t_epoch = time.time() - t0
t0 = time.time(); tst = {} if a.no_test else coach.testEpoch(); torch.cuda.synchronize(); t_test = time.time() - t0
n_joint_full = -(-g.nnz // a.batch)
out = {"users": g.users, "items": g.items, "E": g.nnz, "d": a.latdim, "T": a.T, "batch": a.batch, "epoch_s": t_epoch, "phase_s": dict(zip(("diffusion", "rebuild", "joint"), coach.phase_seconds)),
       "joint_iters_run": a.joint_iters or n_joint_full, "joint_iters_full_epoch": n_joint_full, "test_s": t_test,
       "peak_alloc_GiB": torch.cuda.max_memory_allocated() / 2**30, "train": {k: float(v) for k, v in res.items()}, "test": tst}
out["world"] = world
if rank == 0:
    print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()

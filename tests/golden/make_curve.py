#!/usr/bin/env python3
"""Run the REFERENCE (imported, shimmed) for a few epochs on the synthetic TikTok-shape graph and record
its per-epoch losses / Recall@20 / NDCG@20 -> baselines/ref_cpu_synth_tiktok_*.csv (the statistical parity
band for the HIP path, which cannot share RNG streams with torch-CPU).  Build container only.

    python tests/golden/make_curve.py --epochs 20 --T 5 --batch 1024 --seed 1818
"""
import argparse, csv, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # installs the shims and imports the reference
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=20); ap.add_argument("--T", type=int, default=5)
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--seed", type=int, default=1818)
ap.add_argument("--threads", type=int, default=4); ap.add_argument("--train_epochs_cfg", type=int, default=100)
a = ap.parse_args()
torch.set_num_threads(a.threads)
g = MG.synth.named("tiktok")
cfg = MG.mkcfg("tiktok", base__latdim=64, base__seed=a.seed, hyper__steps=a.T, hyper__noise_scale=0.5, hyper__noise_degree=1.5,
               hyper__cross_cl_rate=0.5, hyper__cross_cl_temp=0.2, hyper__modal_cl_rate=0.01, hyper__modal_cl_temp=0.5,
               hyper__sim_weight=0.01, train__batch=a.batch, train__test_batch=1024, train__reg=1e-4, train__lr=1e-3,
               train__epoch=a.train_epochs_cfg)  # == conf/tiktok.toml
MG.RMain.seed_it(a.seed)
h, _, _ = MG.build_handler(cfg, g, a.batch, seed=a.seed)
# real shuffling loaders, as the reference uses (DataHandler.py:119,129)
h.trainLoader = torch.utils.data.DataLoader(h.trainData, batch_size=a.batch, shuffle=True, num_workers=0)
h.diffusionLoader = torch.utils.data.DataLoader(h.diffusionData, batch_size=a.batch, shuffle=True, num_workers=0)
coach = MG.RMain.Coach(h, cfg); coach.prepareModel()
out = os.path.join(MG.REPO, "baselines", f"ref_cpu_synth_tiktok_T{a.T}_B{a.batch}_seed{a.seed}.csv")
rows = []
for ep in range(a.epochs):
    t0 = time.time(); res = coach.trainEpoch(); t1 = time.time()
    coach.model_scheduler.step(); coach.image_scheduler.step(); coach.text_scheduler.step(); coach.audio_scheduler.step()
    with torch.no_grad():
        tres = coach.testEpoch()
    row = dict(epoch=ep, train_s=round(t1 - t0, 2), **{k: round(float(v), 6) for k, v in res.items()}, **{k: round(float(v), 6) for k, v in tres.items()})
    rows.append(row); print(row, flush=True)
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)

#!/usr/bin/env python3
"""Train the HIP path on the synthetic TikTok-shape graph with the reference's conf/tiktok.toml hyper-parameters (T=5, B=1024,
seed 1818) and record per-epoch losses / Recall@20 / NDCG@20 next to the reference's own run on the same data
(baselines/ref_cpu_synth_tiktok_T5_B1024_seed1818.csv, tests/golden/make_curve.py).  RNG streams differ, so the comparison is statistical."""
import argparse, csv, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from diffmm_amd.Conf import load_config
from diffmm_amd.DataHandler import DataHandler
from diffmm_amd.Main import Coach, seed_it

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=30); ap.add_argument("--precision", type=int, default=1); ap.add_argument("--seed", type=int, default=1818)
ap.add_argument("--T", type=int, default=5); ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--out", default=None)
a = ap.parse_args()
cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
cfg.base.seed, cfg.hyper.steps, cfg.train.batch = a.seed, a.T, a.batch
seed_it(a.seed)
h = DataHandler(cfg); h.LoadData(graph_data=__import__("diffmm_amd.synth", fromlist=["named"]).named("tiktok", seed=1818))
coach = Coach(h, cfg, precision=a.precision); coach.prepareModel()
out = a.out or os.path.join(REPO, "baselines", f"hip_synth_tiktok_T{a.T}_B{a.batch}_seed{a.seed}_prec{a.precision}.csv")
rows = []
for ep in range(a.epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = coach.trainEpoch()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tres = coach.testEpoch()
    row = dict(epoch=ep, train_s=round(t1 - t0, 4), **{k: round(float(v), 6) for k, v in res.items()}, **{k: round(float(v), 6) for k, v in tres.items()})
    rows.append(row)
    if ep % 5 == 0 or ep == a.epochs - 1:
        print(row, flush=True)
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
print("wrote", out)

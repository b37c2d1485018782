#!/usr/bin/env python3
"""Experiment: rebuild with the fp32-state chain vs the bf16-state chain on the same (briefly trained) weights: time and edge overlap."""
import os, sys, time, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--epochs", type=int, default=3); a = ap.parse_args()
args = argparse.Namespace(latdim=64, T=40, batch=4096, workload="tiktok", precision=1)
coach, h, cfg = bench.make_coach(args, 0, 1)
for _ in range(a.epochs):
    coach.trainEpoch()
DE, dd = coach.diff_engine, h.diffusionData
batches = [idx for _, idx in h.diffusionLoader]
res = {}
for state in ("fp32", "bf16", "fp32", "bf16"):
    DE.hp["chain_state"] = state
    torch.cuda.synchronize(); t0 = time.perf_counter()
    edges = DE.rebuild_edges(dd.rows, batches, dd.deg_ptr32, int(dd.ptr[-1]), precision=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[state] = [e.clone() for e in edges]
    print(f"{state}: rebuild chains + top-k {dt*1e3:.1f} ms")
ptr = dd.ptr.cpu().numpy()
for m in range(3):
    a_, b_ = res["fp32"][m].cpu().numpy(), res["bf16"][m].cpu().numpy()
    same = 0
    for u in range(dd.users):
        same += len(set(a_[ptr[u]:ptr[u+1]].tolist()) & set(b_[ptr[u]:ptr[u+1]].tolist()))
    print(f"modality {m}: {same}/{int(ptr[-1])} edges identical ({same/int(ptr[-1]):.5f})")

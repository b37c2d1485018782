#!/usr/bin/env python3
"""Run-to-run spread of the paired 12-epoch run (tests/test_epoch_gpu.py::test_paired_twelve_epochs_follow_the_reference): the same seeded problem
N times per arithmetic mode; prints the plateau Recall@20 / NDCG@20 differences from the reference's trajectory for every run.  The float atomics of
the scatter kernels make the order of the gradient sums differ between runs, and 12 graph rebuilds amplify that; the test's bound has to sit above
this spread.  usage: paired_spread.py [runs=6]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_epoch_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
z = T.load("epoch_paired.npz")
for precision in (3, 1):
    out = []
    for run in range(n):
        coach, h, M = T._coach(z)
        coach.precision = precision
        coach.model.precision = precision
        draws = T._Draws(z)
        rec, ndcg, worst = [], [], 0.0
        for ep in range(int(z["epochs"])):
            h.trainData.set_negs(z[f"ep{ep}/negs"])
            res = coach.trainEpoch(draws=draws)
            tres = coach.testEpoch()
            worst = max(worst, max(abs(float(res[k]) - float(z[f"ep{ep}/train/{k}"])) / abs(float(z[f"ep{ep}/train/{k}"])) for k in ("Loss", "BPR Loss", "CL loss")))
            rec.append(float(tres["Recall"]) - float(z[f"ep{ep}/test/Recall"]))
            ndcg.append(float(tres["NDCG"]) - float(z[f"ep{ep}/test/NDCG"]))
        out.append((np.mean(rec[8:]), np.mean(ndcg[8:]), worst))
        print(f"precision {precision} run {run}: plateau Recall@20 {out[-1][0]:+.4f} NDCG@20 {out[-1][1]:+.4f} worst loss deviation {worst:.4%}", flush=True)
    r = np.array([o[0] for o in out]); g = np.array([o[1] for o in out])
    print(f"precision {precision}: Recall difference mean {r.mean():+.4f} sd {r.std(ddof=1):.4f} range [{r.min():+.4f}, {r.max():+.4f}]; NDCG mean {g.mean():+.4f} sd {g.std(ddof=1):.4f}")

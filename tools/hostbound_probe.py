import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import argparse
from bench import make_coach
a = argparse.Namespace(latdim=64, T=40, batch=4096, workload="tiktok", precision=1)
coach, h, cfg = make_coach(a, 0, 1)
for _ in range(2): coach.trainEpoch()
J = coach.model.engine
batches = list(h.trainLoader)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for u, p, n in batches:
        J.step(u, p, n, precision=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"joint x{len(batches)}: host issue {1e3*(t1-t0):.2f} ms, wall {1e3*(t2-t0):.2f} ms; per iter host {1e6*(t1-t0)/len(batches):.0f} us wall {1e6*(t2-t0)/len(batches):.0f} us; calls per plan {len(next(iter(J._plans.values()))['calls'])}")

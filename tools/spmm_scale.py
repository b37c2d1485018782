#!/usr/bin/env python3
"""SpMM at shard scale: a synthetic graph of the TikTok degree law with 1M users (one GPU's share of the 10M-user configuration),
operands far beyond L2 / Infinity Cache.  Reports time, algorithmic GB/s (nnz*8 + (N+1)*4 + 2*N*d*4 bytes) and the gather volume."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import graph, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000); ap.add_argument("--items", type=int, default=500_000)
ap.add_argument("--dims", default="64,256")
a = ap.parse_args()
t0 = time.time()
g = synth.make_graph(a.users, a.items, feats=(), seed=1818, heavy_p=0.011, heavy=(300, 520))
print(f"graph: U={g.users} I={g.items} E={g.nnz} ({time.time()-t0:.1f}s)", flush=True)
dev = torch.device("cuda:0")
for D in [int(x) for x in a.dims.split(",")]:
    t0 = time.time()
    adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, D, dev)
    print(f"d={D}: nnz={adj.nnz} tasks={adj.n_tasks} partial_blocks={adj.n_partial_blocks} giant_rows={0 if adj.fix is None else adj.fix.shape[0]} "
          f"(host build {time.time()-t0:.1f}s)", flush=True)
    X = torch.randn(adj.n, D, device=dev); Y = torch.empty_like(X)
    op = ops.SpmmOp(adj, X, y=Y)
    for remap in ("0", "1"):
        os.environ["DMM_SPMM_REMAP"] = remap
        ops.spmm_batched([op]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.spmm_batched([op])
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        alg = adj.algorithmic_bytes(4)
        print(f"   remap={remap}: {t*1e6:9.1f} us   algorithmic {alg/1e9:.2f} GB -> {alg/t/1e9:8.1f} GB/s ({alg/t/8e12:.3f} of 8 TB/s);  "
              f"gather volume {adj.nnz*D*4/1e9:.2f} GB -> {adj.nnz*D*4/t/1e9:8.1f} GB/s", flush=True)
    del adj, X, Y, op
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""SpMM at shard scale: a synthetic graph of the TikTok degree law with 1M users x 500k items (one GPU's share of the 10M-user
configuration), operands far beyond L2 / Infinity Cache.  Per variant (fp32 operands: dmm_spmm_csr_f32; bf16 operands:
dmm_spmm_csr_bf16): time per launch (HIP events over back-to-back launches on the launch stream), the compulsory-bytes rate
(nnz*8 + (N+1)*4 + 2*N*d*s bytes, s = element size of the variant's X and Y) against the 8 TB/s roofline, and the gather
volume nnz*d*s.  Prints one JSON line per variant (kept under profiles/)."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import graph, ops, synth, _lib as L


def measure(g, D, dev, variant, reps=10, task_len=graph.TASK_LEN, order="row"):
    vec = 8 if variant.startswith("bf16") else 4
    if order == "cluster":  # the generator's own communities as the schedule key: the ceiling of what any locality ordering could give
        rowptr, col, val = graph.normalized_bipartite_csr(g.trn_rows, g.trn_cols, g.users, g.items)
        key = np.concatenate([g.clusters[0].astype(np.int64), g.clusters[1].astype(np.int64) + 1000])
        adj = graph.adjacency_from_csr(rowptr, col, val, D, dev, task_len, vec=vec, row_key=key)
    else:
        adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, D, dev, vec=vec, task_len=task_len, hub_order=order == "hub")
    X = torch.randn(adj.n, D, device=dev)
    Y = torch.empty_like(X)
    if variant == "bf16_x_hot":  # EXPERIMENT: the n_hot most-referenced X rows staged in LDS (dmm_spmm_csr_bf16_hotlds); everything else as bf16_x
        import ctypes as C
        n_hot = int(os.environ.get("HOT_ROWS", 896))
        col = adj.col.cpu().numpy()
        refs = np.bincount(col, minlength=adj.n)
        hot = np.argsort(-refs, kind="stable")[:n_hot].astype(np.int32)
        slot = np.full(adj.n, -1, dtype=np.int64); slot[hot] = np.arange(n_hot)
        col_hot = np.where(slot[col] >= 0, -1 - slot[col], col).astype(np.int32)
        share = float(refs[hot].sum()) / float(refs.sum())
        hot_d, col_d = torch.from_numpy(hot).to(dev), torch.from_numpy(col_hot).to(dev)
        X16 = X.bfloat16()
        op = ops.SpmmOp(adj, X16, y=Y)
        if adj.n_partial_blocks:
            op.p.partial = L.ptr(adj.partial_buffer(0))
        grid = int(os.environ.get("HOT_GRID", 256))
        run = lambda: L.call("dmm_spmm_csr_bf16_hotlds", C.byref(op.p), L.ptr(hot_d), n_hot, L.ptr(col_d), grid, L.stream())
        run(); torch.cuda.synchronize()
        ref = torch.empty_like(Y)
        ops.spmm_batched([ops.SpmmOp(adj, X16, y=ref)]); torch.cuda.synchronize()
        assert torch.equal(Y, ref), "hot-LDS variant must reproduce dmm_spmm_csr_bf16 bit for bit (same products, same order)"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        alg = adj.nnz * 8 + (adj.n + 1) * 4 + adj.n * D * (2 + 4)
        return dict(variant=variant, hot_rows=n_hot, grid=grid, share_of_gathers_from_lds=share, users=g.users, items=g.items, nnz=adj.nnz, d=D,
                    us_per_launch=t * 1e6, algorithmic_bytes=alg, algorithmic_GBps=alg / t / 1e9, frac_of_8TBps=alg / t / 8e12,
                    gather_bytes=adj.nnz * D * 2, gather_GBps=adj.nnz * D * 2 / t / 1e9)
    if variant == "f32":
        op, s_x, s_y = ops.SpmmOp(adj, X, y=Y), 4, 4
    elif variant == "bf16_x":      # bf16 operand, fp32 output (what a layer followed by fp32 consumers writes)
        op, s_x, s_y = ops.SpmmOp(adj, X.bfloat16(), y=Y), 2, 4
    else:                           # bf16 operand, bf16 output only (an inner layer of a bf16-storage chain)
        op, s_x, s_y = ops.SpmmOp(adj, X.bfloat16(), y16=torch.empty(adj.n, D, device=dev, dtype=torch.bfloat16)), 2, 2
    with L.Recorder() as plan:
        ops.spmm_batched([op])
    L.replay(plan); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.replay(plan)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    alg = adj.nnz * 8 + (adj.n + 1) * 4 + adj.n * D * (s_x + s_y)
    return dict(variant=variant, order=order, task_len=task_len, users=g.users, items=g.items, nnz=adj.nnz, d=D, tasks=adj.n_tasks, giant_rows=0 if adj.fix is None else int(adj.fix.shape[0]),
                us_per_launch=t * 1e6, algorithmic_bytes=alg, algorithmic_GBps=alg / t / 1e9, frac_of_8TBps=alg / t / 8e12,
                gather_bytes=adj.nnz * D * s_x, gather_GBps=adj.nnz * D * s_x / t / 1e9)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_000_000); ap.add_argument("--items", type=int, default=500_000)
    ap.add_argument("--dims", default="64"); ap.add_argument("--variants", default="f32,bf16_x,bf16_xy"); ap.add_argument("--task-lens", default=str(graph.TASK_LEN)); ap.add_argument("--orders", default="row")
    a = ap.parse_args()
    t0 = time.time()
    g = synth.make_graph(a.users, a.items, feats=(), seed=1818, heavy_p=0.011, heavy=(300, 520))
    print(f"graph: U={g.users} I={g.items} E={g.nnz} ({time.time()-t0:.1f}s)", flush=True)
    dev = torch.device("cuda:0")
    for D in [int(x) for x in a.dims.split(",")]:
        for v in a.variants.split(","):
            for tl in [int(x) for x in a.task_lens.split(",")]:
                for od in a.orders.split(","):  # row: tasks in row order; hub: rows sharing a highest-degree neighbour scheduled together (graph.hub_keys); cluster: by the generator's communities
                    print(json.dumps(measure(g, D, dev, v, task_len=tl, order=od)), flush=True)
                    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Tuning probe (not part of the product): times kernel variants on the GPU with HIP events."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, graph, synth, _lib as L

def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n  # us

dev = torch.device("cuda:0")
out = {}
if "gemm" in sys.argv or len(sys.argv) == 1:
    B, I, H, Kp = 4096, 6710, 1024, 6720
    x16 = torch.randn(B, Kp, device=dev).bfloat16(); w1 = torch.randn(H, Kp, device=dev).bfloat16() * 0.02
    h16 = torch.randn(B, H, device=dev).bfloat16(); w2 = torch.randn(I, H, device=dev).bfloat16() * 0.02
    c32, r32, c16 = torch.zeros(B, Kp, device=dev), torch.randn(B, Kp, device=dev), torch.zeros(B, Kp, dtype=torch.bfloat16, device=dev)
    hb = torch.zeros(B, H, dtype=torch.bfloat16, device=dev); b1 = torch.zeros(H, device=dev); b2 = torch.zeros(I, device=dev)
    f1, f2 = 2.0 * B * Kp * H, 2.0 * B * H * I
    t = timed(lambda: ops.gemm_bf16_nt(x16, w1, out_bf16=hb, bias=b1, act=L.ACT_TANH)); out["L1 full"] = (t, f1 / t / 1e6)
    t = timed(lambda: ops.gemm_bf16_nt(h16, w2, out_bf16=c16, N=I)); out["L2 bf16-out only"] = (t, f2 / t / 1e6)
    t = timed(lambda: ops.gemm_bf16_nt(h16, w2, out=c32, N=I)); out["L2 f32-out only"] = (t, f2 / t / 1e6)
    t = timed(lambda: ops.gemm_bf16_nt(h16, w2, out=c32, out_bf16=c16, bias=b2, residual=r32, e1=.5, e2=.5, N=I)); out["L2 full"] = (t, f2 / t / 1e6)
    for K in (2048, 4096):
        hk = torch.randn(B, K, device=dev).bfloat16(); wk = torch.randn(I, K, device=dev).bfloat16() * 0.02
        t = timed(lambda: ops.gemm_bf16_nt(hk, wk, out_bf16=c16, N=I)); out[f"L2-shape K={K} bf16-out"] = (t, 2.0 * B * K * I / t / 1e6)
    Mc = 9308  # the reverse chain's real launches: 3 modalities batched, all users
    xs = [torch.randn(Mc, Kp, device=dev).bfloat16() for _ in range(3)]; hs = [torch.randn(Mc, H, device=dev).bfloat16() for _ in range(3)]
    hbs = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    cs = [torch.zeros(Mc, Kp, device=dev) for _ in range(3)]; rs = [torch.randn(Mc, Kp, device=dev) for _ in range(3)]
    c16s = [torch.zeros(Mc, Kp, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    t = timed(lambda: ops.gemm_bf16_nt_batched([dict(A=xs[m], B=w1, out_bf16=hbs[m], bias=b1, act=L.ACT_TANH) for m in range(3)]), 10)
    out["chain L1 x3 (M=9308)"] = (t, 3 * 2.0 * Mc * Kp * H / t / 1e6)
    t = timed(lambda: ops.gemm_bf16_nt_batched([dict(A=hs[m], B=w2, out=cs[m], out_bf16=c16s[m], bias=b2, residual=rs[m], e1=.5, e2=.5, N=I) for m in range(3)]), 10)
    out["chain L2 x3 (M=9308) full epilogue"] = (t, 3 * 2.0 * Mc * H * I / t / 1e6)
    a = torch.randn(4096, 4096, device=dev).bfloat16(); b = torch.randn(4096, 4096, device=dev).bfloat16(); cc = torch.zeros(4096, 4096, dtype=torch.bfloat16, device=dev)
    t = timed(lambda: ops.gemm_bf16_nt(a, b, out_bf16=cc)); out["4096^3"] = (t, 2.0 * 4096 ** 3 / t / 1e6)
if "spmm" in sys.argv or len(sys.argv) == 1:
    g = synth.named("tiktok")
    for tl in (16, 32, 64):
        adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, 64, dev, task_len=tl)
        X = torch.randn(adj.n, 64, device=dev); Y = torch.empty_like(X)
        op = ops.SpmmOp(adj, X, y=Y)
        t = timed(lambda: ops.spmm_batched([op]), 100)
        out[f"spmm task_len={tl} tasks={adj.n_tasks} head={adj.n_head_tasks} remap={os.environ.get('DMM_SPMM_NO_REMAP','0')!='1'}"] = (t, adj.algorithmic_bytes() / t / 1e3)
    op4 = [ops.SpmmOp(adj, X, y=torch.empty_like(X)) for _ in range(4)]
    t = timed(lambda: ops.spmm_batched(op4), 100); out["spmm x4 batched (last task_len)"] = (t, 4 * adj.algorithmic_bytes() / t / 1e3)
    # launch floor: an almost empty kernel through the same path
    tiny = graph.make_adjacency(np.array([0]), np.array([0]), 2, 2, 64, dev)
    Xt = torch.randn(4, 64, device=dev); opt = ops.SpmmOp(tiny, Xt, y=torch.empty_like(Xt))
    out["spmm tiny (launch floor)"] = (timed(lambda: ops.spmm_batched([opt]), 200), 0)
for k, v in out.items():
    print(f"{k:70s} {v[0]:9.2f} us   {v[1]:9.1f} (TF/s or GB/s)")

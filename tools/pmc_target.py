#!/usr/bin/env python3
"""Profiling target: launches ONLY the roofline kernels of bench.py (reverse-chain GEMM layer 1 / layer 2, base-graph SpMM) a few
times each, so rocprofv3 --pmc / --kernel-trace rows can be attributed without the rest of the epoch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, graph, synth, _lib as L

dev = torch.device("cuda:0")
B, I, H, Kp = 4096, 6710, 1024, 6720
x16 = torch.randn(B, Kp, device=dev).bfloat16(); w1 = (torch.randn(H, Kp, device=dev) * 0.02).bfloat16()
h16 = torch.randn(B, H, device=dev).bfloat16(); w2 = (torch.randn(I, H, device=dev) * 0.02).bfloat16()
c32, r32, c16 = torch.zeros(B, Kp, device=dev), torch.randn(B, Kp, device=dev), torch.zeros(B, Kp, dtype=torch.bfloat16, device=dev)
hb = torch.zeros(B, H, dtype=torch.bfloat16, device=dev); b1 = torch.zeros(H, device=dev); b2 = torch.zeros(I, device=dev)
g = synth.named("tiktok")
adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, 64, dev)
X = torch.randn(adj.n, 64, device=dev); Y = torch.empty_like(X)
op = ops.SpmmOp(adj, X, y=Y)
for _ in range(10):
    ops.gemm_bf16_nt(x16, w1, out_bf16=hb, bias=b1, act=L.ACT_TANH)
for _ in range(10):
    ops.gemm_bf16_nt(h16, w2, out=c32, out_bf16=c16, bias=b2, residual=r32, e1=.5, e2=.5, N=I)
for _ in range(20):
    ops.spmm_batched([op])
torch.cuda.synchronize()
print("algorithmic: L1 flops", 2.0 * B * Kp * H, "L2 flops", 2.0 * B * H * I, "spmm bytes", adj.algorithmic_bytes())

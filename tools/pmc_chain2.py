#!/usr/bin/env python3
"""Profiling target: the reverse chain's two batched launches at the bench shape (3 modalities, 9308 users), 4 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
Mc, I, Kp, H = 9308, 6710, 6720, 1024
w1 = [(torch.randn(H, Kp, device=dev) * 0.02).bfloat16() for _ in range(3)]; b1 = torch.zeros(H, device=dev)
w2 = [(torch.randn(I, H, device=dev) * 0.02).bfloat16() for _ in range(3)]; b2 = torch.zeros(I, device=dev)
x16 = [torch.randn(Mc, Kp, device=dev).bfloat16() for _ in range(3)]
h16 = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
x32 = [torch.randn(Mc, Kp, device=dev) for _ in range(3)]
y32 = [torch.zeros(Mc, Kp, device=dev) for _ in range(3)]
y16 = [torch.zeros(Mc, Kp, dtype=torch.bfloat16, device=dev) for _ in range(3)]
for _ in range(4):
    ops.gemm_bf16_nt_batched([dict(A=x16[m], B=w1[m], out_bf16=h16[m], bias=b1, act=L.ACT_TANH) for m in range(3)])
    ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], out=y32[m], out_bf16=y16[m], bias=b2, residual=x32[m], e1=0.9, e2=0.1, N=I) for m in range(3)])
torch.cuda.synchronize()

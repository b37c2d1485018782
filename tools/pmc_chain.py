#!/usr/bin/env python3
"""Profiling target: the reverse chain's batched layer-1 GEMM (3 x [9308 x 6720] x [1024 x 6720]^T), 6 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
Mc, Kp, H = 9308, 6720, 1024
w1 = (torch.randn(H, Kp, device=dev) * 0.02).bfloat16(); b1 = torch.zeros(H, device=dev)
xs = [torch.randn(Mc, Kp, device=dev).bfloat16() for _ in range(3)]
hbs = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
for _ in range(6):
    ops.gemm_bf16_nt_batched([dict(A=xs[m], B=w1, out_bf16=hbs[m], bias=b1, act=L.ACT_TANH) for m in range(3)])
torch.cuda.synchronize()

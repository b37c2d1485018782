#!/usr/bin/env python3
"""One dmm_gemm_bf16_nt shape launched back to back for several seconds, timed in chunks: does the rate hold under sustained load?
usage: gemm_sustained.py M,N,K [chunks] [launches per chunk] [randn|zeros]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1].split(","))
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
per = int(sys.argv[3]) if len(sys.argv) > 3 else 100
fill = sys.argv[4] if len(sys.argv) > 4 else "randn"
A = (torch.randn(M, K, device=dev) if fill == "randn" else torch.zeros(M, K, device=dev)).bfloat16()
B = ((torch.randn(N, K, device=dev) * 0.02) if fill == "randn" else torch.zeros(N, K, device=dev)).bfloat16()
out = torch.empty(M, (N + 63) & ~63, device=dev)
with L.Recorder() as plan:
    ops.gemm_bf16_nt_batched([dict(A=A, B=B, N=N, out=out)])
L.replay(plan); torch.cuda.synchronize()
ts = []
for c in range(chunks):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(per):
        L.replay(plan)
    e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1) / per * 1e3, 1))
print(json.dumps(dict(shape=sys.argv[1], fill=fill, us_per_launch_by_chunk=ts, TFLOPs_first=round(2.0 * M * N * K / ts[0] / 1e6, 1), TFLOPs_last=round(2.0 * M * N * K / ts[-1] / 1e6, 1))))

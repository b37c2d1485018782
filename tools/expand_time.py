#!/usr/bin/env python3
"""dmm_item_expand_fused alone at the diffusion step's shape (default B = 4096, I = 6710, d = 64; argv: B I d): forward mode and loss-gradient
modes, HIP-event time.  DMM_EXPAND_WIDE=1 (default): the strip kernel (P fragments resident over a strip of item tiles, F staged per tile in LDS); 0: the
[64 rows, 256 items] tile.  EXPAND_F16=0: without the bf16 copy of F."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops
dev = torch.device("cuda:0")
B, I, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 6710, 64)
Kp = (I + 10 + 63) // 64 * 64
P, F = torch.randn(B, D, device=dev), torch.randn(I, D, device=dev)
R = torch.randn(B, Kp, device=dev)
y16, yT = torch.zeros(B, Kp, device=dev, dtype=torch.bfloat16), torch.zeros(Kp, B, device=dev, dtype=torch.bfloat16)
w = torch.rand(B, device=dev, dtype=torch.float64) + 0.5
bits = torch.randint(0, 2 ** 31 - 1, (B, (I + 31) // 32), device=dev, dtype=torch.int32)
x0 = (torch.rand(B, Kp, device=dev) < 0.01).float()
cs = torch.zeros(I, device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"B={B} I={I} d={D} DMM_EXPAND_WIDE={os.environ.get('DMM_EXPAND_WIDE', '1')}")
F16 = F.bfloat16() if os.environ.get("EXPAND_F16", "1") == "1" else None  # (EXPAND_F16=0: the kernel stages the fp32 F)
print("forward      %.1f us" % t(lambda: ops.item_expand_fused(P, F, R, y16, yT, I, Kp, Kp, F16=F16)))
print("loss, bits   %.1f us" % t(lambda: ops.item_expand_fused(P, F, R, y16, yT[:I], I, Kp, I, w=w, gscale=1e-3, csum=cs, x0bits=bits, F16=F16)))
print("loss, dense  %.1f us" % t(lambda: ops.item_expand_fused(P, F, R, y16, yT[:I], I, Kp, I, w=w, gscale=1e-3, csum=cs, x0=x0, F16=F16)))

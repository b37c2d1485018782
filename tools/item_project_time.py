#!/usr/bin/env python3
"""dmm_item_project alone at the diffusion step's shape ([4096, 6720] x [6710, 64]): HIP-event time per call over back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops
dev = torch.device("cuda:0")
B, I, D = 4096, 6710, 64
X = torch.randn(B, 6720, device=dev); F = torch.randn(I, D, device=dev); out = torch.zeros(B, D, device=dev)
for _ in range(3):
    ops.item_project(X, F, out, I)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.item_project(X, F, out, I)
e1.record(); torch.cuda.synchronize()
print(f"item_project {B}x{I}x{D}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (DMM_ITEM_KS={os.environ.get('DMM_ITEM_KS', 'auto')})")

"""dmm_topk_rows at the rebuild's shape (9308 rows x 6710 columns, ld 6720) under three degree profiles: the TikTok law, every k = 2, every k = 400;
next to a pure read of the same buffer (torch sum): is the kernel bound by bytes, by the light rows' rounds or by the heavy rows' radix passes?"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, synth
dev = torch.device("cuda:0")
g = synth.named("tiktok")
U, I, Kp = g.users, g.items, 6720
x = torch.randn(U, Kp, device=dev)
deg = np.bincount(g.trn_rows, minlength=U)
def run(deg, tag):
    ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)).to(dev)
    out = torch.zeros(int(ptr[-1]) + 1, dtype=torch.int32, device=dev)
    ids = torch.arange(U, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.topk_rows(x, ids, ptr, out, I)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.topk_rows(x, ids, ptr, out, I)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:28s} {e0.elapsed_time(e1) * 100:.1f} us per launch")
run(deg, "TikTok degree law")
run(np.full(U, 2), "every k = 2")
run(np.full(U, 8), "every k = 8")
run(np.full(U, 9), "every k = 9 (radix path)")
run(np.full(U, 400), "every k = 400")
h = deg.copy(); h[deg > 8] = 2
run(h, "TikTok law, heavy rows -> 2")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x.sum(); torch.cuda.synchronize(); e0.record()
for _ in range(10):
    x.sum()
e1.record(); torch.cuda.synchronize()
print(f"{'torch sum of the buffer':28s} {e0.elapsed_time(e1) * 100:.1f} us")

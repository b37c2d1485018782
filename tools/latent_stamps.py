"""Debug only (DMM_LIB=build_timing/libdiffmm_hip_timing.so, built by tools/build_timing_lib.sh): where a step of the persistent latent chain
kernel spends its time, per workgroup (100 MHz wall clock): 0 step start, 1 first LDS units landed (prologue), 2 main loop done, 3 epilogue code done
(stores issued), 4 stores drained + workgroup barrier, 5 group barrier passed (arrive, poll, acquire)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
Mc, H, T = int(os.environ.get("ROWS", 9308)), 1024, 40
torch.manual_seed(0)
G = [(torch.randn(H, H, device=dev) * 0.02).bfloat16() for _ in range(3)]; g = torch.zeros(H, device=dev); bt = torch.zeros(T, H, device=dev)
z32 = [torch.randn(Mc, H, device=dev) for _ in range(3)]
h16 = [torch.randn(Mc, H, device=dev).bfloat16() for _ in range(3)]; hb16 = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
coef = torch.zeros(2 * T, device=dev); coef[0::2] = 0.9; coef[1::2] = 0.5
flags = torch.zeros(514, dtype=torch.int32, device=dev)
chain = lambda: ops.latent_chain([dict(G=G[m], g=g, z=z32[m], h_a=h16[m], h_b=hb16[m], bias_t=bt) for m in range(3)], rows=Mc, T=T, coef=coef, s_init=0.5, flags=flags)
for _ in range(3):
    chain()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); chain(); e1.record(); torch.cuda.synchronize()
print(f"chain call: {e0.elapsed_time(e1) * 1e3:.0f} us, error word {int(flags[-1])}")
n = 512 * 64 * 8
buf = np.zeros(n, dtype=np.uint64)
assert L.lib().dmm_debug_latent_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(n)) == 0
s = buf.reshape(512, 64, 8).astype(np.int64)  # (the stamps of the LAST launch of the call: the second slice of groups)
wg = np.flatnonzero(s[:, 1, 0] > 0)
print(f"workgroups of the last launch: {len(wg)}")
s = s[wg]
us = lambda a: a * 0.01
steps = slice(2, T - 2)
names = ["prologue (first units landed)", "main loop", "epilogue (stores issued)", "store drain + wg barrier", "group barrier (arrive/poll/acquire)"]
tot = us(s[:, 3:T - 1, 0] - s[:, 2:T - 2, 0])
print(f"step period: mean {tot.mean():.2f} us  p10 {np.percentile(tot, 10):.2f}  p90 {np.percentile(tot, 90):.2f}")
for i, nm in enumerate(names):
    d = us(s[:, steps, i + 1] - s[:, steps, i])
    print(f"  {nm:40s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
d = us(s[:, 3:T - 1, 0] - s[:, 2:T - 2, 5])
print(f"  {'barrier passed -> next step start':40s} mean {d.mean():6.2f} us")
skew = us(s[:, steps, 4].reshape(-1, 4, s[:, steps, 4].shape[1]).max(1) - s[:, steps, 4].reshape(-1, 4, s[:, steps, 4].shape[1]).min(1)) if len(wg) % 4 == 0 else None

cyc = s[:, 60, :6]
if (cyc[:, 0] > 0).any():  # second form, timing build: inside stage 8 of step 5 (shader clocks)
    c = cyc[cyc[:, 0] > 0]
    names = ["wait for the stage's loads (vmcnt)", "workgroup barrier", "issue 6 LDS-DMA loads", "12 fragment reads until the first 5 landed", "32 MFMAs (+ staged waits)"]
    for i, nm in enumerate(names):
        d = c[:, i + 1] - c[:, i]
        print(f"  stage: {nm:45s} mean {d.mean():7.0f}  p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f} cycles")

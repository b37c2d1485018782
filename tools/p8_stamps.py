"""Debug only (library built with -DP8_TIMING): per-block timestamps of the 256x256 GEMM kernel (100 MHz wall clock)."""
import ctypes as C, sys
import numpy as np, torch
from diffmm_amd import ops, _lib as L

dev = torch.device("cuda:0")
import os
M, N, nb = 9308, 6710, 3
cases = ((64, "cb"), (1024, "cb"), (1024, "res"), (6720, "l1"))
if len(sys.argv) > 1 and sys.argv[1] == "small":
    M, N, nb = 2048, 2048, 1
    cases = ((1024, "res"), (1024, "cb"))
for K, mode in cases:
    if mode == "l1":
        N = 1024
    Np = (N + 63) // 64 * 64
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nb)]
    B = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nb)]
    R = [torch.randn(M, Np, device=dev)[:, :N] for _ in range(nb)]
    Cc = [torch.empty(M, Np, device=dev)[:, :N] for _ in range(nb)]
    Cb = [torch.empty(M, Np, device=dev, dtype=torch.bfloat16)[:, :N] for _ in range(nb)]
    if mode == "res":
        probs = [dict(A=A[i], B=B[i], out=Cc[i], out_bf16=Cb[i], residual=R[i], e1=0.7, e2=-0.3) for i in range(nb)]
    else:
        probs = [dict(A=A[i], B=B[i], out=None, out_bf16=Cb[i]) for i in range(nb)]
    for _ in range(2):
        ops.gemm_bf16_nt_batched(probs)
    torch.cuda.synchronize()
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros(8 * nblk, dtype=np.uint64)
    rc = L.lib().dmm_debug_p8_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(8 * nblk))
    assert rc == 0
    s = buf.reshape(nblk, 8).astype(np.int64)
    t0 = s[:, 0].min()
    d = lambda a, b: (s[:, b] - s[:, a]) * 0.01  # us
    print(f"K={K} {mode}: blocks {nblk}  span {(s[:, 4].max() - t0) * 0.01:.1f} us")
    for name, a, b in (("setup+prologue", 0, 1), ("main loop", 1, 2), ("epilogue", 2, 3), ("store drain", 3, 4), ("total", 0, 4)):
        x = d(a, b)
        print(f"   {name:15s} mean {x.mean():7.2f}  p10 {np.percentile(x, 10):7.2f}  p90 {np.percentile(x, 90):7.2f} us")
    starts = np.sort((s[:, 0] - t0) * 0.01)
    print("   block starts (us) every 100th:", np.round(starts[::100], 1)[:12])
    if os.environ.get("P8_DUMP"):
        np.save(os.path.join("gpurun_out", f"stamps_{mode}_{K}_{os.environ['P8_DUMP']}.npy"), s)

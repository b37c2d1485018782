"""Wall-clock breakdown (with synchronisation) of the rebuild phase on the bench workload."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffmm_amd import graph, ops

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="tiktok"); ap.add_argument("--T", type=int, default=40)
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--latdim", type=int, default=64); ap.add_argument("--precision", type=int, default=1)
args = ap.parse_args()
coach, h, cfg = bench.make_coach(args, 0, 1)
for _ in range(2):
    coach.trainEpoch()
DE, dd = coach.diff_engine, h.diffusionData
sync = torch.cuda.synchronize


def timed(fn, n=3):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3, r


batches = [idx for _, idx in list(h.diffusionLoader)]
n_edges = int(dd.ptr[-1])
ms, edges = timed(lambda: DE.rebuild_edges(dd.rows, batches, dd.deg_ptr32, n_edges, precision=1))
print(f"rebuild_edges total {ms:.2f} ms")
all_users = torch.cat([b.reshape(-1) for b in batches])
ms, x0 = timed(lambda: dd.rows(all_users)); print(f"  dense_rows {ms:.2f} ms")
ms, views = timed(lambda: DE.generate_views_bf16(x0)); print(f"  generate_views_bf16 (copies + chain) {ms:.2f} ms")
ms, _ = timed(lambda: [DE.cxa[m][:x0.shape[0], :DE.I].copy_(x0) for m in range(DE.M)]); print(f"    of which x0 copies {ms:.2f} ms")
out = [torch.zeros(n_edges, dtype=torch.int32, device=x0.device) for _ in range(DE.M)]
ub = all_users.to(torch.int32)
ms, _ = timed(lambda: [ops.topk_rows(views[m], ub, dd.deg_ptr32, out[m], DE.I) for m in range(DE.M)]); print(f"  topk x{DE.M} {ms:.2f} ms")
gb = coach.graph_builder
ms, adjs = timed(lambda: gb.build_all(edges)); print(f"graph build x{DE.M} {ms:.2f} ms")
J = coach.model.engine
ms, _ = timed(lambda: J.set_adjacency(h.torchBiAdj, adjs)); print(f"set_adjacency {ms:.2f} ms")

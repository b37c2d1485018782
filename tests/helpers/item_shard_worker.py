"""One replica of the item-sharded Denoise (diffmm_amd/engine.py: ItemShard), launched by tests/test_item_shard_gpu.py under
torch.distributed.run (backend gloo: the replicas share the one GPU of the test box) -- or called in-process with world = 1.  Every rank
runs the SAME seeded problem: one diffusion training step without update (loss means + every gradient it holds), two optimiser steps,
then the rebuild of the same users' edges; it writes what it holds to <out>/rank<r>.npz for the test to compare with the plain engine."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

D, H, M, T_, B, TE = 64, 256, 3, 6, 128, 10
HP = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=5)


def problem():
    """Seeded inputs shared by the workers and the test (host tensors): graph, full Denoise state dicts, features, batch, draws."""
    from diffmm_amd import synth
    g = synth.named("mid")
    U, I = g.users, g.items
    gen = torch.Generator().manual_seed(123)
    rn = lambda *s, sc=0.05: torch.randn(*s, generator=gen) * sc
    sds = []
    for _ in range(M):
        sds.append({"emb_layer.weight": rn(TE, TE, sc=0.3), "emb_layer.bias": rn(TE, sc=0.1), "in_layers.0.weight": rn(H, I + TE), "in_layers.0.bias": rn(H),
                    "out_layers.0.weight": rn(I, H), "out_layers.0.bias": rn(I), "gate_layer.weight": rn(D, D, sc=0.1), "gate_layer.bias": rn(D, sc=0.1)})
    feats = [rn(I, D, sc=0.5) for _ in range(M)]
    i_embs = rn(I, D, sc=0.1)
    users = torch.randperm(U, generator=gen)[:B]
    t = [torch.randint(0, T_, (B,), generator=gen, dtype=torch.int32) for _ in range(M)]
    noise = [torch.randn(B, I, generator=gen) for _ in range(M)]
    return g, U, I, sds, feats, i_embs, users, t, noise


def run(dev, shard=None) -> dict:
    """The sequence of calls on one engine: the plain one (shard None) or one replica of the item-sharded one."""
    from diffmm_amd.DataHandler import DiffusionData
    from diffmm_amd.engine import DiffusionEngine
    g, U, I, sds, feats, i_embs, users, t, noise = problem()
    E = DiffusionEngine(I, D, H, TE, M, T_, HP, dev, B, chain_rows=B, item_shard=shard)
    for m in range(M):
        E.load_reference_state(m, sds[m])
    i0, i1 = (shard.i0, shard.i1) if shard is not None else (0, I)
    dd = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
    users_d = users.to(dev)
    rows = dd.csr_rows(users_d)
    f = [x[i0:i1].contiguous().to(dev) for x in feats]
    nz = [x[:, i0:i1].contiguous().to(dev) for x in noise]
    ts = [x.to(dev) for x in t]
    ie = i_embs.to(dev)
    out = {"i0": np.int64(i0), "i1": np.int64(i1)}
    means = E.train_step(rows, f, ie, t=ts, noise=nz, precision=1, update=False, host_means=False)
    out["means"] = means.cpu().numpy()
    for m in range(M):
        for k in E.P[m].names:
            gk = E.P[m].g(k)
            out[f"grad{m}/{k}"] = (gk[:, : E.I + TE] if k == "in_layers.0.weight" else gk).cpu().numpy().copy()
    E.rng_step = 0
    for _ in range(2):
        E.train_step(rows, f, ie, t=ts, noise=nz, precision=1, update=True, host_means=False)
    for m in range(M):
        for k in E.P[m].names:
            pk = E.P[m].view(k)
            out[f"param{m}/{k}"] = (pk[:, : E.I + TE] if k == "in_layers.0.weight" else pk).cpu().numpy().copy()
    edges = E.rebuild_edges(dd.rows, [users_d], dd.deg_ptr32, int(dd.ptr[-1]), precision=1, csr=(dd.ptr, dd.col32, 0))
    E.check_chain_error()
    for m in range(M):
        out[f"edges{m}"] = edges[m].cpu().numpy()
    out["deg_ptr"] = dd.deg_ptr32.cpu().numpy()
    out["users"] = users.numpy()
    return out


def main():
    import torch.distributed as dist
    from diffmm_amd.engine import ItemShard
    outdir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    _, _, I = problem()[:3]
    out = run(dev, ItemShard(rank, world, I))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

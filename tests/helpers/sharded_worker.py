"""One replica of the user-sharded joint engine, launched by tests/test_sharded_gpu.py under torch.distributed.run (backend gloo: the
replicas share the one GPU of the test box).  Runs the SAME seeded problem on every rank: one joint step without update (loss + every
gradient), then two optimiser steps; writes what it holds to <out>/rank<r>.npz for the test to compare with the single-GPU engine."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def problem(fx: str, cl_method: int):
    """Seeded inputs shared by the worker and the test (numpy only)."""
    z = np.load(os.path.join(REPO, "tests", "golden", fx + ".npz"))
    U, I, M, D = int(z["users"]), int(z["items"]), int(z["M"]), int(z["d"])
    rng = np.random.default_rng(11)
    B = 6 if U < 50 else 200
    e = rng.integers(0, len(z["trn_rows"]), size=B)
    users, pos = z["trn_rows"][e].astype(np.int32), z["trn_cols"][e].astype(np.int32)
    neg = rng.integers(0, I, size=B).astype(np.int32)
    noise = [rng.random((U + I, D)).astype(np.float32) for _ in range(3)]
    hp = dict(modal_adj_weight=0.2, residual_weight=0.5, noise_degree=1.5, cross_cl_temp=0.2, cross_cl_rate=0.5, modal_cl_temp=0.5,
              modal_cl_rate=0.01, reg=1e-4, lr=1e-3, cl_method=cl_method, seed=1818)
    return z, U, I, M, D, users, pos, neg, noise, hp


def build_sharded(z, U, I, M, D, hp, dev, rank, world, group=None, B_max=256, vec=None, two_d=None):
    """two_d (or DMM_TEST_JOINT_SHARD=2d): the engine that shards the items as well (diffmm_amd/sharded2d.py)."""
    from diffmm_amd import sharded
    lay = sharded.ShardLayout.make(z["trn_rows"], U, I, world, rank)
    two_d = os.environ.get("DMM_TEST_JOINT_SHARD", "1d") == "2d" if two_d is None else two_d
    if two_d:
        from diffmm_amd.sharded2d import Sharded2DJointEngine as Engine
    else:
        Engine = sharded.ShardedJointEngine
    E = Engine(lay, D, [torch.from_numpy(z[f"feat{m}"]) for m in range(M)], hp, dev, B_max, group=group)
    E.load_reference_state({k[2:]: z[k] for k in z.files if k.startswith("p/")})
    vec = int(os.environ.get("DMM_TEST_SPMM_VEC", "4")) if vec is None else vec  # 8: local blocks planned for bf16 gather operands
    adj = sharded.make_sharded_adjacency(lay, z["trn_rows"], z["trn_cols"], D, dev, vec=vec)
    madj = [sharded.make_sharded_adjacency(lay, z[f"m{m}_rows"], z[f"m{m}_cols"], D, dev, vec=vec) for m in range(M)]
    E.set_adjacency(adj, madj)
    return lay, E


def run_steps(E, lay, users, pos, neg, noise, dev, precision=3):
    """-> dict of numpy arrays: loss parts + gradients of one step (no update), then parameters after two optimiser steps (the second
    one with in-kernel Philox noise through the recorded launch plan)."""
    t = lambda a: torch.from_numpy(a).to(dev)
    loc = [(E.local_rows if hasattr(E, "local_rows") else lay.local_rows)(t(n)) for n in noise]
    E.step(t(users), t(pos), t(neg), noise=loc, precision=precision, update=False)
    out = {"loss/" + k: np.float64(v) for k, v in E.losses().items()}
    for k, v in E.full_grads().items():
        out["grad/" + k] = v.cpu().numpy()
    E.rng_step = 0
    E.step(t(users), t(pos), t(neg), noise=loc, precision=precision, update=True)
    E.step(t(users), t(pos), t(neg), noise=None, precision=precision, update=True)   # planned path, Philox noise
    E.step(t(users[::-1].copy()), t(pos[::-1].copy()), t(neg[::-1].copy()), noise=None, precision=precision, update=True)  # plan replay
    for k, v in E.state_dict().items():
        out["param/" + k] = v.cpu().numpy()
    out["loss3"] = E.loss_parts.cpu().numpy()
    return out


def main():
    fx, cl_method, outdir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group(os.environ.get("DMM_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    z, U, I, M, D, users, pos, neg, noise, hp = problem(fx, cl_method)
    lay, E = build_sharded(z, U, I, M, D, hp, dev, rank, world)
    out = run_steps(E, lay, users, pos, neg, noise, dev)
    out["u0"], out["u1"] = np.int64(lay.u0), np.int64(lay.u1)
    out["i0"], out["i1"] = np.int64(getattr(E, "i0", 0)), np.int64(getattr(E, "i1", I))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

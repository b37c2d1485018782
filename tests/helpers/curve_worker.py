"""Train the HIP path for a number of epochs on the synthetic TikTok-shape graph at the bench configuration (T = 40, batch 4096, the
reference's conf/tiktok.toml otherwise) and return / write the per-epoch metrics.  Used in-process by tests/test_recall_gpu.py and, under
torch.distributed.run (gloo, replicas sharing the test GPU), for the user-sharded path:
    python -m torch.distributed.run --nproc-per-node 2 ... tests/helpers/curve_worker.py <seed> <epochs> <out.csv>"""
import csv
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def train_curve(seed: int, epochs: int, T: int = 40, batch: int = 4096, precision: int = 1, rank: int = 0, world: int = 1):
    from diffmm_amd import synth
    from diffmm_amd.Conf import load_config
    from diffmm_amd.DataHandler import DataHandler
    from diffmm_amd.Main import Coach, seed_it
    cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
    cfg.base.seed, cfg.hyper.steps, cfg.train.batch = seed, T, batch
    cfg.base.gpu = str(torch.cuda.current_device())
    seed_it(seed)
    h = DataHandler(cfg)
    h.LoadData(graph_data=synth.named("tiktok", seed=1818))  # the graph is fixed (tests/golden/make_curve.py); the seed drives the training
    coach = Coach(h, cfg, precision=precision, rank=rank, world=world)
    coach.prepareModel()
    rows = []
    for ep in range(epochs):
        res = coach.trainEpoch()
        tst = coach.testEpoch()
        rows.append(dict(epoch=ep, **{k: float(v) for k, v in res.items()}, **{k: float(v) for k, v in tst.items()}))
    return rows


def main():
    import torch.distributed as dist
    seed, epochs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group(os.environ.get("DMM_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    rows = train_curve(seed, epochs, rank=rank, world=world)
    if rank == 0:
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

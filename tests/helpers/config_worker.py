"""One or more epochs at a BASELINE.json configuration on the synthetic graph of that shape; prints / returns a JSON record.  In-process
(single GPU) from tests/test_configs_gpu.py, or under torch.distributed.run (gloo, replicas sharing the test GPU) for the user-sharded path:
    python -m torch.distributed.run --nproc-per-node 2 ... tests/helpers/config_worker.py <conf> <latdim> <T> <batch> <epochs> <out.json>"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def run_config(conf: str, latdim: int, T: int, batch: int, epochs: int, rank: int = 0, world: int = 1) -> dict:
    from diffmm_amd.Conf import load_config
    from diffmm_amd.DataHandler import DataHandler
    from diffmm_amd.Main import Coach, seed_it
    cfg = load_config(os.path.join(REPO, "conf", conf + ".toml"))
    cfg.base.latdim, cfg.hyper.steps, cfg.train.batch = latdim, T, batch
    cfg.base.gpu = str(torch.cuda.current_device())
    seed_it(cfg.base.seed)
    h = DataHandler(cfg)
    h.LoadData()  # no bundle on the test box: the synthetic graph of the named shape (diffmm_amd/synth.py)
    coach = Coach(h, cfg, precision=1, rank=rank, world=world)
    coach.prepareModel()
    out = {"users": h.graph.users, "items": h.graph.items, "E": h.graph.nnz, "M": len(h.graph.feats), "epochs": []}
    for _ in range(epochs):
        res = coach.trainEpoch()
        tst = coach.testEpoch()
        adjs = [coach.image_adj, coach.text_adj] + ([coach.audio_adj] if coach.M == 3 else [])
        out["epochs"].append({"train": {k: float(v) for k, v in res.items()}, "test": {k: float(v) for k, v in tst.items()},
                              "modal_nnz": [int(a.nnz) for a in adjs], "phase_s": list(coach.phase_seconds)})
    out["sampling_step"] = int(cfg.hyper.sampling_step)
    out["denoise_shard"] = "items" if getattr(coach, "item_shard", None) is not None else "rows"
    out["joint_shard"] = "1d" if getattr(coach.model.engine, "replicated_items", True) else "2d"
    return out


def main():
    import torch.distributed as dist
    conf, latdim, T, batch, epochs, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    dist.init_process_group(os.environ.get("DMM_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    out = run_config(conf, latdim, T, batch, epochs, rank=rank, world=world)
    nnz = torch.tensor(out["epochs"][-1]["modal_nnz"], dtype=torch.int64, device="cuda")
    dist.all_reduce(nnz)  # the replicas' local blocks add up to the full graphs
    out["modal_nnz_sum_over_replicas"] = nnz.tolist()
    if rank == 0:
        json.dump(out, open(path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

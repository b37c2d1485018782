"""Time Coach.testEpoch() on a synthetic graph (tools only; not part of the bench line)."""
import sys, time
import torch
from diffmm_amd.Conf import load_config
from diffmm_amd.Main import Coach
from diffmm_amd.DataHandler import DataHandler

name = sys.argv[1] if len(sys.argv) > 1 else "tiktok"
cfg = load_config("conf/tiktok.toml")
cfg.data.name = name
h = DataHandler(cfg)
h.LoadData()
c = Coach(h, cfg)
c.prepareModel()
c.image_adj = c.text_adj = c.audio_adj = h.torchBiAdj
for _ in range(2):
    r = c.testEpoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    r = c.testEpoch()
torch.cuda.synchronize()
print(f"testEpoch {name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  {r}")

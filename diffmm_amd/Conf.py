"""Configuration surface of the reference (Conf.py:9-77): four sections as dataclasses + load_config(path).

Field names and defaults are the reference's; unknown TOML keys raise TypeError exactly like the reference's
dataclass constructors do (strict, the default).  load_config(path, strict=False) is the tolerant loader of SURVEY 8f.4:
keys the dataclasses do not know -- the reference's own conf/baby.toml and conf/test.toml still carry the key names of an
older code base (trans, keepRate, e_loss, sampling_steps, rebuild_k, norm, sampling_noise) and do not load with its
Conf.py -- are dropped with ONE logged line naming them; nothing is renamed or guessed (a dropped `sampling_steps` leaves
`sampling_step` at its default, which is what the reference's code would run if its loader got that far).
Parsing uses the stdlib-compatible `tomli` (the reference imports `toml`).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field, fields

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover
    import tomli as _toml


@dataclass
class BaseConfig:
    latdim: int = 64
    topk: int = 20
    gpu: str = "0"
    seed: int = 8888
    denoise_dim: str = "[1024]"
    d_emb_size: int = 10
    cl_method: int = 0


@dataclass
class DataConfig:
    name: str = "tiktok"
    user_num: int = 0  # the five fields below are filled in by DataHandler.LoadData()
    item_num: int = 0
    image_feat_dim: int = 0
    text_feat_dim: int = 0
    audio_feat_dim: int = 0


@dataclass
class HyperConfig:
    modal_cl_temp: float = 0.5
    modal_cl_rate: float = 0.01
    cross_cl_temp: float = 0.2
    cross_cl_rate: float = 0.2
    noise_degree: float = 0.2
    noise_scale: float = 0.1
    noise_min: float = 0.0001
    noise_max: float = 0.02
    steps: int = 5
    sim_weight: float = 0.1
    residual_weight: float = 0.5
    modal_adj_weight: float = 0.2
    sampling_step: int = 0
    knn_topk: int = 10


@dataclass
class TrainConfig:
    lr: float = 0.001
    batch: int = 1024
    test_batch: int = 256
    reg: float = 1e-5
    epoch: int = 50
    tstEpoch: int = 1
    gnn_layer: int = 1
    use_lr_scheduler: bool = True


@dataclass
class Config:
    base: BaseConfig = field(default_factory=BaseConfig)
    data: DataConfig = field(default_factory=DataConfig)
    hyper: HyperConfig = field(default_factory=HyperConfig)
    train: TrainConfig = field(default_factory=TrainConfig)


def load_config(path: str, strict: bool = True) -> Config:
    with open(path, "rb") as fh:
        raw = _toml.load(fh)
    sections = {"base": BaseConfig, "data": DataConfig, "hyper": HyperConfig, "train": TrainConfig}
    kept, dropped = {}, []
    for name, cls in sections.items():
        sec = dict(raw.get(name, {}))
        if not strict:
            known = {f.name for f in fields(cls)}
            dropped += [f"{name}.{k}" for k in sec if k not in known]
            sec = {k: v for k, v in sec.items() if k in known}
        kept[name] = cls(**sec)  # strict: an unknown key is a TypeError here, as in the reference (Conf.py:69-77)
    if not strict:
        dropped += [f"[{k}]" for k in raw if k not in sections]
        if dropped:
            logging.getLogger("diffmm_amd.Conf").warning("%s: ignored keys unknown to Conf.py: %s", path, ", ".join(dropped))
    return Config(**kept)

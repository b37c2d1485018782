"""Synthetic user-item graphs and modality features of the shapes BASELINE.json names.

The reference ships its interaction matrices as pickled scipy objects
(/root/reference/Datasets/tiktok/trnMat.pkl, read at DataHandler.py:41-50).  Pickles that ship
inside the reference are never unpickled by this repo (torch.load(weights_only=True) refuses
them), so every test, golden vector and benchmark runs on graphs produced here.  The degree
statistics follow SURVEY.md section 8(d): a bimodal user-degree law (1.1 % "heavy" users with
300..600 interactions holding ~70 % of all edges, everybody else 0..4) and Zipf item popularity.

A planted cluster structure (users prefer items of their own cluster, features are a noisy
cluster code) makes Recall@20 a meaningful, learnable quantity instead of noise.

numpy only; vectorised so the 10M-user configuration can be generated shard by shard.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

# light-user degree histogram measured on the TikTok train matrix
_LIGHT_DEG = np.array([0, 1, 2, 3, 4])
_LIGHT_P = np.array([0.001, 0.377, 0.330, 0.277, 0.004])

SHAPES: Dict[str, dict] = {
    # name: users, items, feature widths, modality names
    "tiktok": dict(users=9308, items=6710, feats=(128, 768, 128), heavy_p=0.011, heavy=(300, 520)),
    "baby": dict(users=19445, items=7050, feats=(4096, 1024), heavy_p=0.0, heavy=(0, 0), light_shift=3),
    "sports": dict(users=35598, items=18357, feats=(4096, 1024), heavy_p=0.0, heavy=(0, 0), light_shift=3),
    "toy": dict(users=12, items=9, feats=(5, 7, 6), heavy_p=0.1, heavy=(5, 7)),
    "mid": dict(users=300, items=200, feats=(32, 48, 16), heavy_p=0.02, heavy=(60, 120)),
    "mid2": dict(users=2000, items=1200, feats=(64, 96, 32), heavy_p=0.011, heavy=(80, 160)),  # the paired 12-epoch golden (tests/golden/epoch_paired.npz)
}


@dataclasses.dataclass
class SynthGraph:
    users: int
    items: int
    trn_rows: np.ndarray  # int32, sorted by (row, col), de-duplicated
    trn_cols: np.ndarray  # int32
    tst_rows: np.ndarray  # int32, disjoint from train
    tst_cols: np.ndarray  # int32
    feats: List[np.ndarray]  # per modality, float32 [items, F_m]
    clusters: Optional[Tuple[np.ndarray, np.ndarray]] = None  # (user_cluster [user_hi - user_lo], item_cluster [items]) of the generator: tools only

    @property
    def nnz(self) -> int:
        return int(self.trn_rows.shape[0])


def _zipf_cdf(n: int, a: float = 1.0) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), a)
    c = np.cumsum(w)
    return c / c[-1]


def _dedup(rows: np.ndarray, cols: np.ndarray, items: int) -> Tuple[np.ndarray, np.ndarray]:
    key = rows.astype(np.int64) * items + cols.astype(np.int64)
    key = np.unique(key)
    return (key // items).astype(np.int32), (key % items).astype(np.int32)


def make_graph(users: int, items: int, feats: Tuple[int, ...] = (128, 768, 128), seed: int = 1818,
               heavy_p: float = 0.011, heavy: Tuple[int, int] = (300, 600), light_shift: int = 0,
               clusters: Optional[int] = None, affinity: float = 0.6, test_frac: float = 0.12,
               feat_noise: float = 1.0, user_lo: int = 0, user_hi: Optional[int] = None) -> SynthGraph:
    """Generate one graph (or the user range [user_lo, user_hi) of it for sharded generation).

    Deterministic in (seed, users, items, ...): the per-user draws are keyed on the user id so that
    a shard generated alone equals the same rows of the full graph.
    """
    user_hi = users if user_hi is None else user_hi
    n_u = user_hi - user_lo
    K = clusters if clusters is not None else max(2, min(16, items // 8))
    g_rng = np.random.default_rng(seed)
    # ---- global (shard independent) structure ------------------------------------------------
    item_cluster = g_rng.integers(0, K, size=items)
    pop_rank = g_rng.permutation(items)  # popularity rank -> item id
    global_cdf = _zipf_cdf(items)
    # items of each cluster ordered by global popularity
    order = np.argsort(item_cluster[pop_rank], kind="stable")
    clustered = pop_rank[order]  # item ids grouped by cluster, popular first inside a group
    csize = np.bincount(item_cluster, minlength=K)
    cstart = np.concatenate([[0], np.cumsum(csize)[:-1]])
    codes = g_rng.standard_normal((K, 16)).astype(np.float32)
    feat_list = []
    for fm in feats:
        proj = g_rng.standard_normal((16, fm)).astype(np.float32) / 4.0
        f = codes[item_cluster] @ proj + feat_noise * g_rng.standard_normal((items, fm)).astype(np.float32)
        feat_list.append(np.ascontiguousarray(f, dtype=np.float32))
    # ---- per-user draws -------------------------------------------------------------------------
    u_rng = np.random.default_rng([seed, 7, user_lo])
    is_heavy = u_rng.random(n_u) < heavy_p
    light = u_rng.choice(_LIGHT_DEG, size=n_u, p=_LIGHT_P / _LIGHT_P.sum()) + light_shift
    hv = u_rng.integers(heavy[0], heavy[1] + 1, size=n_u) if heavy[1] > 0 else np.zeros(n_u, np.int64)
    deg = np.where(is_heavy, hv, light).astype(np.int64)
    deg = np.minimum(deg, max(1, items // 2))
    n_tst = np.maximum((deg * test_frac).astype(np.int64), (deg > 0).astype(np.int64))  # >=1 test edge for active users
    tot = deg + n_tst
    user_cluster = u_rng.integers(0, K, size=n_u)
    # over-sample (Zipf draws collide), keep the first `tot` distinct items of every user in draw order
    draws = (tot * 3 + 4) * (tot > 0)
    owner = np.repeat(np.arange(n_u, dtype=np.int64), draws)
    m = owner.shape[0]
    own = u_rng.random(m) < affinity
    r = u_rng.random(m)
    g_item = pop_rank[np.minimum(np.searchsorted(global_cdf, r), items - 1)]
    uc = user_cluster[owner]
    sz = np.maximum(csize[uc], 1)
    # inverse-CDF of a truncated Zipf(1) inside the user's cluster: exp(r * log(sz+1)) - 1
    pos = np.minimum((np.exp(r * np.log(sz + 1.0)) - 1.0).astype(np.int64), sz - 1)
    c_item = clustered[np.minimum(cstart[uc] + pos, items - 1)]
    item = np.where(own & (csize[uc] > 0), c_item, g_item).astype(np.int64)
    first = np.sort(np.unique(owner * items + item, return_index=True)[1])  # first occurrence of each pair
    owner_k, item_k = owner[first], item[first]
    rank = np.arange(first.shape[0], dtype=np.int64) - np.searchsorted(owner_k, owner_k, side="left")
    is_trn = rank < deg[owner_k]
    is_tst = (~is_trn) & (rank < tot[owner_k])
    rows_g = owner_k + user_lo
    tr, tc = _dedup(rows_g[is_trn], item_k[is_trn], items)
    sr, sc = _dedup(rows_g[is_tst], item_k[is_tst], items)
    keep = np.ones(sr.shape[0], dtype=bool)
    return SynthGraph(users, items, tr, tc, sr[keep], sc[keep], feat_list, clusters=(user_cluster, item_cluster))


def named(name: str, seed: int = 1818, **over) -> SynthGraph:
    spec = dict(SHAPES[name])
    spec.update(over)
    return make_graph(spec["users"], spec["items"], tuple(spec["feats"]), seed=seed,
                      heavy_p=spec.get("heavy_p", 0.011), heavy=tuple(spec.get("heavy", (300, 600))),
                      light_shift=spec.get("light_shift", 0))


def neg_sample(rows: np.ndarray, cols: np.ndarray, items: int, rng: np.random.Generator) -> np.ndarray:
    """Vectorised restatement of TrainData.negSampling (DataHandler.py:159-169): one uniform item per
    interaction, re-drawn until (user, item) is not a train interaction."""
    key = np.sort(rows.astype(np.int64) * items + cols.astype(np.int64))
    neg = rng.integers(0, items, size=rows.shape[0])
    bad = np.ones(rows.shape[0], dtype=bool)
    while True:
        k = rows[bad].astype(np.int64) * items + neg[bad]
        p = np.searchsorted(key, k)
        hit = (p < key.shape[0]) & (key[np.minimum(p, key.shape[0] - 1)] == k)
        idx = np.flatnonzero(bad)[hit]
        if idx.size == 0:
            break
        neg[idx] = rng.integers(0, items, size=idx.size)
        bad[:] = False
        bad[idx] = True
    return neg.astype(np.int32)

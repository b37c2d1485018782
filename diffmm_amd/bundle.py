"""On-disk data format.  The reference ships interactions as pickled scipy matrices (DataHandler.py:41-50) and features as
`.npy` arrays (DataHandler.py:95-105, fp16 on disk).  Pickles are never read here -- unpickling executes code; the bundle below is a plain
`.npz` (numpy arrays only, loads with allow_pickle=False) that DataHandler(config).LoadData() picks up from
`Datasets/<name>/<name>.npz`:

    users, items                int64 scalars
    trn_rows, trn_cols          int32 [E]     train interactions (user, item), duplicates allowed (binarised on load)
    tst_rows, tst_cols          int32 [E_t]   test interactions
    feat0, feat1[, feat2]       float16/float32 [items, F_m]   modality features in the order image, text[, audio]

    python -m diffmm_amd.bundle --synth tiktok --out Datasets/tiktok/tiktok.npz                 # the synthetic graph of a named shape
    python -m diffmm_amd.bundle --edges trn.txt --test-edges tst.txt --features image_feat.npy text_feat.npy audio_feat.npy --out ...
(`--edges`: text file, one "user item" pair per line -- e.g. exported once, outside this code base, from the reference's matrices.)
"""
from __future__ import annotations

import argparse
import os
from typing import List, Optional, Sequence

import numpy as np

from . import synth


def load_feature_npy(path: str) -> np.ndarray:
    """One of the reference's feature files (`image_feat.npy`, ...: DataHandler.py:95-105) as float32 [items, F].  Loaded with
    allow_pickle=False: an object array is refused instead of unpickled."""
    a = np.load(path, allow_pickle=False)
    if a.ndim != 2 or a.dtype.kind != "f":
        raise ValueError(f"{path}: expected a 2-D floating point array, got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a, dtype=np.float32)


def write_bundle(path: str, users: int, items: int, trn_rows, trn_cols, tst_rows, tst_cols, feats: Sequence[np.ndarray], feat_dtype=np.float16) -> str:
    """Write one data set.  Features are stored in `feat_dtype` (the reference's files are fp16; pass np.float32 to keep full precision)."""
    tr, tc = np.asarray(trn_rows, dtype=np.int32), np.asarray(trn_cols, dtype=np.int32)
    sr, sc = np.asarray(tst_rows, dtype=np.int32), np.asarray(tst_cols, dtype=np.int32)
    for nm, r, c in (("train", tr, tc), ("test", sr, sc)):
        if r.shape != c.shape or r.ndim != 1:
            raise ValueError(f"{nm} edges: rows and cols must be 1-D arrays of one length")
        if r.size and (r.min() < 0 or r.max() >= users or c.min() < 0 or c.max() >= items):
            raise ValueError(f"{nm} edges out of range for {users} users x {items} items")
    if not 2 <= len(feats) <= 3:
        raise ValueError("2 or 3 modality feature matrices (image, text[, audio])")
    arrays = dict(users=np.int64(users), items=np.int64(items), trn_rows=tr, trn_cols=tc, tst_rows=sr, tst_cols=sc)
    for m, f in enumerate(feats):
        f = np.asarray(f)
        if f.ndim != 2 or f.shape[0] != items:
            raise ValueError(f"feature matrix {m}: expected [{items}, F], got {f.shape}")
        arrays[f"feat{m}"] = np.ascontiguousarray(f, dtype=feat_dtype)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, **arrays)
    return path if path.endswith(".npz") else path + ".npz"


def read_bundle(path: str) -> synth.SynthGraph:
    z = np.load(path, allow_pickle=False)
    missing = [k for k in ("users", "items", "trn_rows", "trn_cols", "tst_rows", "tst_cols", "feat0", "feat1") if k not in z.files]
    if missing:
        raise ValueError(f"{path}: not a data bundle (missing {missing})")
    feats = [np.asarray(z[k], dtype=np.float32) for k in sorted(f for f in z.files if f.startswith("feat"))]
    return synth.SynthGraph(int(z["users"]), int(z["items"]), np.asarray(z["trn_rows"], dtype=np.int32), np.asarray(z["trn_cols"], dtype=np.int32),
                            np.asarray(z["tst_rows"], dtype=np.int32), np.asarray(z["tst_cols"], dtype=np.int32), feats)


def _read_edges(path: str):
    e = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if e.shape[1] != 2:
        raise ValueError(f"{path}: expected two integer columns (user item)")
    return e[:, 0], e[:, 1]


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="write a DiffMM data bundle (.npz)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--synth", default=None, help=f"synthetic graph of a named shape: {sorted(synth.SHAPES)}")
    ap.add_argument("--seed", type=int, default=1818)
    ap.add_argument("--edges", default=None); ap.add_argument("--test-edges", default=None)
    ap.add_argument("--features", nargs="+", default=None, help="image text [audio] .npy files ([items, F] float arrays)")
    ap.add_argument("--users", type=int, default=None); ap.add_argument("--items", type=int, default=None)
    a = ap.parse_args(argv)
    if a.synth:
        g = synth.named(a.synth, seed=a.seed)
        print(write_bundle(a.out, g.users, g.items, g.trn_rows, g.trn_cols, g.tst_rows, g.tst_cols, g.feats))
        return
    if not (a.edges and a.test_edges and a.features):
        ap.error("give --synth NAME, or --edges, --test-edges and --features")
    tr, tc = _read_edges(a.edges)
    sr, sc = _read_edges(a.test_edges)
    feats = [load_feature_npy(p) for p in a.features]
    items = a.items or feats[0].shape[0]
    users = a.users or int(max(tr.max(), sr.max()) + 1)
    print(write_bundle(a.out, users, items, tr, tc, sr, sc, feats))


if __name__ == "__main__":
    main()

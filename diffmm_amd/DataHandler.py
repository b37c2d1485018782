"""Data side of the reference's surface (DataHandler.py:10-228), re-designed for HBM residency.

Same names and call order as the reference -- DataHandler(config).LoadData(), .trainData/.trainLoader/.testData/
.testLoader/.diffusionData/.diffusionLoader/.torchBiAdj/.image_feats ..., static normalizeAdj / makeTorchAdj,
getUserDegrees(), TrainData.negSampling() -- but:
  * interactions come from a pickle-free bundle (`Datasets/<name>/<name>.npz` with trn_rows, trn_cols, tst_rows,
    tst_cols, feat0..featM-1) or, when none is present, from the synthetic generator of the named shape
    (diffmm_amd/synth.py).  The reference's pickles are never read (they cannot be loaded safely);
  * makeTorchAdj returns a CSR `Adjacency` with its SpMM task list instead of an uncoalesced torch COO tensor;
  * the loaders hand out int32 index tensors that already live on the device; negative sampling is vectorised
    (the reference's python rejection loop, DataHandler.py:159-169, is O(E) interpreter steps);
  * DiffusionData materialises only the requested user rows (the reference keeps a dense U x I matrix in HBM,
    DataHandler.py:128 -- impossible at the 10M x 1M configuration).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import graph, synth
from .Conf import Config


def _config_device(config) -> torch.device:
    """The device the reference's classes derive from the config (DataHandler.py:16,216)."""
    return torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")


def _is_sparse(x) -> bool:
    return hasattr(x, "tocoo") and hasattr(x, "shape")


def _coo_pairs(mat) -> Tuple[np.ndarray, np.ndarray]:
    """(row, col) of the stored non-zero entries of a scipy sparse matrix, in COO order (what the reference iterates: DataHandler.py:150-157,190-199)."""
    m = mat.tocoo()
    keep = m.data != 0
    return np.asarray(m.row[keep], dtype=np.int32), np.asarray(m.col[keep], dtype=np.int32)


class BatchLoader:
    """Minimal stand-in for torch DataLoader(shuffle=..., num_workers=0): index batches drawn on the device."""

    def __init__(self, dataset, batch_size: int, shuffle: bool, device, seed: int = 0):
        self.dataset, self.batch_size, self.shuffle, self.device = dataset, batch_size, shuffle, device
        self.fixed_perm: Optional[torch.Tensor] = None  # tests pin the order through this
        # own generator: data-parallel replicas must draw identical permutations whatever else consumed the global RNG
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def index_batches(self) -> List[torch.Tensor]:
        """The epoch's batches as index tensors only (one permutation draw, nothing materialised): what the rebuild and the user-sharded
        trainer consume -- they densify / slice only the rows they own."""
        n = len(self.dataset)
        if self.fixed_perm is not None:
            perm = self.fixed_perm.to(self.device)
        elif self.shuffle:
            perm = torch.randperm(n, device=self.device, generator=self.gen)
        else:
            perm = torch.arange(n, device=self.device)
        return [perm[s:s + self.batch_size] for s in range(0, n, self.batch_size)]

    def _perm(self) -> torch.Tensor:
        n = len(self.dataset)
        if self.fixed_perm is not None:
            return self.fixed_perm.to(self.device)
        if self.shuffle:
            return torch.randperm(n, device=self.device, generator=self.gen)
        return torch.arange(n, device=self.device)

    def prefetch(self) -> bool:
        """Draw the next epoch's permutation and gather its records NOW, on the current stream (Coach.trainEpoch: a side stream under the rebuild's
        reverse chain -- a dozen small sort / gather launches that otherwise sit between the rebuild and the first joint step); the next __iter__
        hands them out.  The same draw from the same generator as __iter__ would make.  False: this dataset has no whole-epoch gather."""
        if not hasattr(self.dataset, "epoch"):
            return False
        self._prefetched = self.dataset.epoch(self._perm())
        return True

    def __iter__(self) -> Iterator:
        n = len(self.dataset)
        pre = getattr(self, "_prefetched", None)
        self._prefetched = None
        perm = self._perm() if pre is None else None
        if hasattr(self.dataset, "epoch"):  # gather the whole epoch once (3 launches), then hand out contiguous slices: no per-batch index kernels
            cols = self.dataset.epoch(perm) if pre is None else pre
            for s in range(0, n, self.batch_size):
                yield tuple(c[s:s + self.batch_size] for c in cols)
            return
        for s in range(0, n, self.batch_size):
            yield self.dataset.batch(perm[s:s + self.batch_size])


class TrainData:
    """(user, positive item, sampled negative item) triples -- one per train interaction."""

    def __init__(self, rows, cols=None, config: Optional[Config] = None, device=None):
        """Two positional forms: the reference's `TrainData(coomat, config)` (DataHandler.py:147: a scipy (user x item) COO matrix; the
        device comes from the config like everywhere in the reference) and `TrainData(rows, cols, config, device)` with index arrays (what
        LoadData() passes: no scipy object is needed for a synthetic or bundled graph)."""
        if _is_sparse(rows):
            coomat, config = rows, (cols if config is None else config)
            if not isinstance(config, Config):
                raise TypeError("TrainData(coomat, config): config must be a Conf.Config")
            rows, cols = _coo_pairs(coomat)
            device = _config_device(config) if device is None else device
        elif config is None:
            raise TypeError("TrainData(rows, cols, config[, device]) or TrainData(coomat, config)")
        self.config, self.device = config, torch.device("cpu" if device is None else device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(config.base.seed) + 17)
        self.rows, self.cols = np.asarray(rows, dtype=np.int32), np.asarray(cols, dtype=np.int32)
        self._negs_host: Optional[np.ndarray] = np.zeros(len(self.rows), dtype=np.int32)
        items = int(config.data.item_num)
        self._keys = torch.from_numpy(np.sort(self.rows.astype(np.int64) * items + self.cols)).to(self.device)
        self._rows_d = torch.from_numpy(self.rows).to(self.device)
        self._cols_d = torch.from_numpy(self.cols).to(self.device)
        self._negs_d = torch.zeros(len(self.rows), dtype=torch.int32, device=self.device)
        self._rows64 = None  # user id * item_num as int64 keys (device), built by the first negSampling()
        self._pos_ptr = self._pos_col = None  # CSR of the users' sorted train positives (device), built by the first negSampling() on the GPU
        self.neg_calls = 0  # negSampling() calls so far: the Philox offset of the next draw (the GPU sampler's whole state)
        self._user_pos_items: Optional[List[np.ndarray]] = None

    @property
    def user_pos_items(self) -> List[np.ndarray]:
        """Per-user arrays of train items (the reference's attribute, DataHandler.py:152-157); built on first access -- nothing on the training path
        reads it, and a Python list per user is minutes of host time at 10M users."""
        if self._user_pos_items is None:
            order = np.argsort(self.rows, kind="stable")
            ptr = np.searchsorted(self.rows[order], np.arange(int(self.config.data.user_num) + 1))
            self._user_pos_items = [self.cols[order[ptr[u]:ptr[u + 1]]] for u in range(int(self.config.data.user_num))]
        return self._user_pos_items

    @property
    def negs(self) -> np.ndarray:
        """Host mirror of the current negatives (read back on first use after a negSampling(): the training path never needs it)."""
        if self._negs_host is None:
            self._negs_host = self._negs_d.cpu().numpy()
        return self._negs_host

    def negSampling(self) -> None:
        """One uniform negative per interaction that is not a train interaction of its user (DataHandler.py:164-176).  On the GPU: dmm_neg_sample --
        the r-th non-positive item of the user for a uniform r, the distribution the reference's rejection loop samples, in ONE launch with no
        host round trip (the rejection rounds below needed a device -> host check per four rounds: a drained queue at the start of every epoch).
        Draws are Philox(seed, call counter): `neg_calls` is the sampler's whole state."""
        items = int(self.config.data.item_num)
        n = len(self.rows)
        if self._rows_d.is_cuda and n:
            from . import _lib as L
            if self._pos_ptr is None:  # CSR of the sorted train positives per user (from the sorted keys user * items + item)
                U = int(self.config.data.user_num)
                uk = torch.unique(self._keys)  # (sorted; duplicate interactions count once, as in the reference's dok matrix)
                self._pos_ptr = torch.searchsorted(uk, torch.arange(U + 1, device=self.device, dtype=torch.int64) * items).contiguous()
                self._pos_col = (uk % items).to(torch.int32).contiguous()
            out = torch.empty(n, dtype=torch.int32, device=self.device)
            L.call("dmm_neg_sample", L.ptr(self._pos_ptr), L.ptr(self._pos_col), L.ptr(self._rows_d), n, items, int(self.config.base.seed) + 17, self.neg_calls * n,
                   L.ptr(out), L.stream())
            self.neg_calls += 1
            self._negs_d, self._negs_host = out, None
            return
        if self._rows64 is None:
            self._rows64 = self._rows_d.long() * items
        last = self._keys.numel() - 1

        def is_positive(neg):
            key = self._rows64 + neg
            pos = torch.searchsorted(self._keys, key).clamp_(max=last)
            return self._keys[pos] == key

        neg = torch.randint(0, items, (n,), device=self.device, generator=self.gen)
        while True:
            for _ in range(4):
                neg = torch.where(is_positive(neg), torch.randint(0, items, (n,), device=self.device, generator=self.gen), neg)
            if not bool(is_positive(neg).any()):
                break
        self._negs_d = neg.to(torch.int32)
        self._negs_host = None

    def set_negs(self, negs: np.ndarray) -> None:
        self._negs_host = np.asarray(negs, dtype=np.int32)
        self._negs_d = torch.from_numpy(self._negs_host).to(self.device)

    def __len__(self) -> int:
        return len(self.rows)

    def __getitem__(self, idx):
        return self.rows[idx], self.cols[idx], self.negs[idx]

    def batch(self, idx: torch.Tensor):
        return self._rows_d[idx].contiguous(), self._cols_d[idx].contiguous(), self._negs_d[idx].contiguous()

    def epoch(self, perm: torch.Tensor):
        """The whole shuffled epoch: (users, positives, negatives) gathered once into the rows of ONE [3, E] tensor; the loader slices
        minibatches out of them (and the engine, seeing three equally spaced views of one buffer, moves a minibatch with one copy)."""
        rec = torch.stack([self._rows_d[perm], self._cols_d[perm], self._negs_d[perm]])
        return rec[0], rec[1], rec[2]


class TestData:
    def __init__(self, tst_rows, tst_cols, trn_rows=None, trn_cols=None, users: Optional[int] = None, items: Optional[int] = None, device=None):
        """Two positional forms: the reference's `TestData(testMat, trainMat)` (DataHandler.py:180: scipy (user x item) matrices; the reference's class carries no
        device -- its index arrays go to the current GPU when there is one, where testEpoch() ranks, unless `device=` says otherwise) and `TestData(tst_rows, tst_cols, trn_rows, trn_cols, U, I, device)`."""
        if _is_sparse(tst_rows):
            testMat, trainMat = tst_rows, tst_cols
            if not _is_sparse(trainMat):
                raise TypeError("TestData(testMat, trainMat): both must be scipy sparse matrices")
            if trn_rows is not None and device is None and not isinstance(trn_rows, np.ndarray):
                device = trn_rows  # TestData(testMat, trainMat, device)
            users, items = int(testMat.shape[0]), int(testMat.shape[1])
            tst_rows, tst_cols = _coo_pairs(testMat)
            trn_rows, trn_cols = _coo_pairs(trainMat)
            if device is None and torch.cuda.is_available():
                device = torch.device("cuda", torch.cuda.current_device())
        elif trn_rows is None or trn_cols is None or users is None or items is None:
            raise TypeError("TestData(tst_rows, tst_cols, trn_rows, trn_cols, users, items[, device]) or TestData(testMat, trainMat)")
        self.device = torch.device("cpu" if device is None else device)
        self.users, self.items = users, items
        order = np.argsort(tst_rows, kind="stable")
        r, c = np.asarray(tst_rows)[order], np.asarray(tst_cols)[order]
        self.test_users = np.unique(r)
        ptr = np.searchsorted(r, np.arange(users + 1))
        self._its_src, self._test_user_its = (c, ptr), None
        self._tst_ptr = torch.from_numpy(ptr.astype(np.int64)).to(self.device)
        self._tst_col = torch.from_numpy(c.astype(np.int64)).to(self.device)
        # train interactions as CSR over users, for the score mask
        o2 = np.argsort(trn_rows, kind="stable")
        self._trn_ptr = torch.from_numpy(np.searchsorted(np.asarray(trn_rows)[o2], np.arange(users + 1)).astype(np.int64)).to(self.device)
        self._trn_col = torch.from_numpy(np.asarray(trn_cols)[o2].astype(np.int64)).to(self.device)
        self._users_d = torch.from_numpy(self.test_users.astype(np.int64)).to(self.device)
        self._users_i32 = self._users_d.to(torch.int32)

    @property
    def test_user_its(self) -> List[Optional[list]]:
        """Per-user lists of test items (the reference's attribute, DataHandler.py:186-199), built on first access: the evaluation reads the CSR."""
        if self._test_user_its is None:
            c, ptr = self._its_src
            self._test_user_its = [list(c[ptr[u]:ptr[u + 1]]) if ptr[u + 1] > ptr[u] else None for u in range(self.users)]
        return self._test_user_its

    def __len__(self) -> int:
        return len(self.test_users)

    def batch(self, idx: torch.Tensor):
        usr = self._users_d[idx]
        return usr, dense_rows(self._trn_ptr, self._trn_col, usr, self.items)

    def __getitem__(self, idx):
        u = int(self.test_users[idx])
        mask = np.zeros(self.items, dtype=np.float32)
        mask[self._trn_col[self._trn_ptr[u]:self._trn_ptr[u + 1]].cpu().numpy()] = 1.0
        return u, mask


def dense_rows(ptr: torch.Tensor, col: torch.Tensor, users: torch.Tensor, items: int, ld: Optional[int] = None) -> torch.Tensor:
    """0/1 rows of the interaction matrix for the given users, built from its CSR: dmm_csr_dense_rows on the GPU (int32 column ids);
    index plumbing in torch for host tensors (CPU-side tests)."""
    users = users.long()
    if ptr.is_cuda and col.dtype == torch.int32:
        from . import ops
        width = (items + 3) & ~3 if ld is None else ld
        out = torch.empty((users.shape[0], width), dtype=torch.float32, device=ptr.device)
        ops.csr_dense_rows(ptr, col, users.contiguous(), items, out)
        return out if width == items else out[:, :items]
    start, cnt = ptr[users], ptr[users + 1] - ptr[users]
    total = int(cnt.sum())
    out = torch.zeros((users.shape[0], ld or items), dtype=torch.float32, device=ptr.device)
    if total:
        b = torch.repeat_interleave(torch.arange(users.shape[0], device=ptr.device), cnt, output_size=total)
        first = torch.cumsum(cnt, 0) - cnt
        pos = torch.arange(total, device=ptr.device) - first[b] + start[b]
        out[b, col[pos]] = 1.0
    return out


class DiffusionData:
    """User rows of the train interaction matrix (what the Denoise MLP sees)."""

    def __init__(self, trn_rows, trn_cols=None, users: Optional[int] = None, items: Optional[int] = None, device=None):
        """Two positional forms: the reference's `DiffusionData(data, config)` (DataHandler.py:213: `data` the dense (user x item) 0/1 tensor -- a
        scipy matrix is taken too --, the device from the config) and `DiffusionData(trn_rows, trn_cols, U, I, device)`.  Either way only the
        CSR of the rows is kept: the dense matrix is consumed, never stored (`.data` rebuilds it on request for code that reads the attribute)."""
        if isinstance(trn_cols, Config) or (trn_cols is None and (torch.is_tensor(trn_rows) or _is_sparse(trn_rows))):
            data, config = trn_rows, trn_cols
            if _is_sparse(data):
                users, items = int(data.shape[0]), int(data.shape[1])
                trn_rows, trn_cols = _coo_pairs(data)
            else:
                data = torch.as_tensor(data)
                if data.dim() != 2:
                    raise ValueError("DiffusionData(data, config): data must be a (user_num, item_num) matrix")
                users, items = int(data.shape[0]), int(data.shape[1])
                nz = (data != 0).nonzero()  # row-major order: already sorted by user
                trn_rows, trn_cols = nz[:, 0].cpu().numpy().astype(np.int32), nz[:, 1].cpu().numpy().astype(np.int32)
            if device is None:
                device = _config_device(config) if config is not None else (data.device if torch.is_tensor(data) else "cpu")
        elif users is None or items is None:
            raise TypeError("DiffusionData(trn_rows, trn_cols, users, items[, device]) or DiffusionData(data, config)")
        self.device = torch.device("cpu" if device is None else device)
        self.users, self.items = users, items
        o = np.argsort(trn_rows, kind="stable")
        self.ptr = torch.from_numpy(np.searchsorted(np.asarray(trn_rows)[o], np.arange(users + 1)).astype(np.int64)).to(self.device)
        self.col = torch.from_numpy(np.asarray(trn_cols)[o].astype(np.int64)).to(self.device)
        self.col32 = self.col.to(torch.int32)
        self.deg_ptr32 = self.ptr.to(torch.int32)

    def __len__(self) -> int:
        return self.users

    @property
    def data(self) -> torch.Tensor:
        """The reference's attribute (DataHandler.py:215): the dense (user_num, item_num) matrix, materialised on request only."""
        return self.rows(torch.arange(self.users, device=self.device))

    def rows(self, idx: torch.Tensor) -> torch.Tensor:
        return dense_rows(self.ptr, self.col32 if self.ptr.is_cuda else self.col, idx, self.items)

    def csr_rows(self, idx: torch.Tensor):
        """The same rows without densifying them: what the engines take as x_0 (engine.CsrRows)."""
        from .engine import CsrRows
        return CsrRows(self.ptr, self.col32, idx)

    def batch(self, idx: torch.Tensor):
        return self.rows(idx), idx

    def __getitem__(self, index):
        i = torch.as_tensor([index], device=self.device)
        return self.rows(i)[0], index


class DataHandler:
    def __init__(self, config: Config):
        self.config = config
        self.device = torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")
        self.predir = f"./Datasets/{config.data.name}/"
        self.bundle = os.path.join(self.predir, f"{config.data.name}.npz")

    # -- reference statics ---------------------------------------------------------------------------
    @staticmethod
    def normalizeAdj(mat):
        """D^-1/2 A D^-1/2 of a scipy sparse matrix (DataHandler.py:52-66), host side, for API compatibility."""
        import scipy.sparse as sp
        csr = mat.tocsr()
        deg = np.asarray(csr.sum(axis=1)).reshape(-1)
        dinv = np.zeros_like(deg, dtype=np.float64)
        dinv[deg > 0] = deg[deg > 0] ** -0.5
        d = sp.diags(dinv)
        return (d @ csr @ d).tocoo()

    @staticmethod
    def makeTorchAdj(mat, rows: int, cols: int, device, latdim: int = 64) -> graph.Adjacency:
        """(user x item) scipy matrix (or (row, col) arrays) -> normalised bipartite adjacency in HBM (DataHandler.py:68-93)."""
        if isinstance(mat, tuple):
            r, c = mat
        else:
            m = mat.tocoo()
            keep = m.data != 0
            r, c = m.row[keep], m.col[keep]
        return graph.make_adjacency(r, c, rows, cols, latdim, device)

    # -- loading ---------------------------------------------------------------------------------------
    def _read(self) -> synth.SynthGraph:
        if os.path.exists(self.bundle):
            from .bundle import read_bundle
            return read_bundle(self.bundle)
        if self.config.data.name in synth.SHAPES:
            return synth.named(self.config.data.name, seed=int(self.config.base.seed))
        raise ValueError(f"Unknown dataset: {self.config.data.name} (no {self.bundle} and no synthetic shape of that name)")

    def LoadData(self, graph_data: Optional[synth.SynthGraph] = None, sharded: bool = False) -> None:
        """sharded = True (one process per GPU, users sharded): the full-graph adjacency is not built -- every replica plans its own local block
        (Coach.prepareModel).  The interaction LIST stays replicated in a real run: every replica draws the same global shuffle of it (the
        reference's minibatches, Main.py:292-296), and a replica of an item-sharded Denoise reads every minibatch user's row over its item slice
        (768 MB at 10M users).  Only the single-process replica emulation (tools/scale_epoch.py --emulate-world) passes a `graph_data` that holds
        the own users' interactions alone (synth.make_graph(user_lo=, user_hi=)): the rows of the other users are then empty."""
        import scipy.sparse as sp
        g = graph_data if graph_data is not None else self._read()
        cfg, dev = self.config, self.device
        U, I = g.users, g.items
        cfg.data.user_num, cfg.data.item_num = U, I
        self.graph = g
        self.trainMat = sp.coo_matrix((np.ones(g.nnz, np.float32), (g.trn_rows, g.trn_cols)), shape=(U, I))
        # bf16 gather operands for graphs whose operand outgrows the caches (graph.spmm_operand_vec; the user-sharded engine plans its own
        # local blocks, fp32)
        self.spmm_vec = graph.spmm_operand_vec(U + I, int(cfg.base.latdim))
        self.torchBiAdj = None if sharded else graph.make_adjacency(g.trn_rows, g.trn_cols, U, I, int(cfg.base.latdim), dev, vec=self.spmm_vec,
                                                                    task_len=graph.spmm_task_len(self.spmm_vec), seg_mult=graph.BASE_SEG_MULT)
        self.trainData = TrainData(g.trn_rows, g.trn_cols, cfg, dev)
        self.trainLoader = BatchLoader(self.trainData, int(cfg.train.batch), True, dev, seed=int(cfg.base.seed) + 1)
        self.testData = TestData(g.tst_rows, g.tst_cols, g.trn_rows, g.trn_cols, U, I, dev)
        self.testLoader = BatchLoader(self.testData, int(cfg.train.test_batch), False, dev)
        f = [torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dev) for x in g.feats]
        self.image_feats, cfg.data.image_feat_dim = f[0], int(f[0].shape[1])
        self.text_feats, cfg.data.text_feat_dim = f[1], int(f[1].shape[1])
        if len(f) > 2:
            self.audio_feats, cfg.data.audio_feat_dim = f[2], int(f[2].shape[1])
        self.diffusionData = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
        self.diffusionLoader = BatchLoader(self.diffusionData, int(cfg.train.batch), True, dev, seed=int(cfg.base.seed) + 2)

    @property
    def user_pos_items(self):
        return self.trainData.user_pos_items

    def getUserDegrees(self) -> np.ndarray:
        if not hasattr(self, "trainMat"):
            raise ValueError("Training matrix not loaded. Please call LoadData() first.")
        return np.bincount(self.graph.trn_rows, minlength=self.graph.users).astype(int)

    def modalities(self) -> int:
        return len(self.graph.feats)

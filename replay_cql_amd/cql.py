"""`CQL` -- the recommender class a user of RePlay imports (the slot `replay/models/cql.py` would fill; the
reference snapshot has no such file).

It follows the conventions of the reference's in-tree torch models: constructor kwargs mirrored by `_init_args`
(replay/models/neuromf.py:239-299), tunables in `_search_space` (neuromf.py:228-236), `_fit` collects the log once
(neuromf.py:332) and trains, `_predict` scores every requested user, `_save_model` / `_load_model` round-trip a single
file (replay/models/base_torch_rec.py:222-233; replay/model_handler.py:29-92).  Unlike TorchRecommender._predict
(base_torch_rec.py:120-149) scoring stays on the GPU: top-k + seen filtering are fused into the HIP scoring pass,
so `_predict` already returns exactly-k unseen rows and the wrapper's seen filter / top-k become no-ops."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import numpy as np
import pandas as pd
import torch

from . import data as D
from .core import CQLCore, CQLHyper
from .recommender_api import PandasRecommender, REC_COLUMNS


class CQL(PandasRecommender):
    """Discrete-action Conservative Q-Learning recommender (CQL(H) + double-Q target), SURVEY.md section 8.0.

    State = the user's last `window` items, action = next item, reward = relevance."""

    can_predict_cold_users = False   # users without history in `log` yield no rows (the MultVAE behaviour,
    can_predict_cold_items = False   # replay/models/mult_vae.py:138 + base_torch_rec.py:141-145)
    _search_space = {
        "learning_rate": {"type": "loguniform", "args": [1e-4, 1e-2]},
        "alpha": {"type": "loguniform", "args": [0.05, 5.0]},
        "embedding_dim": {"type": "categorical", "args": [64, 128, 256]},
        "window": {"type": "int", "args": [10, 100]},
        "gamma": {"type": "uniform", "args": [0.8, 0.999]},
    }

    # pylint: disable=too-many-arguments
    def __init__(self, embedding_dim: int = 128, window: int = 50, batch_size: int = 4096, epochs: int = 1,
                 n_steps: Optional[int] = None, learning_rate: float = 1e-3, gamma: float = 0.99, alpha: float = 1.0,
                 tau: float = 0.005, seed: int = 0, predict_cold_users: bool = False, device: Optional[str] = None,
                 valid_split_size: float = 0.0, patience: int = 3, factor: float = 0.5,
                 checkpoint_dir: Optional[str] = None):
        if embedding_dim not in (64, 128, 256):
            raise ValueError("embedding_dim must be 64, 128 or 256")
        if window <= 0 or batch_size <= 0 or epochs < 0:
            raise ValueError("window and batch_size must be positive, epochs non-negative")
        self.embedding_dim, self.window, self.batch_size = embedding_dim, window, batch_size
        self.epochs, self.n_steps = epochs, n_steps
        self.learning_rate, self.gamma, self.alpha, self.tau, self.seed = learning_rate, gamma, alpha, tau, seed
        self.predict_cold_users = predict_cold_users
        self.device = device
        if not 0.0 <= valid_split_size < 1.0:
            raise ValueError("valid_split_size must be in [0, 1)")
        # per-epoch validation, ReduceLROnPlateau and best-epoch checkpoint as TorchRecommender.train does
        # (replay/models/base_torch_rec.py:57-98; defaults of NeuroMF: replay/models/neuromf.py:222-227, :356-358)
        self.valid_split_size, self.patience, self.factor = valid_split_size, patience, factor
        self.checkpoint_dir = checkpoint_dir
        self.valid_losses: Optional[np.ndarray] = None
        self.best_epoch: Optional[int] = None
        self.core: Optional[CQLCore] = None
        self.train_losses: Optional[np.ndarray] = None
        self._rank, self._world, self._pg = 0, 1, None

    @property
    def _init_args(self) -> Dict[str, Any]:
        return {"embedding_dim": self.embedding_dim, "window": self.window, "batch_size": self.batch_size,
                "epochs": self.epochs, "n_steps": self.n_steps, "learning_rate": self.learning_rate,
                "gamma": self.gamma, "alpha": self.alpha, "tau": self.tau, "seed": self.seed,
                "predict_cold_users": self.predict_cold_users, "valid_split_size": self.valid_split_size,
                "patience": self.patience, "factor": self.factor}

    def set_distributed(self, rank: int, world: int, process_group=None) -> None:
        """Data-parallel training: every rank fits on its own user shard of the log; gradients are summed with RCCL."""
        self._rank, self._world, self._pg = rank, world, process_group

    # ------------------------------------------------------------------------------------------- fit
    def _hyper(self) -> CQLHyper:
        return CQLHyper(d=self.embedding_dim, window=self.window, batch=self.batch_size, gamma=self.gamma,
                        alpha=self.alpha, lr=self.learning_rate, tau=self.tau, seed=self.seed)

    def _fit(self, log: pd.DataFrame, user_features=None, item_features=None) -> None:
        dev = self.device or "cuda"
        offsets, items, rewards = D.build_csr_device(log["user_idx"].to_numpy(), log["item_idx"].to_numpy(),
                                                     log["timestamp"].to_numpy(), log["relevance"].to_numpy(),
                                                     self._user_dim_size, device=dev)      # sort + CSR on the GPU (f2)
        self.fit_arrays(offsets, items, rewards, self._item_dim_size)

    def fit_arrow(self, log, fit_users=None, fit_items=None) -> None:
        """fit() on Arrow record batches with LOG_SCHEMA columns (f1): what `_fit_wrap` derives from the log with four
        Spark actions (replay/models/base_rec.py:365-372: distinct users / items, max idx + 1) is one pass over the
        device columns; the CSR is built on the GPU; no pandas in between.
        fit_users / fit_items: id arrays a caller already holds (the Spark adapter passes what `_fit_wrap` computed,
        which may include ids that only occur in feature frames); default: the distinct ids of the log."""
        from . import arrow_io as A
        dev = torch.device(self.device or "cuda")
        c = A.columns_to_device(log, dev)
        if c["timestamp"] is None or c["relevance"] is None:
            raise ValueError("fit needs the LOG_SCHEMA columns user_idx, item_idx, timestamp, relevance")
        u, i = c["user_idx"], c["item_idx"]
        if u.numel() == 0:
            raise ValueError("empty log")
        lim = torch.stack([u.min(), u.max(), i.min(), i.max()]).cpu().tolist()
        if lim[0] < 0 or lim[2] < 0:
            raise ValueError("user_idx / item_idx must be non-negative dense indices")
        fu = torch.unique(u).cpu().numpy() if fit_users is None else np.unique(np.asarray(fit_users, dtype=np.int64))
        fi = torch.unique(i).cpu().numpy() if fit_items is None else np.unique(np.asarray(fit_items, dtype=np.int64))
        self.fit_users = pd.DataFrame({"user_idx": fu})
        self.fit_items = pd.DataFrame({"item_idx": fi})
        self._num_users, self._num_items = len(fu), len(fi)
        self._user_dim_size = max(int(fu[-1]), int(lim[1])) + 1
        self._item_dim_size = max(int(fi[-1]), int(lim[3])) + 1
        offsets, items, rewards = D.build_csr_device(u, i, c["timestamp"], c["relevance"], self._user_dim_size,
                                                     device=dev, check=False)
        self.fit_arrays(offsets, items, rewards, self._item_dim_size)

    def fit_arrays(self, offsets, items, rewards, n_items: int) -> None:
        """Array entry point (what bench.py and a Spark/Arrow adapter call): CSR by user, see data.build_csr.  Under
        data parallelism (set_distributed) every rank passes ITS user shard; the epoch length, the plateau decision
        and the best-epoch choice are agreed across ranks so that all ranks issue the same collectives."""
        from . import dist as DD
        self.core = CQLCore(n_items, self._hyper(), device=self.device, rank=self._rank, world=self._world,
                            process_group=self._pg)
        as_t = lambda x: x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x))   # noqa: E731
        offsets, items, rewards = as_t(offsets), as_t(items), as_t(rewards)
        n_users = offsets.numel() - 1
        n_valid = int(n_users * self.valid_split_size)
        valid = None
        if n_valid > 0 and self.n_steps is None and self.epochs > 0:
            # hold out the LAST n_valid users (rows of one user never straddle the split)
            cut = int(offsets[n_users - n_valid])
            valid = (offsets[n_users - n_valid:] - cut, items[cut:], rewards[cut:])
            offsets, items, rewards = offsets[: n_users - n_valid + 1], items[:cut], rewards[:cut]
        self.core.set_log(offsets, items, rewards)
        nnz = int(offsets[-1])
        if self.n_steps is not None:
            losses = self.core.train(int(self.n_steps))
            self.train_losses = losses.cpu().numpy()          # the only host sync of fit
            return
        # one epoch = one pass over the GLOBAL log: every rank takes the same number of steps (a per-rank count would
        # let ranks issue different numbers of gradient all-reduces)
        nnz_global = int(DD.sum_over_ranks(float(nnz), self.core.device, self._pg))
        steps_per_epoch = math.ceil(nnz_global / (self.batch_size * self._world))
        all_losses, valid_losses = [], []
        best, best_state, bad_epochs, lr = float("inf"), None, 0, self.learning_rate
        for epoch in range(self.epochs):
            all_losses.append(self.core.train(steps_per_epoch))
            if valid is None:
                continue
            n_vb = max(1, math.ceil(int(valid[0][-1]) / self.batch_size))
            v = self.core.eval_loss(*valid, n_batches=n_vb, seed=self.seed + 1)
            v = DD.sum_over_ranks(v, self.core.device, self._pg) / self._world      # same value, same branch, on every rank
            valid_losses.append(v)
            self.logger.debug("Epoch[%d] validation average loss: %.5f", epoch, v)
            if v < best * (1.0 - 1e-4):            # ReduceLROnPlateau(mode="min", threshold=1e-4, "rel")
                bad_epochs = 0
            else:
                bad_epochs += 1
                if bad_epochs > self.patience:
                    lr, bad_epochs = lr * self.factor, 0
                    self.core.set_lr(lr)
            if v < best:                           # best-epoch checkpoint (base_torch_rec.py:89-97)
                best, self.best_epoch = v, epoch
                best_state = {k: (t.clone() if torch.is_tensor(t) else t) for k, t in self.core.state_dict().items()}
                if self.checkpoint_dir is not None and self._rank == 0:
                    torch.save(best_state, f"{self.checkpoint_dir}/best_cql_{epoch + 1}_loss={v}.pt")
        if best_state is not None:                 # reload the best epoch (base_torch_rec.py:98)
            self.core.load_state_dict(best_state)
        self.train_losses = torch.cat(all_losses).cpu().numpy() if all_losses else np.zeros(0, np.float32)
        self.valid_losses = np.asarray(valid_losses, dtype=np.float64) if valid_losses else None
        if len(self.train_losses):
            self.logger.debug("CQL fit: %d steps, first/last loss %.5f / %.5f", len(self.train_losses),
                              self.train_losses[0], self.train_losses[-1])

    # ------------------------------------------------------------------------------------------- predict
    def _require_fit(self) -> CQLCore:
        if self.core is None:
            raise RuntimeError("CQL model is not fitted")
        return self.core

    @staticmethod
    def _pdf_cols(log: Optional[pd.DataFrame], dev) -> Optional[Dict[str, Optional[torch.Tensor]]]:
        """pandas log -> device columns (plumbing of the pandas / Spark entry points; Arrow callers skip this)."""
        if log is None or len(log) == 0:
            return None
        col = lambda n, dt: D._dev_col(log[n].to_numpy(), dt, dev)      # noqa: E731
        ts = D._dev_col(D.timestamp_key(log["timestamp"].to_numpy()), torch.int64, dev) if "timestamp" in log else None
        return {"user_idx": col("user_idx", torch.int32), "item_idx": col("item_idx", torch.int32), "timestamp": ts}

    def _device_states(self, cols, user_ids: torch.Tensor, want_seen: bool):
        """The predict-time CSR of the passed log, built ON THE DEVICE (cqlrec_build_csr; no host lexsort):
        returns (offsets, items in event order, seen items ascending per user or None).  Item ids are range-checked
        here: an id >= n_items would index past E_in in the window gather."""
        core = self._require_fit()
        dev = core.device
        n_rows = int(user_ids.max()) + 1 if user_ids.numel() else 1
        if cols is None or cols["user_idx"].numel() == 0:
            return (torch.zeros(n_rows + 1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                    None)
        u, i, t = cols["user_idx"], cols["item_idx"], cols.get("timestamp")
        lim = torch.stack([u.min(), u.max(), i.min(), i.max()]).cpu().tolist()        # one small sync
        if lim[0] < 0 or lim[2] < 0:
            raise ValueError("user_idx / item_idx must be non-negative dense indices")
        if lim[3] >= core.n_items:
            raise ValueError(f"log holds item_idx {int(lim[3])} but the model was fitted on {core.n_items} items; "
                             "filter cold items first (predict() does)")
        n_rows = max(n_rows, int(lim[1]) + 1)
        if t is None:          # a log without timestamps: event order = row order within each user (stable sort)
            t = torch.arange(u.numel(), dtype=torch.int64, device=dev)
        offsets, items, _ = D.build_csr_device(u, i, t, None, n_rows, device=dev, check=False)
        seen = None
        if want_seen:          # (user, item asc): the lists cqlrec_score_topk filters with; one trailing pad element
            _, s_items, _ = D.build_csr_device(u, i, None, None, n_rows, device=dev, check=False)
            seen = torch.cat([s_items, torch.zeros(1, dtype=torch.int32, device=dev)])
        return offsets, items, seen

    def _predict_device(self, cols, k: int, user_ids: torch.Tensor, cand: Optional[torch.Tensor],
                        filter_seen_items: bool):
        """S7 on the device.  cols: device columns of the passed log (or None); user_ids: sorted unique int64 device
        tensor; cand: sorted unique int64 device tensor of candidate items (None = whole catalogue).
        Returns device tensors (users int32 [n], idx int32 [n,k], val float32 [n,k], cnt int32 [n])."""
        core = self._require_fit()
        dev = core.device
        empty = (torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros((0, max(k, 0)), dtype=torch.int32, device=dev),
                 torch.zeros((0, max(k, 0)), dtype=torch.float32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev))
        if user_ids.numel() == 0 or k <= 0:
            return empty
        offsets, items, seen = self._device_states(cols, user_ids, filter_seen_items)
        if not self.predict_cold_users:       # "no history -> no rows" (base_torch_rec.py:141-145 inner join)
            user_ids = user_ids[(offsets[user_ids + 1] - offsets[user_ids]) > 0]
            if user_ids.numel() == 0:
                return empty
        users32 = user_ids.to(torch.int32)
        if cand is not None:
            cand = cand[cand < core.n_items]
            if cand.numel() == core.n_items:
                cand = None
        idx, val, cnt = core.encode_topk(offsets, items, users32, int(k), cand_items=cand,
                                         seen=None if seen is None else (offsets, seen),
                                         seen_rows=users32 if seen is not None else None)
        return users32, idx, val, cnt

    def _predict(self, log: Optional[pd.DataFrame], k: int, users: pd.DataFrame, items: pd.DataFrame,
                 user_features=None, item_features=None, filter_seen_items: bool = True) -> pd.DataFrame:
        from . import arrow_io as A
        core = self._require_fit()
        dev = core.device
        user_ids = A.ids_to_device(users["user_idx"].to_numpy(), "user_idx", dev)
        cand = A.ids_to_device(items["item_idx"].to_numpy(), "item_idx", dev)
        out = self._predict_device(self._pdf_cols(log, dev), k, user_ids, cand, filter_seen_items)
        return A.recs_to_arrow(*out).to_pandas()

    def predict_arrow(self, log, k: int, users=None, items=None, filter_seen_items: bool = True):
        """predict() for Arrow callers (f1): `log` = record batches with (at least) user_idx, item_idx [, timestamp];
        users / items = batches, arrays or iterables of ids (default: the users of `log` / the items seen at fit).
        Returns ONE pyarrow.RecordBatch with REC_SCHEMA holding exactly-k, seen-filtered rows per user ordered by
        (user_idx, relevance desc, item_idx asc) -- what `_predict_wrap` + `_filter_seen` + `get_top_k_recs`
        (replay/models/base_rec.py:467-539, :417-464; replay/utils.py:112-127) deliver, without their window passes.
        Cold users / items are dropped as `_filter_cold_for_predict` does (base_rec.py:560-603)."""
        from . import arrow_io as A
        core = self._require_fit()
        dev = core.device
        cols = None if log is None else A.columns_to_device(log, dev, ("user_idx", "item_idx", "timestamp"))
        if cols is not None and cols["user_idx"].numel() == 0:
            cols = None
        fit_u = torch.as_tensor(self.fit_users["user_idx"].to_numpy().astype(np.int64)).to(dev)
        fit_i = torch.as_tensor(self.fit_items["item_idx"].to_numpy().astype(np.int64)).to(dev)
        if users is not None:
            user_ids = A.ids_to_device(users, "user_idx", dev)
        elif cols is not None:
            user_ids = torch.unique(cols["user_idx"].to(torch.int64))
        else:
            user_ids = torch.unique(fit_u)
        cand = torch.unique(fit_i) if items is None else A.ids_to_device(items, "item_idx", dev)
        if not self.can_predict_cold_users:
            user_ids = user_ids[torch.isin(user_ids, fit_u)]
        if not self.can_predict_cold_items:
            cand = cand[torch.isin(cand, fit_i)]
        if cols is not None and not (self.can_predict_cold_users and self.can_predict_cold_items):
            keep = torch.isin(cols["user_idx"].to(torch.int64), fit_u) & torch.isin(cols["item_idx"].to(torch.int64), fit_i)
            cols = {n: (None if c is None else c[keep]) for n, c in cols.items()}
        return A.recs_to_arrow(*self._predict_device(cols, k, user_ids, cand, filter_seen_items))

    def _predict_pairs(self, pairs: pd.DataFrame, log=None, user_features=None, item_features=None) -> pd.DataFrame:
        core = self._require_fit()
        if log is None:
            raise ValueError("log is not provided, but it is required for prediction")   # as mult_vae / neuromf do
        dev = core.device
        pu = torch.as_tensor(pairs["user_idx"].to_numpy().astype(np.int64)).to(dev)
        pi = torch.as_tensor(pairs["item_idx"].to_numpy().astype(np.int64)).to(dev)
        if pi.numel() and (int(pi.min()) < 0 or int(pi.max()) >= core.n_items):
            raise ValueError("pairs hold an item_idx outside the fitted catalogue; filter cold items first")
        uniq, inv = torch.unique(pu, return_inverse=True)
        offsets, items, _ = self._device_states(self._pdf_cols(log, dev), uniq, want_seen=False)
        hb = core.encode(offsets, items, uniq.to(torch.int32))
        rel = core.pair_scores(hb.index_select(0, inv), pi.to(torch.int32))
        if not self.predict_cold_users:
            keep = (offsets[pu + 1] - offsets[pu]) > 0
            pu, pi, rel = pu[keep], pi[keep], rel[keep]
        return pd.DataFrame({"user_idx": pu.cpu().numpy().astype(np.int32), "item_idx": pi.cpu().numpy().astype(np.int32),
                             "relevance": rel.cpu().numpy().astype(np.float64)})

    def evaluate(self, log: pd.DataFrame, ground_truth: pd.DataFrame, ks=(10,), filter_seen_items: bool = True):
        """Quality of top-max(ks) recommendations for the users of `ground_truth`, computed on the GPU (f4): what
        optuna_objective.eval_quality (replay/optuna_objective.py:80-111) does with predict + a Spark metric, without
        the U x k block leaving the device.  Returns {metric: {k: value}} with the reference's metric definitions.
        `log` goes through the same cold filter as predict() (base_rec.py:560-603): a test-period log may hold users
        and items the model never saw."""
        from .metrics import evaluate_topk
        core = self._require_fit()
        dev = core.device
        kmax = int(max(ks))
        _, log = self._filter_cold(log, "user")
        _, log = self._filter_cold(log, "item")
        gt = ground_truth[["user_idx", "item_idx"]].drop_duplicates()
        gt_users = np.sort(gt["user_idx"].unique().astype(np.int64))
        known = np.isin(gt_users, self.fit_users["user_idx"].to_numpy())        # cold users count with empty predictions
        rec = torch.full((len(gt_users), kmax), -1, dtype=torch.int32, device=dev)
        if known.any():
            cand = torch.unique(torch.as_tensor(self.fit_items["item_idx"].to_numpy().astype(np.int64)).to(dev))
            cold_flag, self.predict_cold_users = self.predict_cold_users, False       # history-less users: empty rows
            try:
                users32, idx, _, _ = self._predict_device(self._pdf_cols(log, dev), kmax,
                                                          torch.as_tensor(gt_users[known]).to(dev), cand, filter_seen_items)
            finally:
                self.predict_cold_users = cold_flag
            if users32.numel():
                rows = torch.searchsorted(torch.as_tensor(gt_users).to(dev), users32.to(torch.int64))
                rec[rows] = idx
        # ground-truth CSR over the evaluated users (items ascending, unique): (row, item) sort on the device
        rows_h = np.searchsorted(gt_users, gt["user_idx"].to_numpy().astype(np.int64))
        g_off, g_items, _ = D.build_csr_device(rows_h, gt["item_idx"].to_numpy(), None, None, len(gt_users), device=dev)
        g_items = torch.cat([g_items, torch.zeros(1, dtype=torch.int32, device=dev)])
        return evaluate_topk(rec, g_off, g_items, ks)

    def _get_features(self, ids: pd.DataFrame, features):
        """Item embeddings (rows of E_out) in the shape ALS uses (replay/models/als.py:137-148)."""
        core = self._require_fit()
        if "item_idx" not in ids.columns:
            return None, None
        it = ids["item_idx"].to_numpy().astype(np.int64)
        E = core.segment(core.theta, "E_out").index_select(0, torch.as_tensor(it).to(core.device)).cpu().numpy()
        return pd.DataFrame({"item_idx": it.astype(np.int32), "item_factors": list(E.astype(np.float64))}), E.shape[1]

    # ------------------------------------------------------------------------------------------- persistence
    def _save_model(self, path: str) -> None:
        core = self._require_fit()
        # tensors, numbers and strings only: the file loads with weights_only=True (nothing in it is executed)
        torch.save({"state": core.state_dict(), "init_args": self._init_args,
                    "fit": {"users": torch.as_tensor(self.fit_users["user_idx"].to_numpy().astype(np.int64)),
                            "items": torch.as_tensor(self.fit_items["item_idx"].to_numpy().astype(np.int64)),
                            "user_dim": int(self._user_dim_size), "item_dim": int(self._item_dim_size)}}, path)

    def _load_model(self, path: str) -> None:
        blob = torch.load(path, weights_only=True, map_location="cpu")
        for k, v in blob["init_args"].items():
            setattr(self, k, v)
        self.core = CQLCore(int(blob["state"]["n_items"]), self._hyper(), device=self.device)
        self.core.load_state_dict(blob["state"])
        f = blob["fit"]
        self.fit_users = pd.DataFrame({"user_idx": f["users"].numpy()})
        self.fit_items = pd.DataFrame({"item_idx": f["items"].numpy()})
        self._num_users, self._num_items = len(self.fit_users), len(self.fit_items)
        self._user_dim_size, self._item_dim_size = int(f["user_dim"]), int(f["item_dim"])

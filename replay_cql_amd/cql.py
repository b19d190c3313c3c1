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

    def fit_arrays(self, offsets, items, rewards, n_items: int) -> None:
        """Array entry point (what bench.py and a Spark/Arrow adapter call): CSR by user, see data.build_csr."""
        self.core = CQLCore(n_items, self._hyper(), device=self.device, rank=self._rank, world=self._world,
                            process_group=self._pg)
        to_np = lambda x: x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)   # noqa: E731
        offsets = to_np(offsets)
        n_users = len(offsets) - 1
        n_valid = int(n_users * self.valid_split_size)
        valid = None
        if n_valid > 0 and self.n_steps is None and self.epochs > 0:
            # hold out the LAST n_valid users (rows of one user never straddle the split)
            cut = int(offsets[n_users - n_valid])
            items, rewards = to_np(items), to_np(rewards)
            valid = (offsets[n_users - n_valid:] - cut, items[cut:], rewards[cut:])
            offsets, items, rewards = offsets[: n_users - n_valid + 1], items[:cut], rewards[:cut]
        self.core.set_log(offsets, items, rewards)
        nnz = int(offsets[-1])
        if self.n_steps is not None:
            losses = self.core.train(int(self.n_steps))
            self.train_losses = losses.cpu().numpy()          # the only host sync of fit
            return
        steps_per_epoch = math.ceil(nnz / (self.batch_size * self._world))
        all_losses, valid_losses = [], []
        best, best_state, bad_epochs, lr = float("inf"), None, 0, self.learning_rate
        for epoch in range(self.epochs):
            all_losses.append(self.core.train(steps_per_epoch))
            if valid is None:
                continue
            n_vb = max(1, math.ceil(int(valid[0][-1]) / self.batch_size))
            v = self.core.eval_loss(*valid, n_batches=n_vb, seed=self.seed + 1)
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
                if self.checkpoint_dir is not None:
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

    def _states_for(self, log: Optional[pd.DataFrame], users: np.ndarray):
        """CSR of the passed log restricted to what predict needs; returns device tensors and row ids of `users`."""
        core = self._require_fit()
        n_rows = int(users.max()) + 1 if len(users) else 1
        if log is None or len(log) == 0:
            offsets = np.zeros(n_rows + 1, dtype=np.int64)
            items = np.zeros(1, dtype=np.int32)
        else:
            n_rows = max(n_rows, int(log["user_idx"].max()) + 1)
            offsets, items, _ = D.build_csr(log["user_idx"].to_numpy(), log["item_idx"].to_numpy(),
                                            log["timestamp"].to_numpy(), log["relevance"].to_numpy(), n_rows)
        dev = core.device
        d_off = torch.as_tensor(offsets).to(dev)
        d_items = torch.as_tensor(np.ascontiguousarray(items)).to(dev)
        return offsets, items, d_off, d_items

    def _predict(self, log: Optional[pd.DataFrame], k: int, users: pd.DataFrame, items: pd.DataFrame,
                 user_features=None, item_features=None, filter_seen_items: bool = True) -> pd.DataFrame:
        core = self._require_fit()
        user_ids = np.sort(users["user_idx"].to_numpy().astype(np.int64))
        offsets, log_items, d_off, d_items = self._states_for(log, user_ids)
        if not self.predict_cold_users:       # "no history -> no rows" (base_torch_rec.py:141-145 inner join)
            has_hist = (offsets[user_ids + 1] - offsets[user_ids]) > 0
            user_ids = user_ids[has_hist]
        if len(user_ids) == 0 or k <= 0:
            return pd.DataFrame({c: [] for c in REC_COLUMNS})
        cand = np.sort(items["item_idx"].to_numpy().astype(np.int64))
        cand = cand[cand < core.n_items]
        full = len(cand) == core.n_items
        dev = core.device
        d_users = torch.as_tensor(user_ids.astype(np.int32)).to(dev)
        hb = core.encode(d_off, d_items, d_users)
        seen = None
        if filter_seen_items and log is not None and len(log):
            seen = (d_off, torch.as_tensor(np.concatenate([D.sorted_seen(offsets, log_items), [0]]).astype(np.int32)).to(dev))
        idx, val, cnt = core.score_topk(hb, int(k), cand_items=None if full else torch.as_tensor(cand),
                                        seen=seen, seen_rows=d_users if seen is not None else None)
        idx, val, cnt = idx.cpu().numpy(), val.cpu().numpy(), cnt.cpu().numpy()
        keep = np.arange(idx.shape[1])[None, :] < cnt[:, None]
        return pd.DataFrame({"user_idx": np.repeat(user_ids, idx.shape[1]).reshape(idx.shape)[keep].astype(np.int32),
                             "item_idx": idx[keep].astype(np.int32),
                             "relevance": val[keep].astype(np.float64)})

    def _predict_pairs(self, pairs: pd.DataFrame, log=None, user_features=None, item_features=None) -> pd.DataFrame:
        core = self._require_fit()
        if log is None:
            raise ValueError("log is not provided, but it is required for prediction")   # as mult_vae / neuromf do
        pu = pairs["user_idx"].to_numpy().astype(np.int64)
        pi = pairs["item_idx"].to_numpy().astype(np.int64)
        uniq, inv = np.unique(pu, return_inverse=True)
        offsets, _, d_off, d_items = self._states_for(log, uniq)
        dev = core.device
        hb = core.encode(d_off, d_items, torch.as_tensor(uniq.astype(np.int32)).to(dev))
        hb_pairs = hb.index_select(0, torch.as_tensor(inv).to(dev))
        rel = core.pair_scores(hb_pairs, torch.as_tensor(pi.astype(np.int32)).to(dev)).cpu().numpy()
        out = pd.DataFrame({"user_idx": pu.astype(np.int32), "item_idx": pi.astype(np.int32),
                            "relevance": rel.astype(np.float64)})
        if not self.predict_cold_users:
            out = out[(offsets[pu + 1] - offsets[pu]) > 0]
        return out

    def evaluate(self, log: pd.DataFrame, ground_truth: pd.DataFrame, ks=(10,), filter_seen_items: bool = True):
        """Quality of top-max(ks) recommendations for the users of `ground_truth`, computed on the GPU (f4): what
        optuna_objective.eval_quality (replay/optuna_objective.py:80-111) does with predict + a Spark metric, without
        the U x k block leaving the device.  Returns {metric: {k: value}} with the reference's metric definitions."""
        from .metrics import evaluate_topk
        core = self._require_fit()
        kmax = int(max(ks))
        gt = ground_truth[["user_idx", "item_idx"]].drop_duplicates()
        gt_users = np.sort(gt["user_idx"].unique().astype(np.int64))
        known = np.isin(gt_users, self.fit_users["user_idx"].to_numpy())        # cold users count with empty predictions
        offsets, log_items, d_off, d_items = self._states_for(log, gt_users[known] if known.any() else gt_users[:0])
        has_hist = np.zeros(len(gt_users), dtype=bool)
        has_hist[known] = (offsets[gt_users[known] + 1] - offsets[gt_users[known]]) > 0
        rec = torch.full((len(gt_users), kmax), -1, dtype=torch.int32, device=core.device)
        if has_hist.any():
            users = torch.as_tensor(gt_users[has_hist].astype(np.int32)).to(core.device)
            hb = core.encode(d_off, d_items, users)
            seen = None
            if filter_seen_items and len(log):
                seen = (d_off, torch.as_tensor(np.concatenate([D.sorted_seen(offsets, log_items), [0]]).astype(np.int32))
                        .to(core.device))
            cand = np.sort(self.fit_items["item_idx"].to_numpy().astype(np.int64))
            full = len(cand) == core.n_items
            idx, _, _ = core.score_topk(hb, kmax, cand_items=None if full else torch.as_tensor(cand), seen=seen,
                                        seen_rows=users if seen is not None else None)
            rec[torch.as_tensor(np.nonzero(has_hist)[0]).to(core.device)] = idx
        # ground-truth CSR over the evaluated users (items ascending, unique)
        row_of = {u: r for r, u in enumerate(gt_users)}
        rows = gt["user_idx"].map(row_of).to_numpy()
        order = np.lexsort((gt["item_idx"].to_numpy(), rows))
        g_items = gt["item_idx"].to_numpy()[order].astype(np.int32)
        g_off = np.zeros(len(gt_users) + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=len(gt_users)), out=g_off[1:])
        return evaluate_topk(rec, torch.as_tensor(g_off).to(core.device),
                             torch.as_tensor(np.concatenate([g_items, [0]]).astype(np.int32)).to(core.device), ks)

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
        torch.save({"state": core.state_dict(), "init_args": self._init_args,
                    "fit": {"users": self.fit_users["user_idx"].to_numpy(), "items": self.fit_items["item_idx"].to_numpy(),
                            "user_dim": self._user_dim_size, "item_dim": self._item_dim_size}}, path)

    def _load_model(self, path: str) -> None:
        blob = torch.load(path, weights_only=False)       # a file this class wrote itself
        for k, v in blob["init_args"].items():
            setattr(self, k, v)
        self.core = CQLCore(int(blob["state"]["n_items"]), self._hyper(), device=self.device)
        self.core.load_state_dict(blob["state"])
        f = blob["fit"]
        self.fit_users = pd.DataFrame({"user_idx": f["users"]})
        self.fit_items = pd.DataFrame({"item_idx": f["items"]})
        self._num_users, self._num_items = len(f["users"]), len(f["items"])
        self._user_dim_size, self._item_dim_size = int(f["user_dim"]), int(f["item_dim"])

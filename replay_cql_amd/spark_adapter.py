"""`replay.models.CQL` -- the class a RePlay checkout gets: a thin subclass of the REAL `Recommender`
(replay/models/base_rec.py:1202-1335) around `replay_cql_amd.cql.CQL`, which carries the model and the GPU hot path.

    # replay/models/cql.py of a checkout (and `from replay.models.cql import CQL` in replay/models/__init__.py:10-27,
    # which model_handler.load's globals()[name] lookup needs -- replay/model_handler.py:69):
    from replay_cql_amd.spark_adapter import CQL

pyspark / replay are imported lazily: `spark_adapter.CQL` resolves on first access; the class itself is built by
`build_adapter(Recommender, State, REC_SCHEMA)`, which is also how tests/test_spark_adapter.py exercises it without
pyspark (duck-typed stand-ins for the DataFrame, the base class and the session).

What the adapter does at the boundary, and where the reference does the same:
  * `_fit`: ONE collect of the four LOG_SCHEMA columns -- as Arrow record batches when the DataFrame offers them
    (`_collect_as_arrow`, what `toPandas()` itself uses under `spark.sql.execution.arrow.pyspark.enabled`,
    replay/session_handler.py:47), else `toPandas()` exactly like NeuroMF._fit (replay/models/neuromf.py:332);
    the bookkeeping `_fit_wrap` computed (fit_users / fit_items / dims, base_rec.py:329-373) is handed to the inner
    model, which needs it for cold filtering, evaluate() and persistence.
  * `_predict`: scores ON THE DRIVER (GPU handles cannot be pickled into `applyInPandas` workers, cf.
    replay/models/base_torch_rec.py:132-148) and returns exactly-k, seen-filtered rows, so the wrapper's
    `_filter_seen` + `get_top_k_recs` (base_rec.py:514-528) are passes over U*k rows that drop nothing.
  * hyper-parameters are plain attributes (what `set_params`, base_rec.py:315-324, and optuna trials assign with
    setattr): they are forwarded to the inner model, `_init_args` reads them back (model_handler.save, :40-43), and
    `__init__` names every one of them explicitly because model_handler.load inspects the signature (:71-80).
  * `_save_model` / `_load_model` (base_rec.py:277-284): one file with parameters, Adam state, target network, step
    and the fit bookkeeping; `model_handler.load` sets `fit_users` / `fit_items` on the instance before `_load_model`."""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from .cql import CQL as _ArrayCQL

_HYPER = ("embedding_dim", "window", "batch_size", "epochs", "n_steps", "learning_rate", "gamma", "alpha", "tau", "seed",
          "predict_cold_users", "valid_split_size", "patience", "factor", "device", "checkpoint_dir")


def _ids(df, column: str) -> np.ndarray:
    """distinct-id DataFrame (or anything `_get_ids` accepts downstream) -> numpy ids."""
    return df.select(column).toPandas()[column].to_numpy()


def _collect_arrow(df, columns):
    """The selected columns as Arrow record batches (no pandas materialisation) when the DataFrame can hand them over,
    else as ONE batch made from toPandas()."""
    import pyarrow as pa
    sel = df.select(*columns)
    collect = getattr(sel, "_collect_as_arrow", None)
    if callable(collect):
        batches = collect()
        if batches:
            return batches
    return [pa.RecordBatch.from_pandas(sel.toPandas(), preserve_index=False)]


def build_adapter(Recommender, State, REC_SCHEMA):
    """Class factory: `Recommender` = replay.models.base_rec.Recommender, `State` = replay.session_handler.State,
    `REC_SCHEMA` = replay.constants.REC_SCHEMA (or stand-ins with the same duck type)."""

    class CQL(Recommender):  # pylint: disable=too-many-instance-attributes
        """Discrete-action Conservative Q-Learning recommender on MI355X (see replay_cql_amd.cql.CQL)."""

        can_predict_cold_users = False
        can_predict_cold_items = False
        _search_space = _ArrayCQL._search_space

        # pylint: disable=too-many-arguments
        def __init__(self, embedding_dim: int = 128, window: int = 50, batch_size: int = 4096, epochs: int = 1,
                     n_steps: Optional[int] = None, learning_rate: float = 1e-3, gamma: float = 0.99,
                     alpha: float = 1.0, tau: float = 0.005, seed: int = 0, predict_cold_users: bool = False,
                     valid_split_size: float = 0.0, patience: int = 3, factor: float = 0.5,
                     device: Optional[str] = None, checkpoint_dir: Optional[str] = None):
            object.__setattr__(self, "_impl", _ArrayCQL(
                embedding_dim=embedding_dim, window=window, batch_size=batch_size, epochs=epochs, n_steps=n_steps,
                learning_rate=learning_rate, gamma=gamma, alpha=alpha, tau=tau, seed=seed,
                predict_cold_users=predict_cold_users, device=device, valid_split_size=valid_split_size,
                patience=patience, factor=factor, checkpoint_dir=checkpoint_dir))

        # ---- hyper-parameters live on the inner model; attribute access on the wrapper reaches them ------------
        def __getattr__(self, name):          # only called when normal lookup fails
            if name in _HYPER:
                return getattr(object.__getattribute__(self, "_impl"), name)
            raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

        def __setattr__(self, name, value):
            if name in _HYPER:
                setattr(self._impl, name, value)
            else:
                object.__setattr__(self, name, value)

        @property
        def _init_args(self) -> Dict[str, Any]:
            return dict(self._impl._init_args)

        @property
        def _dataframes(self) -> Dict[str, Any]:
            return {}                          # fit_users / fit_items are written by model_handler.save itself

        def _clear_cache(self) -> None:
            pass

        @property
        def train_losses(self):
            return self._impl.train_losses

        # ---- hooks --------------------------------------------------------------------------------------------
        def _fit(self, log, user_features=None, item_features=None) -> None:
            impl = self._impl
            batches = _collect_arrow(log, ("user_idx", "item_idx", "timestamp", "relevance"))    # the one collect
            impl.fit_arrow(batches, fit_users=_ids(self.fit_users, "user_idx"), fit_items=_ids(self.fit_items, "item_idx"))
            # `_fit_wrap`'s numbers are authoritative (feature frames may add ids the log does not hold)
            impl._num_users, impl._num_items = int(self._num_users), int(self._num_items)

        def _to_spark(self, pdf, schema=None):
            if schema is None:
                return State().session.createDataFrame(pdf)
            return State().session.createDataFrame(pdf, schema=schema)

        # pylint: disable=too-many-arguments
        def _predict(self, log, k, users, items, user_features=None, item_features=None, filter_seen_items=True):
            batches = None if log is None else _collect_arrow(
                log, [c for c in ("user_idx", "item_idx", "timestamp") if c in log.columns])
            rb = self._impl.predict_arrow(batches, int(k), users=_ids(users, "user_idx"), items=_ids(items, "item_idx"),
                                          filter_seen_items=filter_seen_items)
            return self._to_spark(rb.to_pandas(), REC_SCHEMA)

        def _predict_pairs(self, pairs, log=None, user_features=None, item_features=None):
            if log is None:
                raise ValueError("log is not provided, but it is required for prediction")
            out = self._impl._predict_pairs(pairs.select("user_idx", "item_idx").toPandas(),
                                            log.select("user_idx", "item_idx", "timestamp").toPandas())
            return self._to_spark(out, REC_SCHEMA)

        def _get_features(self, ids, features):
            vecs, rank = self._impl._get_features(ids.toPandas(), None)
            if vecs is None:
                return None, None
            vecs = vecs.assign(item_factors=vecs["item_factors"].map(lambda v: [float(x) for x in v]))
            return self._to_spark(vecs), rank

        def evaluate(self, log, ground_truth, ks=(10,), filter_seen_items: bool = True):
            """NDCG / HitRate / Precision / Recall / MAP / MRR @ks on the device: what
            optuna_objective.eval_quality (replay/optuna_objective.py:80-111) computes through predict + Spark metrics."""
            return self._impl.evaluate(log.select("user_idx", "item_idx", "timestamp").toPandas(),
                                       ground_truth.select("user_idx", "item_idx").toPandas(), ks, filter_seen_items)

        def _save_model(self, path: str) -> None:
            self._impl._save_model(path)

        def _load_model(self, path: str) -> None:
            self._impl._load_model(path)

    CQL.__module__ = __name__
    return CQL


_REAL = None


def __getattr__(name):      # PEP 562: `from replay_cql_amd.spark_adapter import CQL` imports pyspark / replay only now
    global _REAL
    if name != "CQL":
        raise AttributeError(name)
    if _REAL is None:
        from replay.constants import REC_SCHEMA                # pylint: disable=import-outside-toplevel
        from replay.models.base_rec import Recommender         # pylint: disable=import-outside-toplevel
        from replay.session_handler import State               # pylint: disable=import-outside-toplevel
        _REAL = build_adapter(Recommender, State, REC_SCHEMA)
    return _REAL

"""`replay.models.CQL` -- the class a RePlay checkout gets: a thin subclass of the REAL `Recommender`
(replay/models/base_rec.py:1202-1335) around `replay_cql_amd.cql.CQL`, which carries the model and the GPU hot path.

    # replay/models/cql.py of a checkout (and `from replay.models.cql import CQL` in replay/models/__init__.py:10-27,
    # which model_handler.load's globals()[name] lookup needs -- replay/model_handler.py:69):
    from replay_cql_amd.spark_adapter import CQL

pyspark / replay are imported lazily: `spark_adapter.CQL` resolves on first access; the class itself is built by
`build_adapter(Recommender, State, REC_SCHEMA)`, which is also how tests/test_spark_adapter.py exercises it without
pyspark (duck-typed stand-ins for the DataFrame, the base class and the session).

What the adapter does at the boundary, and where the reference does the same:
  * `_fit`: ONE collect of the four LOG_SCHEMA columns through public API -- `DataFrame.toArrow()` where it exists
    (PySpark >= 4.0: Arrow record batches, no pandas), else `toPandas()` exactly like NeuroMF._fit
    (replay/models/neuromf.py:332; an Arrow collect under `spark.sql.execution.arrow.pyspark.enabled`,
    replay/session_handler.py:47);
    the bookkeeping `_fit_wrap` computed (fit_users / fit_items / dims, base_rec.py:329-373) is handed to the inner
    model, which needs it for cold filtering, evaluate() and persistence.
  * `_predict`: scores ON THE DRIVER (GPU handles cannot be pickled into `applyInPandas` workers, cf.
    replay/models/base_torch_rec.py:132-148) and returns exactly-k, seen-filtered rows, so the wrapper's
    `_filter_seen` + `get_top_k_recs` (base_rec.py:514-528) are passes over U*k rows that drop nothing.  The rows
    leave as a `pyarrow.Table` through `createDataFrame` where the session takes one (PySpark >= 4.0), as a pandas
    frame otherwise (base_torch_rec.py:143-148 returns pandas frames from its UDF too).
  * hyper-parameters are plain attributes (what `set_params`, base_rec.py:315-324, and optuna trials assign with
    setattr): they are forwarded to the inner model, `_init_args` reads them back (model_handler.save, :40-43), and
    `__init__` names every one of them explicitly because model_handler.load inspects the signature (:71-80).
  * `_save_model` / `_load_model` (base_rec.py:277-284): one file with parameters, Adam state, target network, step
    and the fit bookkeeping; `model_handler.load` sets `fit_users` / `fit_items` on the instance before `_load_model`."""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from .cql import CQL as _ArrayCQL

_ARROW_EGRESS: Dict[type, bool] = {}     # session type -> does createDataFrame take a pyarrow.Table?

_HYPER = ("embedding_dim", "window", "batch_size", "epochs", "n_steps", "learning_rate", "gamma", "alpha", "tau", "seed",
          "predict_cold_users", "valid_split_size", "patience", "factor", "device", "checkpoint_dir")


def _to_arrow_public(sel):
    """`DataFrame.toArrow()` -- PUBLIC API since PySpark 4.0 -- as record batches, or None where the DataFrame has no such
    method (PySpark 3.x: the reference pins 3.1.3)."""
    to_arrow = getattr(sel, "toArrow", None)
    if not callable(to_arrow):
        return None
    return to_arrow().to_batches()


def _ids(df, column: str) -> np.ndarray:
    """distinct-id DataFrame (or anything `_get_ids` accepts downstream) -> numpy ids."""
    sel = df.select(column)
    batches = _to_arrow_public(sel)
    if batches is not None:
        return np.concatenate([b.column(0).to_numpy(zero_copy_only=False) for b in batches]) if batches \
            else np.zeros(0, dtype=np.int64)
    return sel.toPandas()[column].to_numpy()


def _batch_to_pandas(rb):
    """REC_SCHEMA record batch -> pandas: the egress of sessions that cannot take Arrow data (PySpark 3.x).  A function
    of its own so that the Arrow-capable path can be shown never to call it (tests/test_spark_adapter.py)."""
    return rb.to_pandas()


def _collect_arrow(df, columns):
    """The selected columns as Arrow record batches through PUBLIC PySpark API only: `toArrow()` where the DataFrame has it
    (PySpark >= 4.0: no pandas materialisation), else `toPandas()` -- itself an Arrow collect on the reference's session
    (`spark.sql.execution.arrow.pyspark.enabled`, replay/session_handler.py:47) and exactly what NeuroMF._fit does
    (replay/models/neuromf.py:332) -- wrapped into ONE batch.  PySpark 3.x's private `DataFrame._collect_as_arrow`
    (what its toPandas() calls) skips the pandas copy there; it is used only on explicit request
    (CQL_SPARK_PRIVATE_ARROW=1), never by default."""
    import os
    import pyarrow as pa
    sel = df.select(*columns)
    batches = _to_arrow_public(sel)
    if batches:
        return batches
    if os.environ.get("CQL_SPARK_PRIVATE_ARROW") == "1":
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

        def _recs_to_spark(self, rb):
            """REC_SCHEMA record batch -> Spark DataFrame.  `SparkSession.createDataFrame(pyarrow.Table)` is public API
            since PySpark 4.0 (the release that added `DataFrame.toArrow`): there the U x k block goes over as Arrow
            data, no pandas in between.  Older sessions reject a Table with a TypeError (nothing was created): they get
            the pandas frame, which PySpark 3.x converts through Arrow itself under the session flag
            (replay/session_handler.py:47).  The answer is remembered per session type."""
            import pyarrow as pa
            session = State().session
            key = type(session)
            known = _ARROW_EGRESS.get(key)
            if known is True:
                return session.createDataFrame(pa.Table.from_batches([rb]), schema=REC_SCHEMA)
            if known is None:           # first use with this kind of session: try, remember
                try:
                    out = session.createDataFrame(pa.Table.from_batches([rb]), schema=REC_SCHEMA)
                    _ARROW_EGRESS[key] = True
                    return out
                except Exception as exc:  # pylint: disable=broad-except
                    # PySpark 3.x: "TypeError: ... can not accept object ... pyarrow.lib.Table"; whatever it was, the pandas
                    # route below re-raises a genuine failure
                    _ARROW_EGRESS[key] = False
                    self.logger.debug("createDataFrame(pyarrow.Table) not supported by this session (%r): pandas egress", exc)
            return self._to_spark(_batch_to_pandas(rb), REC_SCHEMA)

        # pylint: disable=too-many-arguments
        def _predict(self, log, k, users, items, user_features=None, item_features=None, filter_seen_items=True):
            batches = None if log is None else _collect_arrow(
                log, [c for c in ("user_idx", "item_idx", "timestamp") if c in log.columns])
            rb = self._impl.predict_arrow(batches, int(k), users=_ids(users, "user_idx"), items=_ids(items, "item_idx"),
                                          filter_seen_items=filter_seen_items)
            return self._recs_to_spark(rb)

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

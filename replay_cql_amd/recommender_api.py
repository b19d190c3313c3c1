"""pandas mirror of the reference's model API so that `CQL` can be used, and tested, without Spark.

The reference's boundary is the abstract class `Recommender(BaseRecommender)` (replay/models/base_rec.py:58-953,
:1202-1335) over Spark DataFrames; pyspark and a JVM are absent in this image, so the real base class
cannot even be imported.  This module restates the *wrapper* semantics -- same method names, argument meaning and
error behaviour -- on pandas DataFrames with the same schemas (replay/constants.py:16-31):

    fit            -> _fit_wrap          base_rec.py:1205-1217, :329-373
    predict        -> _predict_wrap      base_rec.py:1220-1257, :467-539   (cold filter, _predict, seen filter, top-k)
    predict_pairs  -> _predict_pairs_wrap base_rec.py:1259-1300, :725-782
    fit_predict                           base_rec.py:1303-1335
    _filter_seen                          base_rec.py:417-464
    _filter_cold / _filter_cold_for_predict  base_rec.py:560-603
    _get_ids                              base_rec.py:542-558
    set_params / __str__                  base_rec.py:315-327

A Spark DataFrame passed in is collected with toPandas() at this boundary (what NeuroMF._fit does at
replay/models/neuromf.py:332) and results are handed back as Spark DataFrames when a session exists; see
INTEGRATION.md for the thin subclass of the real `Recommender` that a RePlay checkout would use instead."""
from __future__ import annotations

import collections.abc
import logging
from abc import ABC, abstractmethod
from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np
import pandas as pd

LOG_COLUMNS = ["user_idx", "item_idx", "timestamp", "relevance"]   # replay/constants.py:16-23
REC_COLUMNS = ["user_idx", "item_idx", "relevance"]               # replay/constants.py:25-31

AnyDataFrame = Any


def _is_spark(df) -> bool:
    return df is not None and type(df).__module__.startswith("pyspark")


def to_pandas(df: Optional[AnyDataFrame]) -> Optional[pd.DataFrame]:
    if df is None:
        return None
    if isinstance(df, pd.DataFrame):
        return df
    if _is_spark(df):
        return df.toPandas()
    raise ValueError(f"Wrong type {type(df)}")


def get_top_k(df: pd.DataFrame, partition_by_col: str, order_by: Iterable[Tuple[str, bool]], k: int) -> pd.DataFrame:
    """row_number() over (partition by .. order by ..) <= k  (replay/utils.py:59-109)."""
    cols = [c for c, _ in order_by]
    asc = [a for _, a in order_by]
    out = df.sort_values([partition_by_col] + cols, ascending=[True] + asc, kind="stable")
    return out[out.groupby(partition_by_col, sort=False).cumcount() < k]


def get_top_k_recs(recs: pd.DataFrame, k: int) -> pd.DataFrame:
    """replay/utils.py:112-127; the reference leaves ties unspecified -- here: relevance desc, then item_idx asc."""
    return get_top_k(recs, "user_idx", [("relevance", False), ("item_idx", True)], k)


class PandasRecommender(ABC):
    """Same template-method structure as BaseRecommender/Recommender, on pandas."""

    can_predict_cold_users: bool = False
    can_predict_cold_items: bool = False
    can_predict_item_to_item: bool = False
    _search_space: Optional[Dict[str, Dict[str, Any]]] = None
    fit_users: pd.DataFrame
    fit_items: pd.DataFrame
    _num_users: int
    _num_items: int
    _user_dim_size: int
    _item_dim_size: int

    @property
    def logger(self) -> logging.Logger:
        return logging.getLogger("replay")          # same logger name as replay/session_handler.py:56-70

    # -------------------------------------------------------------------------------- bookkeeping
    @property
    def _init_args(self) -> Dict[str, Any]:
        return {}

    @property
    def _dataframes(self) -> Dict[str, Any]:
        return {}

    def set_params(self, **params: Dict[str, Any]) -> None:
        for param, value in params.items():
            setattr(self, param, value)
        self._clear_cache()

    def _clear_cache(self) -> None:
        pass

    def __str__(self) -> str:
        return type(self).__name__

    # -------------------------------------------------------------------------------- fit
    def fit(self, log, user_features=None, item_features=None) -> None:
        self._fit_wrap(log, user_features, item_features)

    def _fit_wrap(self, log, user_features=None, item_features=None) -> None:
        self.logger.debug("Starting fit %s", type(self).__name__)
        log, user_features, item_features = to_pandas(log), to_pandas(user_features), to_pandas(item_features)
        users = log["user_idx"]
        if user_features is not None:
            users = pd.concat([users, user_features["user_idx"]])
        items = log["item_idx"]
        if item_features is not None:
            items = pd.concat([items, item_features["item_idx"]])
        self.fit_users = pd.DataFrame({"user_idx": pd.unique(users)})
        self.fit_items = pd.DataFrame({"item_idx": pd.unique(items)})
        self._num_users = len(self.fit_users)
        self._num_items = len(self.fit_items)
        self._user_dim_size = int(self.fit_users["user_idx"].max()) + 1
        self._item_dim_size = int(self.fit_items["item_idx"].max()) + 1
        self._fit(log, user_features, item_features)

    @abstractmethod
    def _fit(self, log: pd.DataFrame, user_features=None, item_features=None) -> None:
        ...

    # -------------------------------------------------------------------------------- predict
    def predict(self, log, k: int, users=None, items=None, user_features=None, item_features=None,
                filter_seen_items: bool = True, recs_file_path: Optional[str] = None):
        return self._predict_wrap(log, k, users, items, user_features, item_features, filter_seen_items, recs_file_path)

    def fit_predict(self, log, k: int, users=None, items=None, user_features=None, item_features=None,
                    filter_seen_items: bool = True, recs_file_path: Optional[str] = None):
        self.fit(log, user_features, item_features)
        return self.predict(log, k, users, items, user_features, item_features, filter_seen_items, recs_file_path)

    @staticmethod
    def _get_ids(data: Union[Iterable, pd.DataFrame], column: str) -> pd.DataFrame:
        if _is_spark(data):
            data = data.select(column).distinct().toPandas()
        if isinstance(data, pd.DataFrame):
            return pd.DataFrame({column: pd.unique(data[column])})
        if isinstance(data, collections.abc.Iterable):
            return pd.DataFrame({column: pd.unique(pd.Series(list(data)))})
        raise ValueError(f"Wrong type {type(data)}")

    def _filter_cold(self, df: Optional[pd.DataFrame], entity: str, suffix: str = "idx"):
        if getattr(self, f"can_predict_cold_{entity}s") or df is None:
            return 0, df
        col = f"{entity}_{suffix}"
        known = getattr(self, f"fit_{entity}s")[col]
        mask = df[col].isin(known)
        num_cold = int(df.loc[~mask, col].nunique())
        if num_cold == 0:
            return 0, df
        return num_cold, df[mask]

    def _filter_cold_for_predict(self, main_df, log_df, entity: str, suffix: str = "idx"):
        num_new, main_df = self._filter_cold(main_df, entity, suffix)
        if num_new > 0:
            self.logger.info("%s model can't predict cold %ss, they will be ignored", self, entity)
        _, log_df = self._filter_cold(log_df, entity, suffix)
        return main_df, log_df

    def _filter_seen(self, recs: pd.DataFrame, log: pd.DataFrame, k: int, users: pd.DataFrame) -> pd.DataFrame:
        """k .. k + seen(user) best rows per user, then anti-join with the log (base_rec.py:417-464)."""
        users_log = log[log["user_idx"].isin(users["user_idx"])]
        num_seen = users_log.groupby("user_idx")["item_idx"].count().rename("seen_count")
        recs = recs.sort_values(["user_idx", "relevance", "item_idx"], ascending=[True, False, True], kind="stable")
        rank = recs.groupby("user_idx", sort=False).cumcount() + 1
        seen_cnt = recs["user_idx"].map(num_seen).fillna(0).astype(np.int64)
        recs = recs[rank <= seen_cnt + k]
        seen_pairs = pd.MultiIndex.from_frame(users_log[["user_idx", "item_idx"]].drop_duplicates())
        mask = pd.MultiIndex.from_frame(recs[["user_idx", "item_idx"]]).isin(seen_pairs)
        return recs[~mask]

    def _predict_wrap(self, log, k: int, users=None, items=None, user_features=None, item_features=None,
                      filter_seen_items: bool = True, recs_file_path: Optional[str] = None):
        self.logger.debug("Starting predict %s", type(self).__name__)
        if not hasattr(self, "fit_items"):
            raise RuntimeError(f"{self} model is not fitted")
        spark_out = _is_spark(log)
        log, user_features, item_features = to_pandas(log), to_pandas(user_features), to_pandas(item_features)
        user_data = next((x for x in (users, log, user_features, self.fit_users) if x is not None), None)
        users = self._get_ids(user_data, "user_idx")
        users, log = self._filter_cold_for_predict(users, log, "user")
        item_data = items if items is not None else self.fit_items
        items = self._get_ids(item_data, "item_idx")
        items, log = self._filter_cold_for_predict(items, log, "item")
        num_items = len(items)
        if num_items < k:
            self.logger.debug("k = %d > number of items = %d", k, num_items)
        recs = self._predict(log, k, users, items, user_features, item_features, filter_seen_items)
        if filter_seen_items and log is not None:
            recs = self._filter_seen(recs=recs, log=log, users=users, k=k)
        recs = get_top_k_recs(recs, k)[REC_COLUMNS].reset_index(drop=True)
        return self._deliver(recs, recs_file_path, spark_out)

    @abstractmethod
    def _predict(self, log: Optional[pd.DataFrame], k: int, users: pd.DataFrame, items: pd.DataFrame,
                 user_features=None, item_features=None, filter_seen_items: bool = True) -> pd.DataFrame:
        ...

    # -------------------------------------------------------------------------------- pairs
    def predict_pairs(self, pairs, log=None, user_features=None, item_features=None,
                      recs_file_path: Optional[str] = None, k: Optional[int] = None):
        return self._predict_pairs_wrap(pairs, log, user_features, item_features, recs_file_path, k)

    def _predict_pairs_wrap(self, pairs, log=None, user_features=None, item_features=None,
                            recs_file_path: Optional[str] = None, k: Optional[int] = None):
        spark_out = _is_spark(pairs)
        log, user_features, item_features, pairs = [to_pandas(df) for df in (log, user_features, item_features, pairs)]
        if sorted(pairs.columns) != ["item_idx", "user_idx"]:
            raise ValueError("pairs must be a dataframe with columns strictly [user_idx, item_idx]")
        pairs, log = self._filter_cold_for_predict(pairs, log, "user")
        pairs, log = self._filter_cold_for_predict(pairs, log, "item")
        pred = self._predict_pairs(pairs=pairs, log=log, user_features=user_features, item_features=item_features)
        if k:
            pred = get_top_k(pred, "user_idx", [("relevance", False), ("item_idx", True)], k)
        return self._deliver(pred[REC_COLUMNS].reset_index(drop=True), recs_file_path, spark_out)

    def _predict_pairs(self, pairs: pd.DataFrame, log=None, user_features=None, item_features=None) -> pd.DataFrame:
        """Fallback through _predict, as base_rec.py:784-823."""
        self.logger.warning("native predict_pairs is not implemented for this model. "
                            "Falling back to usual predict method and filtering the results.")
        users = pd.DataFrame({"user_idx": pd.unique(pairs["user_idx"])})
        items = pd.DataFrame({"item_idx": pd.unique(pairs["item_idx"])})
        pred = self._predict(log, len(items), users, items, user_features, item_features, filter_seen_items=False)
        return pred.merge(pairs[["user_idx", "item_idx"]], on=["user_idx", "item_idx"], how="inner")

    # -------------------------------------------------------------------------------- features
    def _get_features_wrap(self, ids, features):
        ids = to_pandas(ids)
        if "user_idx" not in ids.columns and "item_idx" not in ids.columns:
            raise ValueError("user_idx or item_idx missing")
        return self._get_features(ids, to_pandas(features))

    def _get_features(self, ids: pd.DataFrame, features):
        self.logger.info("get_features method is not defined for the model %s. Features will not be returned.", str(self))
        return None, None

    def get_nearest_items(self, items, k: int, metric: Optional[str] = "cosine_similarity", candidates=None):
        raise NotImplementedError(f"item-to-item prediction is not implemented for {self}")   # base_rec.py:934-936

    # -------------------------------------------------------------------------------- output
    @staticmethod
    def _deliver(recs: pd.DataFrame, recs_file_path: Optional[str], spark_out: bool):
        recs = recs.astype({"user_idx": np.int32, "item_idx": np.int32, "relevance": np.float64})
        if recs_file_path is not None:
            recs.to_parquet(recs_file_path)
            return None
        if spark_out:  # hand a Spark DataFrame back when we were given one
            from pyspark.sql import SparkSession  # pragma: no cover - pyspark is absent in this image
            return SparkSession.builder.getOrCreate().createDataFrame(recs)  # pragma: no cover
        return recs

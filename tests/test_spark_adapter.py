"""The drop-in class `replay_cql_amd.spark_adapter.CQL(Recommender)` exercised WITHOUT pyspark (absent in this image,
SURVEY F5) through duck-typed stand-ins:

  * `FakeDF`      -- the handful of Spark DataFrame calls the reference's wrappers and the adapter make
                    (select / distinct / count / agg(max).collect / join / toPandas / columns);
  * `FakeRecommender` -- the calls `BaseRecommender` makes around the hooks: `_fit_wrap` (replay/models/base_rec.py:329-373),
                    the id extraction / cold filter / seen filter / top-k of `_predict_wrap` (:467-539, :417-464;
                    replay/utils.py:112-127), `_predict_pairs_wrap` (:725-782), `set_params` (:315-324);
  * `FakeState`   -- `State().session.createDataFrame(pandas_df, schema=...)`;
  * `handler_save` / `handler_load` -- what replay/model_handler.py:29-92 does with a model (init_args JSON,
                    signature inspection, fit_users / fit_items set BEFORE `_load_model`).

Contracts mirrored from the reference's own tests: save/load round trip (tests/models/test_save_load_models.py:48-70),
predict_pairs == predict restricted to the pairs (tests/models/test_all_models.py:128-144), k + seen
(tests/models/test_all_models.py:378-390)."""
import json
from inspect import getfullargspec

import numpy as np
import pandas as pd
import pytest

from replay_cql_amd.spark_adapter import build_adapter

REC_SCHEMA = ("user_idx", "item_idx", "relevance")


class FakeDF:
    def __init__(self, pdf):
        self.pdf = pdf.reset_index(drop=True)

    @property
    def columns(self):
        return list(self.pdf.columns)

    def select(self, *cols):
        return FakeDF(self.pdf[list(cols)])

    def distinct(self):
        return FakeDF(self.pdf.drop_duplicates())

    def count(self):
        return len(self.pdf)

    def agg(self, spec):
        (col, fn), = spec.items()
        assert fn == "max"
        return FakeDF(pd.DataFrame({f"max({col})": [self.pdf[col].max()]}))

    def collect(self):
        return [tuple(r) for r in self.pdf.itertuples(index=False)]

    def join(self, other, on, how="inner"):
        return FakeDF(self.pdf.merge(other.pdf, on=on, how=how))

    def toPandas(self):
        return self.pdf.copy()


class FakeState:
    """a PySpark-3.x-like session: createDataFrame takes pandas frames, not Arrow tables"""
    class _Session:
        @staticmethod
        def createDataFrame(pdf, schema=None):
            if not isinstance(pdf, pd.DataFrame):
                raise TypeError(f"StructType can not accept object {type(pdf).__name__}")
            if schema is not None:
                assert list(pdf.columns) == list(schema)
            return FakeDF(pdf)
    session = _Session()


class ArrowDF(FakeDF):
    """a PySpark-4-like DataFrame: public toArrow(); the private 3.x collect and toPandas() must stay unused"""
    calls = []

    def select(self, *cols):
        return ArrowDF(self.pdf[list(cols)])

    def distinct(self):
        return ArrowDF(self.pdf.drop_duplicates())

    def join(self, other, on, how="inner"):
        return ArrowDF(self.pdf.merge(other.pdf, on=on, how=how))

    def toArrow(self):
        import pyarrow as pa
        ArrowDF.calls.append("toArrow")
        return pa.Table.from_pandas(self.pdf, preserve_index=False)

    def _collect_as_arrow(self):
        raise AssertionError("the adapter must not use PySpark's private _collect_as_arrow by default")


class ArrowState:
    """a PySpark-4-like session: createDataFrame(pyarrow.Table, schema) is public API"""
    class _Session:
        tables = []

        @staticmethod
        def createDataFrame(data, schema=None):
            import pyarrow as pa
            if isinstance(data, pa.Table):
                assert schema is None or data.column_names == list(schema)
                ArrowState._Session.tables.append(data)
                return ArrowDF(data.to_pandas())
            return ArrowDF(data)
    session = _Session()


class FakeRecommender:
    can_predict_cold_users = False
    can_predict_cold_items = False
    study = None
    logger = __import__("logging").getLogger("replay")        # BaseRecommender.logger (base_rec.py:639-646)

    def set_params(self, **params):                       # base_rec.py:315-324
        for param, value in params.items():
            setattr(self, param, value)
        self._clear_cache()

    def __str__(self):
        return type(self).__name__

    def fit(self, log, user_features=None, item_features=None):
        self._fit_wrap(log, user_features, item_features)

    def _fit_wrap(self, log, user_features=None, item_features=None):     # base_rec.py:329-373
        users = log.select("user_idx").distinct()
        items = log.select("item_idx").distinct()
        self.fit_users, self.fit_items = users, items
        self._num_users, self._num_items = users.count(), items.count()
        self._user_dim_size = self.fit_users.agg({"user_idx": "max"}).collect()[0][0] + 1
        self._item_dim_size = self.fit_items.agg({"item_idx": "max"}).collect()[0][0] + 1
        self._fit(log, user_features, item_features)

    def _known(self, df, column, fit_df):                 # _filter_cold: base_rec.py:560-603
        return df.join(fit_df, on=column)

    def predict(self, log, k, users=None, items=None, filter_seen_items=True):        # base_rec.py:467-539
        users = (log if users is None else FakeDF(pd.DataFrame({"user_idx": list(users)}))).select("user_idx").distinct()
        items = (self.fit_items if items is None else FakeDF(pd.DataFrame({"item_idx": list(items)}))).select("item_idx").distinct()
        users, items = self._known(users, "user_idx", self.fit_users), self._known(items, "item_idx", self.fit_items)
        log = self._known(self._known(log, "user_idx", self.fit_users), "item_idx", self.fit_items)
        recs = self._predict(log, k, users, items, None, None, filter_seen_items).toPandas()
        if filter_seen_items:                             # _filter_seen: anti-join with the log (base_rec.py:417-464)
            seen = pd.MultiIndex.from_frame(log.pdf[["user_idx", "item_idx"]].drop_duplicates())
            recs = recs[~pd.MultiIndex.from_frame(recs[["user_idx", "item_idx"]]).isin(seen)]
        recs = recs.sort_values(["user_idx", "relevance", "item_idx"], ascending=[True, False, True], kind="stable")
        return FakeDF(recs[recs.groupby("user_idx").cumcount() < k])         # get_top_k_recs: utils.py:112-127

    def predict_pairs(self, pairs, log=None):             # base_rec.py:725-782
        pairs = self._known(self._known(pairs, "user_idx", self.fit_users), "item_idx", self.fit_items)
        return self._predict_pairs(pairs, log)


def handler_save(model, path):                            # replay/model_handler.py:29-51
    model._save_model(str(path / "model"))
    init_args = model._init_args
    init_args["_model_name"] = str(model)
    (path / "init_args.json").write_text(json.dumps(init_args))
    assert model._dataframes == {}
    return {"fit_users": model.fit_users, "fit_items": model.fit_items}


def handler_load(cls, path, frames):                      # replay/model_handler.py:54-92
    args = json.loads((path / "init_args.json").read_text())
    assert args.pop("_model_name") == cls.__name__
    init_args = getfullargspec(cls.__init__).args
    init_args.remove("self")
    assert not set(args) - set(init_args), "model_handler.load would mis-handle init args missing from __init__"
    model = cls(**args)
    for name, df in frames.items():
        setattr(model, name, df)
    model._load_model(str(path / "model"))
    return model


CQL = build_adapter(FakeRecommender, FakeState, REC_SCHEMA)


def _log(U=90, NI=300, seed=4):
    rng = np.random.default_rng(seed)
    rows = []
    for u in range(U):
        n = int(rng.integers(4, 30))
        its = rng.integers(0, NI, n)
        for t, it in enumerate(its):
            rows.append((u, int(it), pd.Timestamp("2021-01-01") + pd.Timedelta(minutes=int(t)), float(rng.integers(1, 6)) / 5))
    return FakeDF(pd.DataFrame(rows, columns=["user_idx", "item_idx", "timestamp", "relevance"]))


def test_adapter_contract_without_gpu():
    """signature, attribute forwarding, init args: everything model_handler / set_params / optuna touch."""
    m = CQL(embedding_dim=64, window=7, batch_size=32, n_steps=3, seed=9)
    assert issubclass(CQL, FakeRecommender) and CQL.__name__ == "CQL" and str(m) == "CQL"
    sig = getfullargspec(CQL.__init__).args
    assert set(m._init_args) <= set(sig) - {"self"}
    json.dumps(m._init_args)                                                    # JSON-serialisable (model_handler.py:40-43)
    assert m.window == 7 and m._init_args["window"] == 7
    m.set_params(window=11, learning_rate=0.01)                                 # reaches the inner model
    assert m._impl.window == 11 and m._init_args["learning_rate"] == 0.01 and m.window == 11
    assert m._search_space["learning_rate"]["type"] == "loguniform"
    assert m._dataframes == {} and m.can_predict_cold_users is False
    with pytest.raises(AttributeError):
        m.no_such_attribute                                                     # noqa: B018
    with pytest.raises(RuntimeError, match="not fitted"):
        m._save_model("/nonexistent")


def _rec_batch():
    import pyarrow as pa
    from replay_cql_amd.arrow_io import REC_SCHEMA as ARROW_REC
    return pa.record_batch([pa.array([0, 0, 1], pa.int32()), pa.array([5, 7, 2], pa.int32()),
                            pa.array([0.5, 0.25, 1.0], pa.float64())], schema=ARROW_REC)


def test_arrow_capable_session_gets_arrow_tables_no_pandas(monkeypatch):
    """f1, Spark half: on a session whose createDataFrame takes a pyarrow.Table (public API since PySpark
    4.0) the recommendations leave as Arrow data -- the pandas conversion is never called -- and ingest goes through the
    public DataFrame.toArrow(), never through the private _collect_as_arrow or toPandas()."""
    from replay_cql_amd import spark_adapter as SA
    monkeypatch.setattr(SA, "_batch_to_pandas", lambda rb: (_ for _ in ()).throw(AssertionError("to_pandas on the Arrow path")))
    monkeypatch.setattr(ArrowDF, "toPandas", lambda self: (_ for _ in ()).throw(AssertionError("toPandas on the Arrow path")))
    SA._ARROW_EGRESS.clear()
    CQL4 = build_adapter(FakeRecommender, ArrowState, REC_SCHEMA)
    m = CQL4(embedding_dim=64, window=4, batch_size=32, n_steps=1)
    rb = _rec_batch()
    for _ in range(2):                                   # second call: the remembered answer
        ArrowState._Session.tables.clear()
        out = m._recs_to_spark(rb)
        assert len(ArrowState._Session.tables) == 1 and ArrowState._Session.tables[0].num_rows == 3
        assert list(out.pdf.columns) == list(REC_SCHEMA)
    # ingest: record batches from toArrow(); ids from toArrow()
    log = ArrowDF(_log(U=5).pdf)
    ArrowDF.calls.clear()
    batches = SA._collect_arrow(log, ("user_idx", "item_idx", "timestamp", "relevance"))
    assert ArrowDF.calls == ["toArrow"] and sum(b.num_rows for b in batches) == log.count()
    assert batches[0].schema.names == ["user_idx", "item_idx", "timestamp", "relevance"]
    ids = SA._ids(log.select("user_idx").distinct(), "user_idx")
    assert sorted(ids.tolist()) == list(range(5))


def test_pandas_only_session_falls_back_once(monkeypatch):
    """a PySpark-3.x-like session rejects the Table: the adapter falls back to the pandas frame and does not try again"""
    from replay_cql_amd import spark_adapter as SA
    SA._ARROW_EGRESS.clear()
    m = CQL(embedding_dim=64, window=4, batch_size=32, n_steps=1)
    calls = []
    real = FakeState._Session.createDataFrame
    monkeypatch.setattr(FakeState._Session, "createDataFrame",
                        staticmethod(lambda data, schema=None: (calls.append(type(data).__name__), real(data, schema))[1]))
    out = m._recs_to_spark(_rec_batch())
    assert calls == ["Table", "DataFrame"] and out.count() == 3
    out = m._recs_to_spark(_rec_batch())
    assert calls == ["Table", "DataFrame", "DataFrame"]
    # the 3.x private collect is opt-in only: by default a DataFrame without toArrow() is collected with toPandas()
    class Df3(FakeDF):
        def select(self, *cols):
            return Df3(self.pdf[list(cols)])

        def _collect_as_arrow(self):
            raise AssertionError("private API used without CQL_SPARK_PRIVATE_ARROW=1")
    monkeypatch.delenv("CQL_SPARK_PRIVATE_ARROW", raising=False)
    b = SA._collect_arrow(Df3(_log(U=3).pdf), ("user_idx", "item_idx"))
    assert len(b) == 1 and b[0].num_columns == 2


@pytest.fixture(scope="module")
def fitted():
    log = _log()
    m = CQL(embedding_dim=64, window=6, batch_size=64, n_steps=15, seed=2, device="cuda:0")
    m.fit(log)
    return log, m


@pytest.mark.gpu
def test_fit_predict_through_the_wrapper(fitted):
    log, m = fitted
    assert m._impl._user_dim_size == m._user_dim_size and m._impl._item_dim_size == m._item_dim_size
    assert m._impl._num_users == m._num_users == log.select("user_idx").distinct().count()
    assert len(m.train_losses) == 15 and np.all(np.isfinite(m.train_losses))
    k = 5
    raw = m._predict(log, k, m.fit_users, m.fit_items).toPandas()
    recs = m.predict(log, k).toPandas()
    # `_predict` already returns exactly k unseen rows per user, best first: the wrapper's seen filter and top-k drop nothing
    pd.testing.assert_frame_equal(raw.reset_index(drop=True), recs.reset_index(drop=True))
    assert recs.groupby("user_idx").size().eq(k).all() and recs.user_idx.nunique() == m._num_users
    assert not set(zip(recs.user_idx, recs.item_idx)) & set(zip(log.pdf.user_idx, log.pdf.item_idx))
    assert recs.dtypes.tolist() == [np.dtype("int32"), np.dtype("int32"), np.dtype("float64")]
    # k + seen, no filter: seen items may come back (tests/models/test_all_models.py:378-390)
    more = m.predict(log, k, filter_seen_items=False).toPandas()
    assert more.groupby("user_idx").size().eq(k).all()
    # unknown users / items are ignored
    sub = m.predict(log, 3, users=[0, 1, 10_000], items=list(range(40)) + [10**6]).toPandas()
    assert set(sub.user_idx) <= {0, 1} and set(sub.item_idx) <= set(range(40))


@pytest.mark.gpu
def test_save_load_round_trip_through_the_handler(tmp_path, fitted):
    """tests/models/test_save_load_models.py:48-70: recommendations of the re-loaded model are identical."""
    log, m = fitted
    frames = handler_save(m, tmp_path)
    m2 = handler_load(CQL, tmp_path, frames)
    assert m2._init_args == m._init_args
    pd.testing.assert_frame_equal(m2.predict(log, 7).toPandas(), m.predict(log, 7).toPandas())
    assert m2._impl.core.step == m._impl.core.step == 15
    # the inner model got its bookkeeping back from the file: evaluate() and persistence work on the loaded model
    assert len(m2._impl.fit_items) == m._num_items
    handler_save(m2, tmp_path)


@pytest.mark.gpu
def test_predict_pairs_contract(fitted):
    """tests/models/test_all_models.py:128-144: pairs scoring == full scoring restricted to the pairs."""
    log, m = fitted
    known = np.sort(log.pdf.item_idx.unique())
    pairs = FakeDF(pd.DataFrame({"user_idx": [0, 0, 3, 7, 7, 10_000], "item_idx": list(known[[5, 9, 2, 2, 30]]) + [1]}))
    pred = m.predict_pairs(pairs, log).toPandas()
    assert len(pred) == 5 and list(pred.columns) == list(REC_SCHEMA)
    full = m.predict(log, m._item_dim_size, users=[0, 3, 7], filter_seen_items=False).toPandas()
    merged = pred.merge(full, on=["user_idx", "item_idx"], suffixes=("", "_full"))
    assert len(merged) == 5
    np.testing.assert_allclose(merged.relevance, merged.relevance_full, atol=1e-3)
    with pytest.raises(ValueError, match="log is not provided"):
        m.predict_pairs(pairs)


@pytest.mark.gpu
def test_features_and_device_metrics(fitted):
    log, m = fitted
    vecs, rank = m._get_features(FakeDF(pd.DataFrame({"item_idx": [0, 1, 2]})), None)
    assert rank == 64 and vecs.count() == 3 and len(vecs.toPandas().item_factors.iloc[0]) == 64
    test = FakeDF(log.pdf.sample(frac=0.1, random_state=0)[["user_idx", "item_idx"]])
    got = m.evaluate(log, test, ks=[1, 5])
    assert set(got) >= {"NDCG", "HitRate", "Precision", "Recall", "MAP", "MRR"} and 0.0 <= got["HitRate"][5] <= 1.0

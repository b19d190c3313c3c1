"""CPU tests of the host layer: the pandas mirror of the reference's Recommender wrappers (semantics cited from
replay/models/base_rec.py), CSR construction, seen lists, the synthetic generator.  A deterministic fake model stands
in for the GPU core so that the wrapper logic is tested without a GPU."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd import data as D
from replay_cql_amd.recommender_api import PandasRecommender, get_top_k_recs


class ScoreTable(PandasRecommender):
    """relevance(u, i) = table[u, i]; returns, like the reference's torch models, MORE than k rows per user."""

    def __init__(self, table, cold_users=False):
        self.table = table
        self.can_predict_cold_users = cold_users
        self.fit_calls = 0

    def _fit(self, log, user_features=None, item_features=None):
        self.fit_calls += 1

    def _predict(self, log, k, users, items, user_features=None, item_features=None, filter_seen_items=True):
        rows = [(u, i, self.table[u % self.table.shape[0], i]) for u in users["user_idx"] for i in items["item_idx"]]
        return pd.DataFrame(rows, columns=["user_idx", "item_idx", "relevance"])


@pytest.fixture
def log():
    # the shape of the reference's tiny fixture (tests/utils.py:59-76): 4 users, 4 items, 11 rows
    return pd.DataFrame({
        "user_idx": [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3],
        "item_idx": [0, 1, 2, 0, 1, 3, 1, 2, 3, 0, 3],
        "timestamp": pd.to_datetime(["2020-01-%02d" % d for d in range(1, 12)]),
        "relevance": [1.0, 0.5, 0.25, 1.0, 0.75, 0.5, 1.0, 1.0, 0.5, 0.25, 1.0],
    })


TABLE = np.array([[4.0, 3.0, 2.0, 1.0], [1.0, 2.0, 3.0, 4.0], [2.0, 2.0, 2.0, 2.0], [0.5, 4.5, 1.5, 3.5]])


def test_fit_wrap_sets_dims(log):
    m = ScoreTable(TABLE)
    m.fit(log)
    assert m.fit_calls == 1 and m._num_users == 4 and m._num_items == 4
    assert m._user_dim_size == 4 and m._item_dim_size == 4
    assert str(m) == "ScoreTable"


def test_predict_filters_seen_and_takes_top_k(log):
    m = ScoreTable(TABLE)
    recs = m.fit_predict(log, k=2)
    assert list(recs.columns) == ["user_idx", "item_idx", "relevance"]
    seen = set(zip(log.user_idx, log.item_idx))
    assert not (set(zip(recs.user_idx, recs.item_idx)) & seen)
    # user 0 saw items 0,1,2 -> only item 3 is left; user 2 saw 1,2,3 -> only item 0
    assert recs[recs.user_idx == 0].item_idx.tolist() == [3]
    assert recs[recs.user_idx == 2].item_idx.tolist() == [0]
    recs2 = m.predict(log, k=2, filter_seen_items=False)
    assert recs2.groupby("user_idx").size().tolist() == [2, 2, 2, 2]
    assert recs2[recs2.user_idx == 0].item_idx.tolist() == [0, 1]
    assert recs2[recs2.user_idx == 2].item_idx.tolist() == [0, 1]        # all-equal scores: item_idx asc
    assert recs2.relevance.dtype == np.float64 and recs2.item_idx.dtype == np.int32


def test_cold_users_and_items_are_dropped(log):
    m = ScoreTable(TABLE)
    m.fit(log)
    recs = m.predict(log, k=1, users=[0, 7, 9], items=[1, 2, 3, 42], filter_seen_items=False)
    assert set(recs.user_idx) == {0} and set(recs.item_idx) <= {1, 2, 3}
    m2 = ScoreTable(TABLE, cold_users=True)
    m2.fit(log)
    recs = m2.predict(log, k=1, users=[0, 7], filter_seen_items=False)
    assert set(recs.user_idx) == {0, 7}


def test_predict_pairs_contract(log):
    m = ScoreTable(TABLE)
    m.fit(log)
    pairs = pd.DataFrame({"user_idx": [0, 0, 1, 3], "item_idx": [1, 3, 2, 0]})
    pred = m.predict_pairs(pairs, log)
    assert len(pred) == 4
    got = {(u, i): r for u, i, r in pred.itertuples(index=False)}
    assert got[(0, 1)] == 3.0 and got[(3, 0)] == 0.5
    assert m.predict_pairs(pairs, log, k=1).groupby("user_idx").size().max() == 1
    with pytest.raises(ValueError, match="strictly"):
        m.predict_pairs(pairs.assign(extra=1), log)


def test_predict_to_parquet_equals_returned(tmp_path, log):
    m = ScoreTable(TABLE)
    m.fit(log)
    path = str(tmp_path / "recs.parquet")
    assert m.predict(log, k=2, recs_file_path=path) is None
    pd.testing.assert_frame_equal(pd.read_parquet(path), m.predict(log, k=2))


def test_get_ids_and_errors(log):
    assert PandasRecommender._get_ids([3, 3, 1], "user_idx")["user_idx"].tolist() == [3, 1]
    assert PandasRecommender._get_ids(log, "item_idx")["item_idx"].tolist() == [0, 1, 2, 3]
    with pytest.raises(ValueError, match="Wrong type"):
        PandasRecommender._get_ids(5, "user_idx")
    m = ScoreTable(TABLE)
    m.set_params(fit_calls=5)
    assert m.fit_calls == 5
    with pytest.raises(NotImplementedError):
        m.get_nearest_items([1], 2)


def test_get_top_k_recs_tie_rule():
    df = pd.DataFrame({"user_idx": [0] * 4, "item_idx": [3, 1, 2, 0], "relevance": [1.0, 2.0, 2.0, 2.0]})
    assert get_top_k_recs(df, 2).item_idx.tolist() == [0, 1]


def test_build_csr_matches_oracle_and_validates():
    u, i, t, r = O.synth_log(50, 97, seed=5, mean_len=9, max_len=30)
    perm = np.random.default_rng(0).permutation(len(u))
    a = D.build_csr(u[perm], i[perm], t[perm], r[perm], 50)
    b = O.build_csr(u[perm], i[perm], t[perm], r[perm], 50)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # datetime timestamps, ties broken by item_idx
    ts = pd.to_datetime(["2020-01-02", "2020-01-01", "2020-01-01"]).to_numpy()
    off, items, rew = D.build_csr([0, 0, 0], [5, 9, 7], ts, [1.0, 2.0, 3.0], 2)
    assert off.tolist() == [0, 3, 3] and items.tolist() == [7, 9, 5] and rew.tolist() == [3.0, 2.0, 1.0]
    with pytest.raises(ValueError):
        D.build_csr([0, -1], [1, 2], [0, 1], [1.0, 1.0])
    seen = D.sorted_seen(off, items)
    assert seen.tolist() == [5, 7, 9]


def test_synthetic_generator_is_shardable_and_deterministic():
    off, items, rew = D.synth_log_device(3000, 500, seed=9, device="cpu")
    o2, i2, r2 = D.synth_log_device(3000, 500, seed=9, device="cpu", user_lo=1000, user_hi=1700)
    a, b = int(off[1000]), int(off[1700])
    assert torch.equal(items[a:b], i2) and torch.equal(rew[a:b], r2) and torch.equal(off[1000:1701] - off[1000], o2)
    lens = (off[1:] - off[:-1])
    assert int(lens.min()) >= 5 and int(lens.max()) <= 200 and 30 < float(lens.float().mean()) < 60
    assert int(items.min()) >= 0 and int(items.max()) < 500
    assert set(np.round(rew.unique().numpy().astype(np.float64), 4).tolist()) <= {0.2, 0.4, 0.6, 0.8, 1.0}
    o3, i3, _ = D.synth_log_device(3000, 500, seed=10, device="cpu")
    assert not torch.equal(i3[:1000], items[:1000])


def test_float_timestamps_keep_their_order():
    """fractional float timestamps must not be truncated -- host and device builders sort the same
    order-preserving int64 keys (IEEE total-order map), so fit and predict see one event order."""
    ts = np.array([-2.5, -1.0, -0.0, 0.0, 1e-300, 0.25, 0.5, 1.0, 3.25, -1e300, 1e300, 7.75, 7.5])
    k = D.timestamp_key(ts)
    assert k.dtype == np.int64
    assert np.array_equal(np.argsort(k, kind="stable"), np.argsort(ts, kind="stable"))
    assert k[2] == k[3]                                   # -0.0 and +0.0 tie (the tie then falls to item_idx)
    assert np.array_equal(D.timestamp_key(torch.tensor(ts)).numpy(), k)
    ts32 = ts[np.abs(ts) < 1e30].astype(np.float32)
    assert np.array_equal(D.timestamp_key(ts32), D.timestamp_key(ts32.astype(np.float64)))
    with pytest.raises(ValueError, match="NaN"):
        D.timestamp_key(np.array([0.5, np.nan]))
    # three events of one user within the same second: truncation to int64 would reorder them by item_idx
    off, items, rew = D.build_csr([0, 0, 0], [9, 5, 7], [10.75, 10.5, 10.25], [1.0, 2.0, 3.0], 1)
    assert items.tolist() == [7, 5, 9] and rew.tolist() == [3.0, 2.0, 1.0]
    rng = np.random.default_rng(3)
    u, i = rng.integers(0, 40, 3000), rng.integers(0, 50, 3000)
    t = np.round(rng.standard_normal(3000) * 3, 1)          # many ties, negatives, fractions
    r = rng.integers(1, 6, 3000) / 5.0
    for x, y in zip(D.build_csr(u, i, t, r, 40), O.build_csr(u, i, t, r, 40)):
        assert np.array_equal(x, y)


def test_arrow_ingest_and_egress_on_cpu():
    """f1 plumbing without a GPU: Arrow batches -> columns of the hot path's dtypes, REC_SCHEMA egress."""
    pa = pytest.importorskip("pyarrow")
    from replay_cql_amd import arrow_io as A
    ts = pd.to_datetime(["2020-01-03", "2020-01-01", "2020-01-02", "2020-01-05"])
    t1 = pa.table({"user_idx": pa.array([1, 0, 1, 2], pa.int64()), "item_idx": pa.array([3, 2, 1, 0], pa.int32()),
                   "timestamp": pa.array(ts.to_numpy().astype("datetime64[us]")), "relevance": pa.array([1, 2, 3, 4], pa.int32())})
    batches = t1.to_batches(max_chunksize=3)
    assert len(batches) == 2
    c = A.columns_to_device(batches, "cpu")
    assert c["user_idx"].dtype == torch.int32 and c["user_idx"].tolist() == [1, 0, 1, 2]
    assert c["relevance"].dtype == torch.float64 and c["relevance"].tolist() == [1.0, 2.0, 3.0, 4.0]
    assert c["timestamp"].dtype == torch.int64
    assert np.array_equal(np.argsort(c["timestamp"].numpy()), np.argsort(ts.to_numpy()))
    # same columns as the pandas frame gives (LOG_SCHEMA: replay/constants.py:16-23)
    assert A.LOG_SCHEMA.names == ["user_idx", "item_idx", "timestamp", "relevance"]
    only_ids = A.columns_to_device(t1.select(["user_idx", "item_idx"]), "cpu")
    assert only_ids["timestamp"] is None and only_ids["relevance"] is None
    with pytest.raises(ValueError, match="nulls"):
        A.columns_to_device(pa.table({"user_idx": pa.array([1, None], pa.int32()), "item_idx": pa.array([1, 2], pa.int32())}), "cpu")
    with pytest.raises(Exception):                       # int64 id that does not fit IntegerType
        A.columns_to_device(pa.table({"user_idx": pa.array([2**40]), "item_idx": pa.array([1])}), "cpu")
    with pytest.raises(ValueError, match="no column"):
        A.columns_to_device(pa.table({"user_idx": pa.array([1], pa.int32())}), "cpu")
    assert A.ids_to_device([5, 3, 5, 1], "user_idx", "cpu").tolist() == [1, 3, 5]
    assert A.ids_to_device(t1, "user_idx", "cpu").tolist() == [0, 1, 2]
    # egress: [n x k] block + counts -> exactly the valid prefix of every row, REC_SCHEMA
    users = torch.tensor([4, 7], dtype=torch.int32)
    idx = torch.tensor([[9, 8, -1], [1, 2, 3]], dtype=torch.int32)
    val = torch.tensor([[0.5, 0.25, float("-inf")], [3.0, 2.0, 1.0]])
    rb = A.recs_to_arrow(users, idx, val, torch.tensor([2, 3], dtype=torch.int32))
    assert rb.schema.equals(A.REC_SCHEMA)
    assert rb.to_pydict() == {"user_idx": [4, 4, 7, 7, 7], "item_idx": [9, 8, 1, 2, 3], "relevance": [0.5, 0.25, 3.0, 2.0, 1.0]}
    assert A.recs_to_arrow(users[:0], idx[:0], val[:0], torch.zeros(0, dtype=torch.int32)).num_rows == 0


def test_bench_hbm_roofline_never_reports_more_than_the_peak():
    """bench.hbm_roofline: frac is null for a table that fits the Infinity Cache (the algorithmic rate may exceed the HBM
    peak there) and capped at 1 otherwise ('bench never prints frac > 1')."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("bench_mod", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    small = bench.hbm_roofline(alg_bytes=650e6, ms=0.066, table_bytes=25.6e6)           # 9.8 TB/s of cached rows
    assert small["frac"] is None and small["infinity_cache_resident"] and small["achieved"] > bench.PEAK_HBM_GBS
    big = bench.hbm_roofline(alg_bytes=3.47e9, ms=0.29, table_bytes=512e6)                # 12 TB/s algorithmic
    assert big["frac"] == 1.0 and not big["infinity_cache_resident"]
    adam = bench.hbm_roofline(alg_bytes=11.8e9, ms=2.43, table_bytes=5.4e9)
    assert 0.55 < adam["frac"] < 0.65


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failure path: needs a box WITHOUT a GPU")
def test_bench_self_launch_propagates_a_failing_rank():
    """bench.py --gpus 2 without a launcher starts its own ranks (tests/test_gpu_dp.py runs the real thing); here, with
    no GPU, every rank fails at once: the parent must return a non-zero code promptly instead of hanging or printing a
    JSON line."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "stopping the other ranks" in r.stderr or "exited with" in r.stderr

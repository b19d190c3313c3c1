"""The `CQL` recommender on the GPU through the reference-shaped API (fit / predict / predict_pairs / save / load).
Mirrors the contract tests of the reference for its torch models (tests/models/test_all_models.py:128-144, :378-390;
tests/models/test_save_load_models.py:48-70; tests/models/test_neuromf.py:61-88) with the oracle as the checker."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd.cql import CQL

pytestmark = pytest.mark.gpu

U, NI, D_, L = 120, 700, 64, 8


@pytest.fixture(scope="module")
def log():
    u, i, t, r = O.synth_log(U, NI, seed=6, mean_len=14, max_len=40)
    # make item ids sparse at the top so that _item_dim_size > number of distinct items
    return pd.DataFrame({"user_idx": u, "item_idx": i, "timestamp": pd.to_datetime(t, unit="s"), "relevance": r})


@pytest.fixture(scope="module")
def model(log):
    m = CQL(embedding_dim=D_, window=L, batch_size=64, n_steps=12, seed=3, device="cuda:0")
    m.fit(log)
    return m


def _oracle_model(m):
    lay = O.Layout.make(m.core.n_items, D_)
    return lay, m.core.theta.cpu().numpy()


def test_fit_trains_and_matches_oracle_trajectory(log, model):
    assert model.train_losses.shape == (12,) and np.all(np.isfinite(model.train_losses))
    assert model._num_users == log.user_idx.nunique() and model._item_dim_size == log.item_idx.max() + 1
    # replay the same 12 steps on the CPU
    off, items, rew = O.build_csr(log.user_idx, log.item_idx, log.timestamp.values, log.relevance, model._user_dim_size)
    om = O.OracleModel.create(model._item_dim_size, D_, seed=7)
    from replay_cql_amd.core import CQLCore, CQLHyper
    fresh = CQLCore(model._item_dim_size, CQLHyper(d=D_, window=L, batch=64, seed=3), device="cuda:0")
    om.theta[:] = fresh.theta.cpu().numpy()
    om.target[:] = om.theta
    ref = O.train_steps(om, off, items, rew, 12, 64, L, seed=3)
    np.testing.assert_allclose(model.train_losses, ref, rtol=1e-3)
    assert np.linalg.norm(model.core.theta.cpu().numpy() - om.theta) < 1e-3 * np.linalg.norm(om.theta)


def test_predict_matches_oracle_topk(log, model):
    k = 7
    recs = model.predict(log, k=k)
    assert list(recs.columns) == ["user_idx", "item_idx", "relevance"]
    assert recs.groupby("user_idx").size().eq(k).all() and recs.user_idx.nunique() == log.user_idx.nunique()
    seen = set(zip(log.user_idx, log.item_idx))
    assert not (set(zip(recs.user_idx, recs.item_idx)) & seen)
    lay, theta = _oracle_model(model)
    off, items, rew = O.build_csr(log.user_idx, log.item_idx, log.timestamp.values, log.relevance, model._user_dim_size)
    users = np.sort(log.user_idx.unique())
    fit_items = np.sort(log.item_idx.unique())
    ridx, rval, rcnt, hb = O.predict_topk(lay, theta, off, items, users, k, L, filter_seen=True, cand_items=fit_items)
    Q = O.qvalues(hb, O.bf16_round(lay.view(theta, "E_out")), lay.view(theta, "b_out"))
    diff = 0
    for row, u in enumerate(users):
        got = recs[recs.user_idx == u]
        assert np.all(np.diff(got.relevance.values) <= 0)
        for j in set(got.item_idx) ^ set(ridx[row]):
            assert abs(Q[row, j] - rval[row, -1]) < 2e-3
            diff += 1
        np.testing.assert_allclose(got.relevance.values, Q[row, got.item_idx.values], atol=2e-3)
    assert diff <= 4


def test_predict_users_items_subsets_and_cold(log, model):
    recs = model.predict(log, k=3, users=[0, 5, 9, 10_000], items=[1, 2, 3, 4, 5, 6, 10**6], filter_seen_items=False)
    assert set(recs.user_idx) == {0, 5, 9} and set(recs.item_idx) <= {1, 2, 3, 4, 5, 6}
    assert recs.groupby("user_idx").size().eq(3).all()
    # users without history in the passed log produce no rows (reference: tests/models/test_vae.py:56-62)
    sub = log[log.user_idx < 10]
    recs = model.predict(sub, k=3, users=[1, 2, 50])
    assert set(recs.user_idx) == {1, 2}
    # k larger than the number of admissible items
    known = np.sort(log.item_idx.unique())[:20]
    recs = model.predict(log, k=50, users=[0], items=list(known) + [10**6], filter_seen_items=False)
    assert len(recs) == 20 and set(recs.item_idx) == set(known)


def test_predict_pairs(log, model):
    known = np.sort(log.item_idx.unique())
    pairs = pd.DataFrame({"user_idx": [0, 0, 3, 7, 7], "item_idx": known[[5, 9, 2, 2, 100]]})
    pred = model.predict_pairs(pairs, log)
    assert len(pred) == 5 and list(pred.columns) == ["user_idx", "item_idx", "relevance"]
    full = model.predict(log, k=model._item_dim_size, users=[0, 3, 7], filter_seen_items=False)
    merged = pred.merge(full, on=["user_idx", "item_idx"], suffixes=("", "_full"))
    np.testing.assert_allclose(merged.relevance, merged.relevance_full, atol=1e-3)   # the shadowed reference test :63-96
    assert model.predict_pairs(pairs, log, k=1).groupby("user_idx").size().max() == 1
    with pytest.raises(ValueError, match="log is not provided"):
        model.predict_pairs(pairs)


def test_save_load_round_trip(tmp_path, log, model):
    path = str(tmp_path / "model")
    model._save_model(path)
    m2 = CQL(device="cuda:0")
    m2._load_model(path)
    assert m2._init_args == model._init_args
    pd.testing.assert_frame_equal(m2.predict(log, k=5), model.predict(log, k=5))
    assert m2.core.step == model.core.step
    # resume: one more step on both gives identical parameters
    for m in (model, m2):
        off, items, rew = O.build_csr(log.user_idx, log.item_idx, log.timestamp.values, log.relevance, m._user_dim_size)
        m.core.set_log(off, items, rew)
        m.core.train(1)
    # the step is deterministic (no float atomics): resuming from a checkpoint continues bit for bit
    assert torch.equal(m2.core.theta, model.core.theta)


def test_item_features_and_errors(log, model):
    vecs, rank = model._get_features_wrap(pd.DataFrame({"item_idx": [0, 1, 2]}), None)
    assert rank == D_ and len(vecs) == 3 and len(vecs.item_factors.iloc[0]) == D_
    with pytest.raises(ValueError):
        CQL(embedding_dim=100)
    with pytest.raises(RuntimeError, match="not fitted"):
        CQL(device="cuda:0").predict(log, k=1, users=[0])


def test_epochs_validation_plateau_and_best_checkpoint(tmp_path, log):
    """TorchRecommender.train semantics (replay/models/base_torch_rec.py:57-98): per-epoch validation loss on held-out
    users, LR reduced on a plateau, the best epoch is what the model ends up with; a checkpoint file appears
    (reference test: tests/models/test_neuromf.py:123-143)."""
    m = CQL(embedding_dim=D_, window=L, batch_size=64, epochs=4, seed=3, device="cuda:0", valid_split_size=0.25,
            patience=0, factor=0.5, learning_rate=5e-3, checkpoint_dir=str(tmp_path))
    m.fit(log)
    assert m.valid_losses is not None and len(m.valid_losses) == 4 and np.all(np.isfinite(m.valid_losses))
    assert m.best_epoch == int(np.argmin(m.valid_losses))
    files = list(tmp_path.glob("best_cql_*_loss=*.pt"))
    assert files, "no best-epoch checkpoint written"
    # the model holds the parameters of the best epoch: its validation loss is reproduced by eval_loss
    off, items, rew = O.build_csr(log.user_idx, log.item_idx, log.timestamp.values, log.relevance, m._user_dim_size)
    n_users = len(off) - 1
    n_valid = int(n_users * 0.25)
    cut = int(off[n_users - n_valid])
    v = m.core.eval_loss(off[n_users - n_valid:] - cut, items[cut:], rew[cut:],
                         n_batches=max(1, -(-int(off[-1] - cut) // 64)), seed=3 + 1)
    assert abs(v - m.valid_losses.min()) < 1e-5 * abs(v)
    # validation loss equals the oracle's loss on the same held-out batches
    lay = O.Layout.make(m.core.n_items, D_)
    theta, target = m.core.theta.cpu().numpy(), m.core.target.cpu().numpy()
    vo, vi, vr = off[n_users - n_valid:] - cut, items[cut:], rew[cut:]
    ref = []
    for b in range(max(1, -(-int(vo[-1]) // 64))):
        pos = O.sample_positions(4, b, 0, 64, int(vo[-1]))
        us, tp = O.positions_to_transitions(pos, vo)
        ref.append(O.loss_and_grads(lay, theta, target, vo, vi, vr, us, tp, L, 0.99, 1.0).loss)
    assert abs(v - np.mean(ref)) < 1e-3 * abs(v)


def test_on_device_evaluation_matches_reference_metric_definitions(log, model):
    """model.evaluate == the reference's metrics (oracle/metrics_oracle.py, pinned by tests/test_metrics.py values)
    applied to model.predict output."""
    from oracle import metrics_oracle as M
    rng = np.random.default_rng(2)
    test = log.sample(frac=0.15, random_state=1)[["user_idx", "item_idx"]]
    extra = pd.DataFrame({"user_idx": [10_000, 10_000], "item_idx": [1, 2]})       # a cold user in the ground truth
    test = pd.concat([test, extra], ignore_index=True)
    train = log.drop(test.index, errors="ignore")
    ks = [1, 5, 10]
    got = model.evaluate(train, test, ks=ks)
    recs = model.predict(train, k=10, users=test.user_idx.unique())
    ref = M.evaluate(recs.user_idx, recs.item_idx, recs.relevance, test.user_idx, test.item_idx, ks)
    name = {"ndcg": "NDCG", "hitrate": "HitRate", "precision": "Precision", "recall": "Recall", "map": "MAP", "mrr": "MRR"}
    for m, d in ref.items():
        for k, v in d.items():
            assert got[name[m]][k] == pytest.approx(v, rel=1e-9, abs=1e-12), (m, k)


def _arrow_log(log, chunk=500):
    import pyarrow as pa
    tbl = pa.table({"user_idx": pa.array(log.user_idx.to_numpy().astype(np.int32)),
                    "item_idx": pa.array(log.item_idx.to_numpy().astype(np.int32)),
                    "timestamp": pa.array(log.timestamp.to_numpy().astype("datetime64[us]")),
                    "relevance": pa.array(log.relevance.to_numpy().astype(np.float64))})
    return tbl.to_batches(max_chunksize=chunk)


def test_arrow_fit_and_predict_equal_the_pandas_path(log, model):
    """f1: Arrow record batches in, one REC_SCHEMA batch out -- same bookkeeping, same training trajectory (bit for
    bit: the step is deterministic), same recommendations as fit()/predict() on the pandas frame."""
    from replay_cql_amd import arrow_io as A
    batches = _arrow_log(log)
    assert len(batches) > 1
    model = CQL(embedding_dim=D_, window=L, batch_size=64, n_steps=12, seed=3, device="cuda:0")
    model.fit(log)                    # a fresh pandas-path twin (the shared fixture is trained further by other tests)
    m2 = CQL(embedding_dim=D_, window=L, batch_size=64, n_steps=12, seed=3, device="cuda:0")
    m2.fit_arrow(batches)
    assert (m2._num_users, m2._num_items, m2._user_dim_size, m2._item_dim_size) == \
        (model._num_users, model._num_items, model._user_dim_size, model._item_dim_size)
    assert np.array_equal(np.sort(m2.fit_items.item_idx.values), np.sort(model.fit_items.item_idx.values))
    assert np.array_equal(m2.train_losses, model.train_losses)
    assert torch.equal(m2.core.theta, model.core.theta)
    k = 7
    want = model.predict(log, k=k)
    rb = m2.predict_arrow(batches, k)
    assert rb.schema.equals(A.REC_SCHEMA)
    got = rb.to_pandas()
    pd.testing.assert_frame_equal(got, want.sort_values(["user_idx", "relevance", "item_idx"], ascending=[True, False, True],
                                                       kind="stable").reset_index(drop=True))
    # users / items subsets, unseen ids dropped as _filter_cold_for_predict does; no seen filter
    want = model.predict(log, k=3, users=[0, 5, 9, 10_000], items=[1, 2, 3, 4, 5, 6, 10**6], filter_seen_items=False)
    got = m2.predict_arrow(batches, 3, users=[0, 5, 9, 10_000], items=[1, 2, 3, 4, 5, 6, 10**6],
                           filter_seen_items=False).to_pandas()
    pd.testing.assert_frame_equal(got, want.reset_index(drop=True))
    # a log without timestamps is legal for predict: event order = row order
    import pyarrow as pa
    srt = log.sort_values(["user_idx", "timestamp", "item_idx"], kind="stable")
    no_ts = pa.table({"user_idx": pa.array(srt.user_idx.to_numpy().astype(np.int32)),
                      "item_idx": pa.array(srt.item_idx.to_numpy().astype(np.int32))})
    pd.testing.assert_frame_equal(m2.predict_arrow(no_ts, k).to_pandas(), rb.to_pandas())


def test_out_of_range_item_ids_raise_before_any_kernel(log, model):
    """ids >= n_items would be an out-of-bounds gather / scatter; every array entry point checks them."""
    from replay_cql_amd.core import CQLCore, CQLHyper
    core = CQLCore(50, CQLHyper(d=64, window=4, batch=32), device="cuda:0")
    off = np.array([0, 3, 5], dtype=np.int64)
    with pytest.raises(ValueError, match="item ids"):
        core.set_log(off, np.array([1, 2, 50, 3, 4], dtype=np.int32), np.ones(5, np.float32))
    with pytest.raises(ValueError, match="item ids"):
        core.set_log(off, np.array([1, 2, -1, 3, 4], dtype=np.int32), np.ones(5, np.float32))
    bad = pd.concat([log, pd.DataFrame({"user_idx": [0], "item_idx": [model._item_dim_size + 3],
                                        "timestamp": [log.timestamp.max()], "relevance": [1.0]})], ignore_index=True)
    with pytest.raises(ValueError, match="fitted on"):       # _predict called directly, as scenarios do
        model._predict(bad, 3, pd.DataFrame({"user_idx": [0]}), model.fit_items)
    # the wrapper filters cold items out of the log first (base_rec.py:560-603): same recommendations as without the row
    pd.testing.assert_frame_equal(model.predict(bad, k=3, users=[0, 1]), model.predict(log, k=3, users=[0, 1]))
    # evaluate() takes a test-period log with unseen users and items
    test = log.sample(frac=0.1, random_state=2)[["user_idx", "item_idx"]]
    bad2 = pd.concat([bad, pd.DataFrame({"user_idx": [model._user_dim_size + 5], "item_idx": [1],
                                         "timestamp": [log.timestamp.max()], "relevance": [1.0]})], ignore_index=True)
    a, b = model.evaluate(bad2, test, ks=[5]), model.evaluate(log, test, ks=[5])
    assert a == b


def test_training_keeps_its_concurrency_whatever_the_process_did_before():
    """r3 finding (tools/probes/pipe_probe.hip, tools/stream_order_probe3.py; include/cqlrec.h, cqlrec_runtime_init): hardware
    queues sit on the 4 compute pipes in the order the process first used its streams, and two streams on one pipe cannot
    have kernels DISPATCHED side by side -- the training step (four streams) ran 1.33 instead of 0.69 ms at cfg3 when the
    model was constructed before the first kernel on the default stream, or when a torch.cuda.Stream() had been used
    between the log generation and the first step.  The library now picks its streams by test.  Fresh processes, three
    histories: the timed steps must cost the same."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    ms = {}
    for order in (("work", "core"), ("core", "work"), ("work", "tstream", "core"), ("tstream", "core", "work")):
        r = subprocess.run([sys.executable, str(root / "tools" / "stream_order_probe3.py"), *order], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        ms[" ".join(order)] = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    base = ms["work core"]
    assert all(v < 1.2 * base for v in ms.values()), ms

"""The evaluation oracle against the reference's OWN known answers (tests/test_metrics.py:45-100 fixtures `recs`, `true`,
`true_users`; expected values :181-305 and the NDCG doctest replay/metrics/ndcg.py:36-45).  These are data, re-typed."""
import numpy as np
import pytest

from oracle import metrics_oracle as M

RECS = [(0, 0, 3.0), (0, 1, 2.0), (0, 2, 1.0), (1, 0, 3.0), (1, 1, 4.0), (1, 4, 1.0), (2, 0, 5.0), (2, 2, 1.0), (2, 3, 2.0)]
TRUE = [(0, 0), (0, 4), (0, 1), (1, 5), (1, 0), (2, 1)]
TRUE_USERS = [1, 2, 3, 4]
L2 = np.log2

EXPECTED = {
    False: {
        "hitrate": {3: 2 / 3, 1: 1 / 3},
        "ndcg": {1: 1 / 3,
                 3: 1 / 3 * (1 / (1 / L2(2) + 1 / L2(3) + 1 / L2(4)) * (1 / L2(2) + 1 / L2(3))
                             + 1 / (1 / L2(2) + 1 / L2(3)) * (1 / L2(3)))},
        "precision": {1: 1 / 3, 3: (2 / 3 + 1 / 3) / 3},
        "map": {1: 1 / 3, 3: ((1 + 1) / 3 + (0 + 1 / 2) / 3) / 3},
        "recall": {1: 1 / 9, 3: (1 / 2 + 2 / 3) / 3},
    },
    True: {
        "hitrate": {3: 1 / 4, 1: 0.0},
        "ndcg": {1: 0.0, 3: 1 / 4 * (1 / (1 / L2(2) + 1 / L2(3)) * (1 / L2(3)))},
        "precision": {3: 1 / 4 * 1 / 3, 1: 0.0},
        "map": {1: 0.0, 3: 1 / 2 * 1 / 3 * 1 / 4},
        "recall": {1: 0.0, 3: 1 / 2 * 1 / 4},
    },
}


@pytest.mark.parametrize("gt_users", [False, True])
def test_reference_known_answers(gt_users):
    ru, ri, rr = zip(*RECS)
    gu, gi = zip(*TRUE)
    got = M.evaluate(ru, ri, rr, gu, gi, [1, 3], TRUE_USERS if gt_users else None)
    for metric, exp in EXPECTED[gt_users].items():
        for k, v in exp.items():
            assert got[metric][k] == pytest.approx(v, rel=1e-12, abs=1e-15), (metric, k)


def test_reference_edge_cases():
    one = ([1], [1], [1.0])
    two = ([1, 2], [1, 2], [1.0, 1.0])
    for m in ("ndcg", "hitrate", "precision", "recall", "map", "mrr"):
        assert M.evaluate(*one, two[0], two[1], [1])[m][1] == 0.5, m          # test_test_is_bigger (:169-171)
        assert M.evaluate(*two, one[0], one[1], [1])[m][1] == 1.0, m          # test_pred_is_bigger (:174-176)
    # NDCG doctest (replay/metrics/ndcg.py:36-45)
    got = M.evaluate([1, 1, 2, 2], [4, 5, 6, 7], [1, 1, 1, 1], [1, 1, 1, 1, 1, 2], [1, 2, 3, 4, 5, 8], [2])
    assert got["ndcg"][2] == pytest.approx(0.5)
    # duplicates: top max_k rows first, THEN unique items -> fewer than k predictions (base_metric.py:126-128)
    dup = M.evaluate([0, 0, 0], [1, 1, 2], [3.0, 2.0, 1.0], [0, 0], [1, 2], [2])
    assert dup["precision"][2] == 0.5
    for f in (M.ndcg, M.hitrate, M.precision, M.recall, M.mean_ap, M.mrr):
        assert f(4, [], [2, 4]) == 0                      # test_empty_recs (:373-381)
        assert f(4, [1, 3], [2, 4]) == 0                  # test_bad_recs (:384-392)
    for f in (M.ndcg, M.hitrate, M.recall, M.mrr):        # test_not_full_recs (:395-406)
        assert f(4, [4, 1, 2], [2, 4]) == f(3, [4, 1, 2], [2, 4])


def test_block_form_equals_dataframe_form():
    rng = np.random.default_rng(0)
    n, kmax, NI = 50, 7, 40
    rec = np.stack([rng.permutation(NI)[:kmax] for _ in range(n)]).astype(np.int32)
    rec[3, 4:] = -1
    rec[7, :] = -1
    cnt = rng.integers(0, 6, n)
    cnt[5] = 0
    off = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    gt = np.concatenate([np.sort(rng.choice(NI, c, replace=False)) for c in cnt]).astype(np.int32)
    ks = [1, 3, 7]
    blk = M.evaluate_block(rec, off, gt, ks)
    ru = np.repeat(np.arange(n), kmax)[rec.ravel() >= 0]
    ri = rec.ravel()[rec.ravel() >= 0]
    rr = np.tile(np.arange(kmax, 0, -1), n)[rec.ravel() >= 0].astype(float)
    gu = np.repeat(np.arange(n), cnt)
    df = M.evaluate(ru, ri, rr, gu, gt, ks, ground_truth_users=range(n))
    for mi, m in enumerate(M.METRICS):
        for ki, k in enumerate(ks):
            assert blk[:, mi, ki].mean() == pytest.approx(df[m][k], rel=1e-12)

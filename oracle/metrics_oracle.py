"""CPU oracle for the on-device evaluation row  --  TEST INFRASTRUCTURE ONLY.

Unlike the CQL arithmetic, THIS part is pinned by the reference: the per-user formulas restate
replay/metrics/{ndcg.py:50-59, hitrate.py:22-27, precision.py, recall.py, map.py, mrr.py}::_get_metric_value_by_user,
the user set follows get_enriched_recommendations (replay/metrics/base_metric.py:102-140: users of the ground truth,
or `ground_truth_users`; users without recommendations get an empty prediction list), and the known answers of
tests/test_metrics.py:181-305 are reproduced in tests/test_metrics_oracle.py."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

METRICS = ("ndcg", "hitrate", "precision", "recall", "map", "mrr")


def ndcg(k: int, pred: Sequence[int], gt: Sequence[int]) -> float:          # replay/metrics/ndcg.py:50-59
    if len(pred) == 0 or len(gt) == 0:
        return 0.0
    gts = set(gt)
    denom = [1 / math.log2(i + 2) for i in range(k)]
    dcg = sum(denom[i] for i in range(min(k, len(pred))) if pred[i] in gts)
    return dcg / sum(denom[: min(k, len(gt))])


def hitrate(k: int, pred, gt) -> float:                                       # replay/metrics/hitrate.py:22-27
    gts = set(gt)
    return 1.0 if any(i in gts for i in pred[:k]) else 0.0


def precision(k: int, pred, gt) -> float:                                     # replay/metrics/precision.py
    if len(pred) == 0:
        return 0.0
    return len(set(pred[:k]) & set(gt)) / k


def recall(k: int, pred, gt) -> float:                                        # replay/metrics/recall.py
    if len(gt) == 0:
        return 0.0
    return len(set(pred[:k]) & set(gt)) / len(gt)


def mean_ap(k: int, pred, gt) -> float:                                       # replay/metrics/map.py
    if len(gt) == 0 or len(pred) == 0:
        return 0.0
    gts, tp, res = set(gt), 0, 0.0
    for i in range(min(k, len(pred))):
        if pred[i] in gts:
            tp += 1
            res += tp / (i + 1)
    return res / k


def mrr(k: int, pred, gt) -> float:                                           # replay/metrics/mrr.py
    gts = set(gt)
    for i in range(min(k, len(pred))):
        if pred[i] in gts:
            return 1 / (1 + i)
    return 0.0


_FUNCS = {"ndcg": ndcg, "hitrate": hitrate, "precision": precision, "recall": recall, "map": mean_ap, "mrr": mrr}


def evaluate(rec_user, rec_item, rec_rel, gt_user, gt_item, ks: Iterable[int],
             ground_truth_users: Optional[Iterable[int]] = None) -> Dict[str, Dict[int, float]]:
    """{metric: {k: value}} averaged over the ground-truth users (base_metric.py:102-140 semantics)."""
    ks = list(ks)
    preds: Dict[int, List] = {}
    for u, i, r in zip(rec_user, rec_item, rec_rel):
        preds.setdefault(int(u), []).append((float(r), int(i)))
    gts: Dict[int, List[int]] = {}
    for u, i in zip(gt_user, gt_item):
        gts.setdefault(int(u), []).append(int(i))
    users = sorted(gts) if ground_truth_users is None else [int(u) for u in ground_truth_users]
    out = {m: {k: 0.0 for k in ks} for m in METRICS}
    for u in users:
        # sorter(): relevance desc, unique items (base_metric.py:22-51); top max_k first (get_top_k_recs)
        seen, pred = set(), []
        for r, i in sorted(preds.get(u, []), key=lambda t: -t[0])[: max(ks)]:
            if i not in seen:
                seen.add(i)
                pred.append(i)
        gt = gts.get(u, [])
        for m in METRICS:
            for k in ks:
                out[m][k] += _FUNCS[m](k, pred, gt)
    n = max(len(users), 1)
    return {m: {k: v / n for k, v in d.items()} for m, d in out.items()}


def evaluate_block(rec_idx: np.ndarray, gt_off: np.ndarray, gt_items: np.ndarray, ks: Sequence[int]) -> np.ndarray:
    """Array form used against the kernel: rec_idx [n_users x kmax] (-1 padded, unique per row), ground truth CSR
    (row u = user u).  Returns per-user values [n_users][6][len(ks)]."""
    n = rec_idx.shape[0]
    out = np.zeros((n, len(METRICS), len(ks)), dtype=np.float64)
    for u in range(n):
        pred = [int(x) for x in rec_idx[u] if x >= 0]
        gt = [int(x) for x in gt_items[gt_off[u]: gt_off[u + 1]]]
        for mi, m in enumerate(METRICS):
            for ki, k in enumerate(ks):
                out[u, mi, ki] = _FUNCS[m](k, pred, gt)
    return out

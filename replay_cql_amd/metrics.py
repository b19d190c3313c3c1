"""On-device evaluation of a top-k block: NDCG / HitRate / Precision / Recall / MAP / MRR @ k with the
reference's per-user formulas (replay/metrics/*.py) and user set (replay/metrics/base_metric.py:102-140), so that
optimize()-style loops (replay/optuna_objective.py:80-111) keep the U x k result on the GPU."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import _native as N

METRICS = ("NDCG", "HitRate", "Precision", "Recall", "MAP", "MRR")


def evaluate_topk(rec_idx: torch.Tensor, gt_offsets: torch.Tensor, gt_items: torch.Tensor, ks: Iterable[int],
                  rec_rows: Optional[torch.Tensor] = None, n_gt_users: Optional[int] = None,
                  return_per_user: bool = False):
    """rec_idx int32 [n x kmax] (-1 padded, best first); ground-truth CSR with ascending unique items per row; row u of
    rec_idx is evaluated against CSR row rec_rows[u] (default u).  Returns {metric: {k: mean over n_gt_users}}."""
    lib = N.load()
    ks = sorted(int(k) for k in ks)
    n, kmax = int(rec_idx.shape[0]), int(rec_idx.shape[1])
    dev = rec_idx.device
    rec_idx = rec_idx.to(torch.int32).contiguous()
    ks_arr = (C.c_int32 * len(ks))(*ks)
    ws_bytes = int(lib.cqlrec_eval_topk_ws_bytes(n, len(ks)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    sums = torch.zeros(len(METRICS) * len(ks), dtype=torch.float64, device=dev)
    per_user = torch.empty((n, len(METRICS), len(ks)), dtype=torch.float64, device=dev) if return_per_user else None
    N.check(lib.cqlrec_eval_topk(rec_idx.data_ptr(), n, kmax, None if rec_rows is None else rec_rows.data_ptr(),
                                 gt_offsets.data_ptr(), gt_items.data_ptr(), ks_arr, len(ks), ws.data_ptr(), ws_bytes,
                                 None if per_user is None else per_user.data_ptr(), sums.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), "eval_topk")
    denom = float(n if n_gt_users is None else n_gt_users)
    vals = (sums.cpu().numpy() / denom).reshape(len(METRICS), len(ks))
    out: Dict[str, Dict[int, float]] = {m: {k: float(vals[mi, ki]) for ki, k in enumerate(ks)}
                                        for mi, m in enumerate(METRICS)}
    return (out, per_user) if return_per_user else out

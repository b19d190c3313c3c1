#!/usr/bin/env python3
"""Measurement of the "next" rows of SURVEY 8(f) at scale, on one MI355X (inputs resident in HBM unless stated):
  f2  device CSR builder (cqlrec_build_csr): rows/s for a log of --rows interactions, sortedness properties checked
  f4  on-device evaluation (cqlrec_eval_topk): users/s for NDCG/HitRate/Precision/Recall/MAP/MRR @ {1,5,10}
  f1  Arrow ingest / egress (arrow_io): pyarrow Table -> pinned staging -> device columns, and a U x k block -> RecordBatch
      (host-bound: PCIe + one host copy; reported as rows/s INCLUDING the transfers)
    python tools/frows_bench.py [--rows 100000000] [--users 4000000] [--items 1000000]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import arrow_io as A  # noqa: E402
from replay_cql_amd.data import build_csr_device  # noqa: E402
from replay_cql_amd.metrics import evaluate_topk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--users", type=int, default=4_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    out = {"config": vars(a)}

    def timed(fn, reps=a.reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    # ---- f2
    n, U, NI = a.rows, a.users, a.items
    u = torch.randint(0, U, (n,), device=dev, generator=g, dtype=torch.int32)
    it = torch.randint(0, NI, (n,), device=dev, generator=g, dtype=torch.int32)
    ts = torch.randint(0, 1 << 40, (n,), device=dev, generator=g, dtype=torch.int64)
    rel = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
    dt, (off, items, rew) = timed(lambda: build_csr_device(u, it, ts, rel, U, device=dev, check=False))
    assert int(off[-1]) == n and bool((off[1:] >= off[:-1]).all())
    cnt = torch.bincount(u.to(torch.int64), minlength=U)
    assert torch.equal(off[1:] - off[:-1], cnt)
    # a user's items must be that user's items in timestamp order: check a sample of users on the host
    for uu in torch.randint(0, U, (20,)).tolist():
        m = (u == uu)
        order = torch.argsort(ts[m], stable=True)
        assert torch.equal(items[off[uu]: off[uu + 1]], it[m][order])
    out["f2_build_csr"] = {"rows": n, "users": U, "ms": 1e3 * dt, "rows_per_s": n / dt,
                           "algorithmic_bytes": n * (4 + 4 + 8 + 8) + n * 8 + U * 8,
                           "note": "user_idx,item_idx int32 + timestamp int64 + relevance f64 in; items int32 + rewards f32 out"}
    del ts, rel, rew
    # seen lists (timestamp=None form)
    dt2, _ = timed(lambda: build_csr_device(u, it, None, None, U, device=dev, check=False))
    out["f2_seen_lists"] = {"rows": n, "ms": 1e3 * dt2, "rows_per_s": n / dt2}

    # ---- f4: top-10 block of U users against a ground truth CSR (the log above, items ascending per user)
    gt_off, gt_items, _ = build_csr_device(u, it, None, None, U, device=dev, check=False)
    k = 10
    rec = torch.randint(0, NI, (U, k), device=dev, generator=g, dtype=torch.int32)
    rec[:, 0] = gt_items[gt_off[:-1].clamp(max=n - 1)]        # one hit for users with a non-empty ground truth
    dt3, res = timed(lambda: evaluate_topk(rec, gt_off, gt_items, (1, 5, 10)))
    out["f4_eval_topk"] = {"users": U, "k": k, "ms": 1e3 * dt3, "users_per_s": U / dt3,
                           "HitRate@10": res["HitRate"][10], "NDCG@10": res["NDCG"][10]}
    del rec, gt_off, gt_items

    # ---- f1: Arrow in / out (host side included)
    import pyarrow as pa
    m = min(n, 20_000_000)
    tab = pa.table({"user_idx": pa.array(u[:m].cpu().numpy()), "item_idx": pa.array(it[:m].cpu().numpy()),
                    "timestamp": pa.array(np.arange(m, dtype=np.int64)),
                    "relevance": pa.array(np.ones(m, dtype=np.float64))})
    dt4, cols = timed(lambda: A.columns_to_device(tab, dev), reps=2)
    out["f1_arrow_ingest"] = {"rows": m, "ms": 1e3 * dt4, "rows_per_s": m / dt4, "bytes": m * 24,
                              "GBps_including_PCIe": m * 24 / dt4 / 1e9}
    nu = 1_000_000
    users = torch.arange(nu, device=dev, dtype=torch.int32)
    idx = torch.randint(0, NI, (nu, k), device=dev, generator=g, dtype=torch.int32)
    val = torch.rand(nu, k, device=dev, generator=g)
    cntk = torch.full((nu,), k, device=dev, dtype=torch.int32)
    dt5, rb = timed(lambda: A.recs_to_arrow(users, idx, val, cntk), reps=2)
    out["f1_arrow_egress"] = {"users": nu, "k": k, "rows": int(rb.num_rows), "ms": 1e3 * dt5,
                              "rows_per_s": rb.num_rows / dt5}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

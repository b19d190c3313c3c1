"""Shared test helpers: device buffers for the C ABI, oracle model <-> device state."""
from __future__ import annotations

import numpy as np
import torch

from oracle import cql_oracle as O
from replay_cql_amd import _native as N

DEV = "cuda:0"
_KEEP = []   # device tensors handed to the C ABI as raw pointers must outlive the (asynchronous) call


def keep(t):
    _KEEP.append(t)
    return t


def release_kept():
    _KEEP.clear()


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return keep(t.to(DEV).contiguous())


def bf16_dev(x_f32: np.ndarray) -> torch.Tensor:
    """fp32 numpy (any values) -> device bf16 tensor holding oracle-rounded values."""
    bits = O.bf16_bits(np.asarray(x_f32, dtype=np.float32)).astype(np.int16)
    return keep(torch.as_tensor(bits).to(DEV).view(torch.bfloat16).contiguous())


def bf16_to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


def ws_bytes_tensor(nbytes: int) -> torch.Tensor:
    return torch.empty(int(nbytes), dtype=torch.uint8, device=DEV)


def small_log(U=64, N=257, seed=1, mean_len=12, max_len=40):
    u, i, t, r = O.synth_log(U, N, seed=seed, mean_len=mean_len, max_len=max_len)
    return O.build_csr(u, i, t, r, U)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ---- shared case builders of the Q-head / top-K kernel tests (small shapes: test_gpu_kernels; published shapes:
# test_gpu_fullsize) ----------------------------------------------------------------------------------------------
def qhead_inputs(rows, Nn, d, dyadic, seed):
    rng = np.random.default_rng(seed)
    if dyadic:
        H = (rng.integers(-8, 9, (rows, d)) / 8.0).astype(np.float32)
        E = (rng.integers(-8, 9, (Nn, d)) / 8.0).astype(np.float32)
        b = (rng.integers(-16, 17, Nn) / 8.0).astype(np.float32)
    else:
        H = rng.standard_normal((rows, d)).astype(np.float32)
        E = (rng.standard_normal((Nn, d)) / np.sqrt(d)).astype(np.float32)
        b = (rng.standard_normal(Nn) * 0.3).astype(np.float32)
    return O.bf16_round(H), O.bf16_round(E), b


def topk_rule_violations(idx, val, cnt, Q, k, set_tol=1e-4, val_tol=1e-3):
    """P3 for a block of top-k lists against the reference score matrix Q (rows = the same users, inadmissible items at
    -inf), PER ROW: the count of admissible items, descending order, every reported score within val_tol of the
    reference's score of that item, and the set equal to the reference's outside a set_tol margin around the k-th
    reference score (an item in one list only must score within set_tol of it).  Returns (rows that violate it,
    number of boundary swaps over the passing rows)."""
    bad, swaps = [], 0
    ar = np.arange(Q.shape[1])
    for u in range(Q.shape[0]):
        order = np.lexsort((ar, -Q[u].astype(np.float64)))[:k]
        rv = Q[u][order]
        c = int(np.isfinite(rv).sum())
        ok = int(cnt[u]) == c and np.all(np.diff(val[u, :c]) <= 0)
        if ok and c:
            got = idx[u, :c].astype(np.int64)
            ok = (np.unique(got).size == c and got.min() >= 0 and
                  np.all(np.abs(val[u, :c] - Q[u, got]) <= val_tol))
            if ok:
                diff = set(got.tolist()) ^ set(order[:c].tolist())
                ok = all(abs(float(Q[u, j]) - float(rv[c - 1])) < set_tol for j in diff)
                swaps += len(diff) // 2
        if not ok:
            bad.append(u)
    return np.asarray(bad, dtype=np.int64), swaps


def topk_case(lib, n_users, Nn, d, k, dyadic, seed, with_seen, cand=None, guard_bytes=0):
    Hb, Eb, b = qhead_inputs(n_users, Nn, d, dyadic, seed)
    rng = np.random.default_rng(seed)
    ids = np.arange(Nn, dtype=np.int32) if cand is None else cand
    E_c, b_c = Eb[ids], b[ids]
    Q = O.qvalues(Hb, E_c, b_c)
    seen_off = seen_items = None
    if with_seen:
        cnts = rng.integers(0, 40, n_users)
        cnts[0] = 0
        seen_off = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(cnts, out=seen_off[1:])
        rows = []
        for u in range(n_users):
            top = np.argsort(-Q[u])[: cnts[u] // 2]                   # half of the seen items are the best ones
            rnd = rng.integers(0, Nn, cnts[u] - len(top))
            row = np.unique(np.concatenate([ids[top], rnd]).astype(np.int32))
            rows.append(row)
            seen_off[u + 1] = seen_off[u] + len(row)
        seen_items = np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32)
        pos_of = -np.ones(Nn, dtype=np.int64)
        pos_of[ids] = np.arange(len(ids))
        for u in range(n_users):
            p = pos_of[seen_items[seen_off[u]: seen_off[u + 1]]]
            Q[u, p[p >= 0]] = -np.inf
    kk = min(k, len(ids))
    idx_c, val_ref = O.topk_rows(Q, kk)
    idx_ref = np.where(np.isfinite(val_ref), ids[idx_c], -1)

    nb = int(lib.cqlrec_topk_ws_bytes(n_users, len(ids), d, k))
    ws = ws_bytes_tensor(nb + guard_bytes)         # guard_bytes > 0: a poisoned region right behind the declared size
    if guard_bytes:
        ws[nb:] = 0xFF                             # "every item seen" if a live row ever read it as bitmap
    out_idx = torch.empty((n_users, k), dtype=torch.int32, device=DEV)
    out_val = torch.empty((n_users, k), dtype=torch.float32, device=DEV)
    out_cnt = torch.empty(n_users, dtype=torch.int32, device=DEV)
    d_ids = None if cand is None else dev(ids)
    d_so = None if seen_off is None else dev(seen_off)
    d_si = None if seen_items is None else dev(np.concatenate([seen_items, np.zeros(1, np.int32)]))
    N.check(lib.cqlrec_score_topk(ptr(bf16_dev(Hb)), n_users, ptr(bf16_dev(E_c)), ptr(dev(b_c)), len(ids), d, ptr(d_ids),
                                  ptr(d_so), ptr(d_si), None, k, ptr(ws), nb, ptr(out_idx), ptr(out_val), ptr(out_cnt),
                                  stream()))
    sync()
    if guard_bytes:
        assert bool((ws[nb:] == 0xFF).all()), "the pass wrote behind the workspace size it asked for"
    return out_idx.cpu().numpy(), out_val.cpu().numpy(), out_cnt.cpu().numpy(), idx_ref, val_ref, Q

"""qtopk4_kernel's two forms of the seen filter (qhead_topk4.hip): entry lists (one 256-byte slot per 128 users x 64
items + an overflow area, built in the bitmap's space) and -- when the lists do not fit -- the dense bitmap, picked on the
device by the word the list builder leaves.

What the reference does here: `_filter_seen` anti-joins the recommendations with the log (replay/models/base_rec.py:
417-464).  Both forms must give exactly that: dyadic operands, ids and scores bit-identical to the oracle's masked top-k.
Parity is UNPINNED by the reference (no CQL path there): the checker is this repo's oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd import _native as N

from helpers import DEV, bf16_dev, dev, ptr, qhead_inputs, stream, sync, ws_bytes_tensor

pytestmark = pytest.mark.gpu

NN, D_, K = 1024, 128, 10
N_USERS = 512 * 160 + 300          # the smallest launch that takes qtopk4_kernel by default, plus a partial block


def _case(kind):
    """light: a few seen items per user (every list fits its slot); popular: three items seen by nearly everybody (their
    stages' lists run into the overflow area); heavy: every user has seen half the catalogue (the lists do not fit)"""
    rng = np.random.default_rng({"light": 11, "popular": 12, "heavy": 13}[kind])
    Hb, Eb, b = qhead_inputs(4096, NN, D_, True, 5)
    n = N_USERS
    h_row = (7 * np.arange(n)) % 4096
    us = np.unique(np.concatenate([np.arange(0, 4), np.arange(n - 302, n), [127, 128, 511, 512, 65535, 65536],
                                   rng.integers(0, n, 400)]))
    Q = O.qvalues(Hb[h_row[us]], Eb, b)
    cnt = rng.integers(0, 8, n)
    cnt[0] = 0
    keys = [np.repeat(np.arange(n, dtype=np.int64), cnt) * NN + rng.integers(0, NN, int(cnt.sum()))]
    for j, u in enumerate(us):          # the users that are checked have seen some of their best items
        top = np.argsort(-Q[j], kind="stable")[: 3 + (u % 5)]
        keys.append(u * NN + top.astype(np.int64))
    if kind == "popular":
        for item in (3, 40, 700):       # items 3 and 40 share a stage: 2 x ~125 entries per wave in one list
            who = np.nonzero(rng.random(n) < 0.97)[0].astype(np.int64)
            keys.append(who * NN + item)
    if kind == "heavy":
        half = rng.permutation(NN)[: NN // 2].astype(np.int64)
        keys.append((np.arange(n, dtype=np.int64)[:, None] * NN + half[None, :]).ravel())
    key = np.unique(np.concatenate(keys))
    rows, items = key // NN, (key % NN).astype(np.int32)
    if kind == "popular":               # ids repeated in a list (still ascending) count once
        rep = np.where(rows < 300, 2, 1)
        rows, items = np.repeat(rows, rep), np.repeat(items, rep)
    seen_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=seen_off[1:])
    for j, u in enumerate(us):
        Q[j, items[seen_off[u]: seen_off[u + 1]]] = -np.inf
    ridx, rval = O.topk_rows(Q, K)
    return Hb[h_row], Eb, b, seen_off, items, us, ridx, rval


@pytest.mark.parametrize("kind,form", [("light", 0), ("popular", 0), ("heavy", 1)])
def test_topk4_seen_entry_lists_and_the_bitmap_fallback(kind, form):
    lib = N.load()
    H, Eb, b, seen_off, seen_items, us, ridx, rval = _case(kind)
    n = H.shape[0]
    nb = int(lib.cqlrec_topk_ws_bytes(n, NN, D_, K))
    ws = ws_bytes_tensor(nb + 4096)
    ws[nb:] = 0xFF
    out_idx = torch.empty((n, K), dtype=torch.int32, device=DEV)
    out_val = torch.empty((n, K), dtype=torch.float32, device=DEV)
    out_cnt = torch.empty(n, dtype=torch.int32, device=DEV)
    d_si = dev(np.concatenate([seen_items, np.zeros(1, np.int32)]))
    N.check(lib.cqlrec_score_topk(ptr(bf16_dev(H)), n, ptr(bf16_dev(Eb)), ptr(dev(b)), NN, D_, None, ptr(dev(seen_off)),
                                  ptr(d_si), None, K, ptr(ws), nb, ptr(out_idx), ptr(out_val), ptr(out_cnt), stream()))
    sync()
    assert bool((ws[nb:] == 0xFF).all()), "the pass wrote behind the workspace size it asked for"
    got = C.c_int32(-2)
    N.check(lib.cqlrec_topk_seen_form(ptr(ws), n, NN, D_, K, C.byref(got), stream()))
    assert got.value == form, f"{kind}: expected form {form} (0 = lists, 1 = bitmap), got {got.value}"
    assert np.array_equal(out_idx.cpu().numpy()[us], ridx)
    assert np.array_equal(out_val.cpu().numpy()[us], rval)
    assert np.all(out_cnt.cpu().numpy()[us] == K)

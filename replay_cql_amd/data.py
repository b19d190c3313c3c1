"""Host-side data plumbing of the hot path: LOG_SCHEMA rows -> CSR by user (S2), seen lists, synthetic logs.

Mirrors what `NeuroMF._fit` does with `log.toPandas()` + DataLoader construction (replay/models/neuromf.py:325-339);
the arithmetic of training never happens here."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch


_I64_LOW63 = 0x7FFFFFFFFFFFFFFF


def timestamp_key(timestamp):
    """LOG_SCHEMA `timestamp` column (datetime64, integer or float; numpy or torch) -> int64 keys whose signed order is
    the column's order.  Floats go through the IEEE total-order map (sign-magnitude -> two's complement), with -0.0
    folded onto +0.0, so fractional timestamps keep their order -- the SAME keys feed the host lexsort (build_csr) and
    the device radix sort (build_csr_device), which keeps fit and predict on one event order.  NaN is rejected."""
    if torch.is_tensor(timestamp):
        if not timestamp.dtype.is_floating_point:
            return timestamp.to(torch.int64)
        a = timestamp.to(torch.float64) + 0.0
        if bool(torch.isnan(a).any()):
            raise ValueError("timestamp contains NaN")
        b = a.view(torch.int64)
        return torch.where(b < 0, b ^ _I64_LOW63, b)
    a = np.asarray(timestamp)
    if a.dtype.kind == "M":
        return a.astype("datetime64[ns]").astype(np.int64)
    if a.dtype.kind in "iu":
        return a.astype(np.int64)
    if a.dtype.kind != "f":
        return a.astype("datetime64[ns]").astype(np.int64)
    a = a.astype(np.float64) + 0.0
    if np.isnan(a).any():
        raise ValueError("timestamp contains NaN")
    b = a.view(np.int64)
    return np.where(b < 0, b ^ np.int64(_I64_LOW63), b)


def build_csr(user_idx, item_idx, timestamp, relevance, n_users: Optional[int] = None):
    """Sort by (user, timestamp asc, item_idx asc) -> offsets int64[U+1], items int32[nnz], rewards float32[nnz]."""
    user_idx = np.asarray(user_idx, dtype=np.int64)
    item_idx = np.asarray(item_idx, dtype=np.int64)
    timestamp = timestamp_key(timestamp)
    relevance = np.asarray(relevance, dtype=np.float64)
    if not (len(user_idx) == len(item_idx) == len(timestamp) == len(relevance)):
        raise ValueError("log columns differ in length")
    if len(user_idx) and (user_idx.min() < 0 or item_idx.min() < 0):
        raise ValueError("user_idx / item_idx must be non-negative dense indices")
    if n_users is None:
        n_users = int(user_idx.max()) + 1 if len(user_idx) else 0
    order = np.lexsort((item_idx, timestamp, user_idx))
    counts = np.bincount(user_idx, minlength=n_users).astype(np.int64)
    offsets = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets, item_idx[order].astype(np.int32), relevance[order].astype(np.float32)


def sorted_seen(offsets: np.ndarray, items: np.ndarray) -> np.ndarray:
    """Items sorted ascending inside every CSR row (same offsets): the `seen` lists of cqlrec_score_topk."""
    rows = np.repeat(np.arange(len(offsets) - 1, dtype=np.int64), np.diff(offsets))
    order = np.lexsort((items, rows))
    return items[order].astype(np.int32)


# ----------------------------------------------------------------------------------------------------------
# synthetic logs on device: a deterministic function of (seed, user, position) so that any user
# shard is generated identically at any world size.  Benchmark input plumbing, not part of the hot path.
# ----------------------------------------------------------------------------------------------------------
_M1, _M2, _GOLD = -0x40A7B892E31B1A47, -0x6B2FB644ECCEEE15, -0x61C8864680B583EB  # splitmix64 constants as int64


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix64(z: torch.Tensor) -> torch.Tensor:
    z = z + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _u01(h: torch.Tensor) -> torch.Tensor:
    return (_lsr(h, 11).to(torch.float64) + 0.5) * (1.0 / (1 << 53))


def synth_log_device(n_users: int, n_items: int, seed: int = 12345, device="cuda", user_lo: int = 0,
                     user_hi: Optional[int] = None, mean_len: float = 40.0, sigma: float = 0.6, min_len: int = 5,
                     max_len: int = 200, zipf: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR (offsets, items, rewards) of users [user_lo, user_hi) generated on `device`.

    len_u = clip(round(lognormal(ln mean_len, sigma)), min_len, max_len); item ~ Zipf(1) through a fixed affine
    permutation (uniform if zipf=False); reward in {0.2,...,1.0}; timestamp = position (rows are already ordered)."""
    user_hi = n_users if user_hi is None else user_hi
    dev = torch.device(device)
    u = torch.arange(user_lo, user_hi, dtype=torch.int64, device=dev)
    hu = _mix64(u ^ (seed * 0x2545F491))
    # Box-Muller for the log-normal length
    n1 = torch.sqrt(-2.0 * torch.log(_u01(hu))) * torch.cos(2.0 * math.pi * _u01(_mix64(hu)))
    lens = torch.clamp(torch.round(torch.exp(math.log(mean_len) + sigma * n1)), min_len, max_len).to(torch.int64)
    offsets = torch.zeros(user_hi - user_lo + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=offsets[1:])
    nnz = int(offsets[-1])
    row = torch.repeat_interleave(torch.arange(user_hi - user_lo, device=dev), lens)
    pos = torch.arange(nnz, dtype=torch.int64, device=dev) - offsets[row]
    hp = _mix64(hu[row] + pos * 0x9E3779B1)
    if zipf:
        r = torch.floor(torch.exp(_u01(hp) * math.log(n_items + 1.0))).to(torch.int64).clamp_(1, n_items) - 1
    else:
        r = (_lsr(hp, 1) % n_items)
    mult = 2654435761 % n_items
    while math.gcd(mult, n_items) != 1:
        mult += 1
    items = ((r * mult + 12345) % n_items).to(torch.int32)
    rewards = ((_lsr(_mix64(hp), 3) % 5 + 1).to(torch.float32)) * 0.2
    return offsets, items.contiguous(), rewards.contiguous()


def _dev_col(x, dt, dev):
    if torch.is_tensor(x):
        return x.to(device=dev, dtype=dt).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x))).to(device=dev, dtype=dt).contiguous()


def build_csr_device(user_idx, item_idx, timestamp, relevance, n_users: int, device="cuda", check: bool = True):
    """build_csr on the GPU (cqlrec_build_csr: stable rocPRIM radix sorts + boundary scan); returns device tensors
    (offsets int64[U+1], items int32[nnz], rewards float32[nnz]).  Bit-identical to build_csr().

    timestamp=None: rows ordered by (user, item asc) -- the `seen` lists of cqlrec_score_topk (sorted_seen() on the
    device).  relevance=None: no reward column (returns rewards=None)."""
    from . import _native as N
    lib = N.load()
    dev = torch.device(device)
    u, i = _dev_col(user_idx, torch.int32, dev), _dev_col(item_idx, torch.int32, dev)
    t = None if timestamp is None else _dev_col(timestamp_key(timestamp), torch.int64, dev)
    r = None if relevance is None else _dev_col(relevance, torch.float64, dev)
    n = u.numel()
    if n != i.numel() or (t is not None and t.numel() != n) or (r is not None and r.numel() != n):
        raise ValueError("log columns differ in length")
    offsets = torch.zeros(n_users + 1, dtype=torch.int64, device=dev)
    items = torch.empty(n, dtype=torch.int32, device=dev)
    rewards = None if r is None else torch.empty(n, dtype=torch.float32, device=dev)
    if n == 0:
        return offsets, items, rewards
    if check:
        lo = torch.minimum(u.min(), i.min())
        if int(lo) < 0 or int(u.max()) >= n_users:
            raise ValueError("user_idx / item_idx must be non-negative dense indices below n_users")
    ws_bytes = int(lib.cqlrec_build_csr_ws_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    N.check(lib.cqlrec_build_csr(u.data_ptr(), i.data_ptr(), None if t is None else t.data_ptr(),
                                 None if r is None else r.data_ptr(), n, n_users, ws.data_ptr(), ws_bytes,
                                 offsets.data_ptr(), items.data_ptr(), None if rewards is None else rewards.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), "build_csr")
    return offsets, items, rewards

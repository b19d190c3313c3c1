"""CPU oracle for the CQL recommender hot path  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference snapshot (monkey0head/RePlay_cql @ 2025-02-28, replay-rec 0.10.0)
contains no CQL model, no d3rlpy dependency and no golden vector for this path (SURVEY.md F1/F2,
section 8(c)).  This file is therefore a CPU restatement of the *normative specification* in
SURVEY.md section 8.0 (S1-S7, P1-P4), not of reference code.  Where the reference does define
behaviour (schemas, wrapper semantics, top-k ordering) the function cites it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product path (replay_cql_amd) never imports it and fails loudly when the HIP library is missing.

Numerics (mirrored bit-for-bit in DESIGN.md section "Numerics"):
  * every matmul operand is rounded to bf16 (round-to-nearest-even) and accumulated in fp32;
  * state vector  h0 = (sum of the window's bf16 E_in rows, window order, fp32) / len   (len 0 -> 0);
  * encoder       a1 = bf16(h0) W1b^T + b1 ; z = relu(a1) ; h = bf16(z) W2b^T + b2      (W stored [out][in]);
  * Q-head        Q[j] = <bf16(h), E_out_b[j]> + b_out[j];
  * loss          L = mean_b[ 0.5 (q_a - y)^2 + alpha (lse - q_a) ],  y = r + gamma (1-done) Q_target(s', argmax_j Q(s', j));
  * backward      dQ_dense = (alpha/B) bf16(exp(Q - lse)) for the two gradient GEMMs (dH, dE_out),
                  unrounded exp(Q - lse) for db_out; straight-through across every bf16 rounding;
                  encoder backward in fp32 on the bf16-valued forward operands;
  * Adam          exactly the expression sequence of adam_ema_step() below (no fma contraction).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

MASK64 = (1 << 64) - 1
SEG_ALIGN = 64  # every parameter segment starts on a multiple of 64 elements


# --------------------------------------------------------------------------------------
# bf16 helpers
# --------------------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 (RNE) and return the value as fp32 (P1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(x.shape)


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit pattern (uint16)."""
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


# --------------------------------------------------------------------------------------
# parameter layout (S3)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Layout:
    """Flat parameter buffer layout shared by theta, grads, Adam moments, target and bf16 shadows."""

    n_items: int
    d: int
    off: Dict[str, int] = field(default_factory=dict)
    total: int = 0

    @staticmethod
    def make(n_items: int, d: int) -> "Layout":
        segs = [
            ("E_in", (n_items + 1) * d),
            ("E_out", n_items * d),
            ("b_out", n_items),
            ("W1", d * d),
            ("b1", d),
            ("W2", d * d),
            ("b2", d),
        ]
        off, cur = {}, 0
        for name, size in segs:
            off[name] = cur
            cur += -(-size // SEG_ALIGN) * SEG_ALIGN
        return Layout(n_items, d, off, cur)

    def shape(self, name: str) -> Tuple[int, ...]:
        n, d = self.n_items, self.d
        return {
            "E_in": (n + 1, d),
            "E_out": (n, d),
            "b_out": (n,),
            "W1": (d, d),
            "b1": (d,),
            "W2": (d, d),
            "b2": (d,),
        }[name]

    def view(self, flat: np.ndarray, name: str) -> np.ndarray:
        shp = self.shape(name)
        size = int(np.prod(shp))
        return flat[self.off[name] : self.off[name] + size].reshape(shp)

    @property
    def n_params(self) -> int:
        """Algorithmic parameter count P = (2N+1)d + N + 2d^2 + 2d."""
        n, d = self.n_items, self.d
        return (2 * n + 1) * d + n + 2 * d * d + 2 * d


def init_params(layout: Layout, seed: int = 7, dyadic: bool = False) -> np.ndarray:
    """E_in,E_out ~ N(0,1/d); W ~ xavier_normal; biases 0.  PAD row of E_in = 0.

    dyadic=True draws multiples of 2^-6 with |x| <= 2 (P2 exactness fixtures)."""
    rng = np.random.default_rng(seed)
    n, d = layout.n_items, layout.d
    flat = np.zeros(layout.total, dtype=np.float32)

    def draw(shape, std):
        if dyadic:
            return (rng.integers(-16, 17, size=shape) / 64.0).astype(np.float32)
        return (rng.standard_normal(shape) * std).astype(np.float32)

    layout.view(flat, "E_in")[:n] = draw((n, d), 1.0 / math.sqrt(d))
    layout.view(flat, "E_out")[:] = draw((n, d), 1.0 / math.sqrt(d))
    xav = math.sqrt(2.0 / (d + d))
    layout.view(flat, "W1")[:] = draw((d, d), xav)
    layout.view(flat, "W2")[:] = draw((d, d), xav)
    if dyadic:
        layout.view(flat, "b_out")[:] = draw((n,), 0.0)
        layout.view(flat, "b1")[:] = draw((d,), 0.0)
        layout.view(flat, "b2")[:] = draw((d,), 0.0)
    return flat


# --------------------------------------------------------------------------------------
# S1/S2: log -> CSR
# --------------------------------------------------------------------------------------
def build_csr(user_idx, item_idx, timestamp, relevance, n_users: Optional[int] = None):
    """Sort the LOG_SCHEMA rows (replay/constants.py:16-23) by (user, timestamp asc, item_idx asc) -> CSR (S2)."""
    user_idx = np.asarray(user_idx, dtype=np.int64)
    item_idx = np.asarray(item_idx, dtype=np.int64)
    timestamp = np.asarray(timestamp)
    if timestamp.dtype.kind == "M":
        timestamp = timestamp.astype("datetime64[ns]").astype(np.int64)
    relevance = np.asarray(relevance, dtype=np.float64)
    if n_users is None:
        n_users = int(user_idx.max()) + 1 if len(user_idx) else 0
    order = np.lexsort((item_idx, timestamp, user_idx))
    u = user_idx[order]
    counts = np.bincount(u, minlength=n_users).astype(np.int64)
    offsets = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets, item_idx[order].astype(np.int32), relevance[order].astype(np.float32)


# --------------------------------------------------------------------------------------
# counter-based transition sampler (integer arithmetic: GPU must match bit-for-bit)
# --------------------------------------------------------------------------------------
def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def sample_positions(seed: int, step: int, slot0: int, batch: int, nnz: int) -> np.ndarray:
    """Flat transition index p in [0,nnz) for slots slot0..slot0+batch-1 of global step `step`.

    k1 = mix64(seed ^ (step * 0xD1B54A32D192ED03)); k2 = mix64(k1 + slot); p = mulhi64(k2, nnz)."""
    k1 = _mix64((seed ^ ((step * 0xD1B54A32D192ED03) & MASK64)) & MASK64)
    out = np.empty(batch, dtype=np.int64)
    for b in range(batch):
        k2 = _mix64((k1 + slot0 + b) & MASK64)
        out[b] = (k2 * nnz) >> 64
    return out


def positions_to_transitions(pos: np.ndarray, offsets: np.ndarray):
    """p -> (user, t): user = last u with offsets[u] <= p."""
    users = np.searchsorted(offsets, pos, side="right") - 1
    t = pos - offsets[users]
    return users.astype(np.int32), t.astype(np.int32)


# --------------------------------------------------------------------------------------
# S4 forward pieces
# --------------------------------------------------------------------------------------
def gather_pool(E_in_b: np.ndarray, offsets, items, users, ends, L: int):
    """h0 for states (user, end): window = items[off+end-len .. off+end), len = min(end, L).  Returns (h0, len)."""
    n, d = len(users), E_in_b.shape[1]
    h0 = np.zeros((n, d), dtype=np.float32)
    lens = np.minimum(np.asarray(ends, dtype=np.int64), L).astype(np.int32)
    for i in range(n):
        ln = int(lens[i])
        if ln == 0:
            continue
        base = int(offsets[users[i]]) + int(ends[i])
        acc = np.zeros(d, dtype=np.float32)
        for it in items[base - ln : base]:
            acc = acc + E_in_b[it]
        h0[i] = acc / np.float32(ln)
    return h0, lens


def gather_pool_fast(E_in_b, offsets, items, users, ends, L: int):
    """Vectorised gather_pool (same sums up to fp32 summation order); used for the timed CPU baseline."""
    users = np.asarray(users, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    n, d = len(users), E_in_b.shape[1]
    lens = np.minimum(ends, L)
    base = offsets[users] + ends
    h0 = np.zeros((n, d), dtype=np.float32)
    for j in range(1, int(lens.max(initial=0)) + 1):
        m = lens >= j
        h0[m] += E_in_b[items[base[m] - lens[m] + (j - 1)]]
    nz = lens > 0
    h0[nz] /= lens[nz, None].astype(np.float32)
    return h0, lens.astype(np.int32)


def encoder_fwd(h0, W1b, b1, W2b, b2):
    h0b = bf16_round(h0)
    a1 = h0b @ W1b.T + b1
    z = np.maximum(a1, np.float32(0))
    zb = bf16_round(z)
    h = zb @ W2b.T + b2
    hb = bf16_round(h)
    return h0b, zb, h.astype(np.float32), hb


def qvalues(hb, E_out_b, b_out):
    """Full-catalog Q (materialised; the oracle may, the kernels never do)."""
    return (hb @ E_out_b.T + b_out).astype(np.float32)


def logsumexp_rows(Q):
    m = Q.max(axis=1)
    return (m + np.log(np.exp(Q - m[:, None]).sum(axis=1, dtype=np.float32))).astype(np.float32)


def argmax_rows(Q):
    """Ties -> smallest j (S5); numpy argmax returns the first maximum."""
    return Q.argmax(axis=1).astype(np.int32)


def argmax_margin_violations(a_got, a_ref, q_max_ref, hb_ref, E_out_b, b_out, hb_got=None, tol=1e-4):
    """P3 for the double-Q arg-max, per row (not as a rate): rows b with a_got[b] != a_ref[b] that are NOT excused.

    A differing row is excused only if the item chosen is a near-tie: Q_ref[b, a_got[b]] >= max_j Q_ref[b, j] - tol.
    Where the checked path's own bf16 state vector differs from the oracle's (hb_got given: a one-ulp flip of a bf16
    rounding upstream of the Q-head), the row is judged on the scores of THAT vector instead -- its choice must be a
    near-tie of its own row maximum -- so no row can be wrong by more than tol under either reading.
    Returns the array of violating row indices (empty = pass)."""
    a_got, a_ref = np.asarray(a_got, dtype=np.int64), np.asarray(a_ref, dtype=np.int64)
    bad = []
    for b in np.nonzero(a_got != a_ref)[0]:
        q_sel = np.float32(hb_ref[b] @ E_out_b[a_got[b]] + b_out[a_got[b]])
        if q_sel >= q_max_ref[b] - np.float32(tol):
            continue
        if hb_got is not None and not np.array_equal(hb_got[b], hb_ref[b]):
            row = (E_out_b @ hb_got[b] + b_out).astype(np.float32)
            if row[a_got[b]] >= row.max() - np.float32(tol):
                continue
        bad.append(int(b))
    return np.asarray(bad, dtype=np.int64)


# --------------------------------------------------------------------------------------
# S5 + backward: one training step's gradients
# --------------------------------------------------------------------------------------
@dataclass
class StepOut:
    loss: float
    grads: np.ndarray
    q_a: np.ndarray
    lse: np.ndarray
    a_star: np.ndarray
    q_targ: np.ndarray
    y: np.ndarray
    h0_s: np.ndarray
    hb_s: np.ndarray
    hb_sn: np.ndarray
    dH: np.ndarray
    dh0: np.ndarray
    qn_max: Optional[np.ndarray] = None   # max_j Q_theta(s', j): what an arg-max that differs from a_star is judged by


def shadow(flat: np.ndarray) -> np.ndarray:
    return bf16_round(flat)


def encode_states(layout: Layout, theta_b, theta, offsets, items, users, ends, L, fast=False):
    gp = gather_pool_fast if fast else gather_pool
    h0, lens = gp(layout.view(theta_b, "E_in"), offsets, items, users, ends, L)
    h0b, zb, h, hb = encoder_fwd(
        h0,
        layout.view(theta_b, "W1"),
        layout.view(theta, "b1"),
        layout.view(theta_b, "W2"),
        layout.view(theta, "b2"),
    )
    return h0, lens, h0b, zb, h, hb


def loss_and_grads(
    layout: Layout,
    theta: np.ndarray,
    target: np.ndarray,
    offsets,
    items,
    rewards,
    users,
    tpos,
    L: int,
    gamma: float,
    alpha: float,
    fast: bool = False,
    grad_scale_batch: Optional[int] = None,
) -> StepOut:
    """S4+S5 and the analytic gradient.

    grad_scale_batch: the batch the mean is taken over (global batch under data parallelism); default B."""
    B = len(users)
    Bg = B if grad_scale_batch is None else grad_scale_batch
    users = np.asarray(users, dtype=np.int64)
    tpos = np.asarray(tpos, dtype=np.int64)
    theta_b, target_b = shadow(theta), shadow(target)
    cnt = offsets[users + 1] - offsets[users]
    act = items[offsets[users] + tpos].astype(np.int64)
    rew = rewards[offsets[users] + tpos].astype(np.float32)
    done = (tpos == cnt - 1).astype(np.float32)

    h0_s, len_s, h0b_s, zb_s, _, hb_s = encode_states(layout, theta_b, theta, offsets, items, users, tpos, L, fast)
    _, _, _, _, _, hb_sn = encode_states(layout, theta_b, theta, offsets, items, users, tpos + 1, L, fast)
    _, _, _, _, _, hb_tn = encode_states(layout, target_b, target, offsets, items, users, tpos + 1, L, fast)

    E_out_b, b_out = layout.view(theta_b, "E_out"), layout.view(theta, "b_out")
    Et_b, bt = layout.view(target_b, "E_out"), layout.view(target, "b_out")

    Q_s = qvalues(hb_s, E_out_b, b_out)
    lse = logsumexp_rows(Q_s)
    q_a = Q_s[np.arange(B), act]
    Q_n = qvalues(hb_sn, E_out_b, b_out)
    a_star = argmax_rows(Q_n)
    qn_max = Q_n.max(axis=1).astype(np.float32)
    del Q_n
    q_targ = (np.einsum("bd,bd->b", hb_tn, Et_b[a_star], dtype=np.float32) + bt[a_star]).astype(np.float32)
    y = (rew + np.float32(gamma) * (np.float32(1) - done) * q_targ).astype(np.float32)
    delta = q_a - y
    loss_b = np.float32(0.5) * delta * delta + np.float32(alpha) * (lse - q_a)
    loss = float(np.sum(loss_b, dtype=np.float64) / Bg)

    # ---- backward -------------------------------------------------------------------
    P = np.exp(Q_s - lse[:, None]).astype(np.float32)
    del Q_s
    Pb = bf16_round(P)
    coef = ((delta - np.float32(alpha)) / np.float32(Bg)).astype(np.float32)
    s = np.float32(alpha / Bg)

    grads = np.zeros(layout.total, dtype=np.float32)
    g_Eout, g_bout = layout.view(grads, "E_out"), layout.view(grads, "b_out")
    g_Eout[:] = s * (Pb.T @ hb_s)
    g_bout[:] = s * P.sum(axis=0, dtype=np.float32)
    np.add.at(g_Eout, act, coef[:, None] * hb_s)
    np.add.at(g_bout, act, coef)
    dH = (s * (Pb @ E_out_b) + coef[:, None] * E_out_b[act]).astype(np.float32)
    del P, Pb

    W1b, W2b = layout.view(theta_b, "W1"), layout.view(theta_b, "W2")
    layout.view(grads, "b2")[:] = dH.sum(axis=0, dtype=np.float32)
    layout.view(grads, "W2")[:] = dH.T @ zb_s
    dA1 = (dH @ W2b) * (zb_s > 0)
    layout.view(grads, "b1")[:] = dA1.sum(axis=0, dtype=np.float32)
    layout.view(grads, "W1")[:] = dA1.T @ h0b_s
    dh0 = (dA1 @ W1b).astype(np.float32)

    g_Ein = layout.view(grads, "E_in")
    nz = len_s > 0
    contrib = np.zeros_like(dh0)
    contrib[nz] = dh0[nz] / len_s[nz, None].astype(np.float32)
    base = offsets[users] + tpos
    for j in range(1, int(len_s.max(initial=0)) + 1):
        m = len_s >= j
        np.add.at(g_Ein, items[base[m] - len_s[m] + (j - 1)], contrib[m])

    return StepOut(loss, grads, q_a, lse, a_star, q_targ, y, h0_s, hb_s, hb_sn, dH, dh0, qn_max)


# --------------------------------------------------------------------------------------
# S6 Adam + Polyak target
# --------------------------------------------------------------------------------------
def adam_scalars(step_t: int, lr: float, beta1: float, beta2: float):
    """step_t is 1-based.  Host-side double -> float scalars handed to the kernel."""
    bc1 = 1.0 - beta1**step_t
    bc2 = 1.0 - beta2**step_t
    return np.float32(lr / bc1), np.float32(math.sqrt(bc2))


def adam_ema_step(theta, grads, m, v, target, step_t, lr, beta1=0.9, beta2=0.999, eps=1e-8, tau=0.005):
    """In-place Adam on fp32 masters + Polyak target (S6).  Expression order is normative."""
    f = np.float32
    step_size, sqrt_bc2 = adam_scalars(step_t, lr, beta1, beta2)
    b1, b2, e, t = f(beta1), f(beta2), f(eps), f(tau)
    omb1, omb2, omt = f(1) - b1, f(1) - b2, f(1) - t  # fp32 subtraction, as the kernel does
    m[:] = b1 * m + omb1 * grads
    v[:] = b2 * v + (omb2 * grads) * grads
    denom = np.sqrt(v) / sqrt_bc2 + e
    theta[:] = theta - step_size * (m / denom)
    target[:] = omt * target + t * theta
    return theta, m, v, target


# --------------------------------------------------------------------------------------
# S7 predict / top-K
# --------------------------------------------------------------------------------------
def topk_rows(scores: np.ndarray, k: int):
    """Top-k per row by (score desc, index asc) -- the tie rule fixed in SURVEY F7 / 8.0 S7."""
    n = scores.shape[1]
    k = min(k, n)
    idx = np.empty((scores.shape[0], k), dtype=np.int32)
    val = np.empty((scores.shape[0], k), dtype=np.float32)
    ar = np.arange(n)
    for r in range(scores.shape[0]):
        order = np.lexsort((ar, -scores[r].astype(np.float64)))[:k]
        idx[r], val[r] = order, scores[r][order]
    return idx, val


def predict_topk(
    layout: Layout,
    theta,
    offsets,
    items,
    users,
    k: int,
    L: int,
    filter_seen: bool = True,
    cand_items: Optional[np.ndarray] = None,
    fast: bool = False,
):
    """S7: state = last L items of each user in the passed CSR; seen -> -inf; returns (idx, val, count).

    Rows whose number of admissible items is < k are padded with idx -1 / val -inf; count gives the valid prefix.
    Mirrors the semantics of _predict + _filter_seen + get_top_k_recs (replay/models/base_rec.py:417-464,
    :514-528; replay/utils.py:112-127) with the tie rule made explicit."""
    theta_b = shadow(theta)
    users = np.asarray(users, dtype=np.int64)
    cnt = offsets[users + 1] - offsets[users]
    _, _, _, _, _, hb = encode_states(layout, theta_b, theta, offsets, items, users, cnt, L, fast)
    E_out_b, b_out = layout.view(theta_b, "E_out"), layout.view(theta, "b_out")
    if cand_items is None:
        cand_items = np.arange(layout.n_items, dtype=np.int64)
    cand_items = np.asarray(cand_items, dtype=np.int64)
    Q = qvalues(hb, E_out_b[cand_items], b_out[cand_items])
    if filter_seen:
        pos_of = -np.ones(layout.n_items, dtype=np.int64)
        pos_of[cand_items] = np.arange(len(cand_items))
        for r, u in enumerate(users):
            seen = pos_of[items[offsets[u] : offsets[u + 1]]]
            Q[r, seen[seen >= 0]] = -np.inf
    kk = min(k, len(cand_items))
    idx_c, val = topk_rows(Q, kk)
    valid = np.isfinite(val)
    idx = np.where(valid, cand_items[idx_c], -1).astype(np.int32)
    val = np.where(valid, val, -np.inf).astype(np.float32)
    if kk < k:
        idx = np.pad(idx, ((0, 0), (0, k - kk)), constant_values=-1)
        val = np.pad(val, ((0, 0), (0, k - kk)), constant_values=-np.inf)
    return idx, val, valid.sum(axis=1).astype(np.int32), hb


def predict_pairs(layout: Layout, theta, offsets, items, users, pair_items, L: int):
    """relevance for explicit (user, item) pairs (base_rec.py:784-823 semantics; a11)."""
    theta_b = shadow(theta)
    users = np.asarray(users, dtype=np.int64)
    cnt = offsets[users + 1] - offsets[users]
    _, _, _, _, _, hb = encode_states(layout, theta_b, theta, offsets, items, users, cnt, L)
    E_out_b, b_out = layout.view(theta_b, "E_out"), layout.view(theta, "b_out")
    pair_items = np.asarray(pair_items, dtype=np.int64)
    return (np.einsum("bd,bd->b", hb, E_out_b[pair_items], dtype=np.float32) + b_out[pair_items]).astype(np.float32)


# --------------------------------------------------------------------------------------
# synthetic logs: deterministic in (seed, user)
# --------------------------------------------------------------------------------------
def synth_log(n_users: int, n_items: int, seed: int = 12345, mean_len: float = 40.0, sigma: float = 0.6,
              min_len: int = 5, max_len: int = 200, zipf: bool = True, dyadic_rewards: bool = True):
    """Small-scale host generator (tests/fixtures).  bench.py uses the device generator in replay_cql_amd.data."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.lognormal(math.log(mean_len), sigma, n_users)), min_len, max_len).astype(np.int64)
    nnz = int(lens.sum())
    if zipf:
        w = 1.0 / np.arange(1, n_items + 1)
        perm = rng.permutation(n_items)
        it = perm[rng.choice(n_items, size=nnz, p=w / w.sum())]
    else:
        it = rng.integers(0, n_items, nnz)
    users = np.repeat(np.arange(n_users), lens)
    ts = np.concatenate([np.arange(l) for l in lens]) if n_users else np.zeros(0, np.int64)
    rel = rng.integers(1, 6, nnz) / 5.0 if not dyadic_rewards else rng.integers(1, 5, nnz) / 4.0
    return users.astype(np.int32), it.astype(np.int32), ts.astype(np.int64), rel.astype(np.float64)


# --------------------------------------------------------------------------------------
# a tiny trainer (drives n steps exactly as replay_cql_amd.core does)
# --------------------------------------------------------------------------------------
@dataclass
class OracleModel:
    layout: Layout
    theta: np.ndarray
    target: np.ndarray
    m: np.ndarray
    v: np.ndarray
    step: int = 0

    @staticmethod
    def create(n_items, d, seed=7, dyadic=False):
        lay = Layout.make(n_items, d)
        th = init_params(lay, seed, dyadic)
        return OracleModel(lay, th, th.copy(), np.zeros_like(th), np.zeros_like(th))


def train_steps(model: OracleModel, offsets, items, rewards, n_steps, B, L, seed=0, gamma=0.99, alpha=1.0,
                lr=1e-3, tau=0.005, fast=False, rank=0, world=1):
    losses = []
    nnz = int(offsets[-1])
    for _ in range(n_steps):
        pos = sample_positions(seed, model.step, rank * B, B, nnz)
        users, tpos = positions_to_transitions(pos, offsets)
        out = loss_and_grads(model.layout, model.theta, model.target, offsets, items, rewards, users, tpos, L,
                             gamma, alpha, fast=fast, grad_scale_batch=B * world)
        adam_ema_step(model.theta, out.grads, model.m, model.v, model.target, model.step + 1, lr, tau=tau)
        model.step += 1
        losses.append(out.loss)
    return losses

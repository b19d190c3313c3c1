"""GPU parity AT THE PUBLISHED SHAPES (BASELINE.json configs): the HIP step and the top-K pass against the oracle at
cfg3 (N=100 000, d=128, B=4096, L=50), cfg2 (N=10 000, d=64, B=4096, L=50), a d=256 catalogue large enough for the
direct-output mode of the item-side backward (N >= 65 536), and cfg5's per-GPU shape (N=1 000 000, d=256) checked
through size-independent properties against a chunked fp32 torch restatement (the numpy oracle would need 16 GB per
score matrix there).  The synthetic log is the one bench.py trains on (data.synth_log_device, seed 12345), cut to a
20 000-user shard so that the oracle finishes a step in seconds.

Parity is UNPINNED by the reference (it has no CQL path): the checker is this repo's own oracle."""
import numpy as np
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd import _native as N
from replay_cql_amd.core import CQLCore, CQLHyper
from replay_cql_amd.data import synth_log_device

from helpers import (DEV, bf16_dev, bf16_to_np, dev, ptr, qhead_inputs, rel_err, stream, sync, topk_case,
                     ws_bytes_tensor)

pytestmark = pytest.mark.gpu

SEGS = ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2")


@pytest.fixture(scope="module")
def lib():
    return N.load()


def _oracle_model(Nn, d, seed=7):
    m = O.OracleModel.create(Nn, d, seed=seed)
    rng = np.random.default_rng(5)
    for nm in ("b_out", "b1", "b2"):
        m.layout.view(m.theta, nm)[:] = (rng.standard_normal(m.layout.shape(nm)) * 0.05).astype(np.float32)
    m.target[:] = m.theta + (rng.standard_normal(m.theta.shape) * 0.01).astype(np.float32) * (m.theta != 0)
    return m


def _bench_shard(users_total, Nn, shard):
    off, items, rew = synth_log_device(users_total, Nn, seed=12345, device=DEV, user_lo=0, user_hi=shard)
    return (off, items, rew), (off.cpu().numpy(), items.cpu().numpy(), rew.cpu().numpy())


# cfg3, cfg2, and a d=256 catalogue in direct-output mode (N >= 65 536)
@pytest.mark.parametrize("name,users_total,Nn,d,B,L,shard", [
    ("cfg3", 1_000_000, 100_000, 128, 4096, 50, 20_000),
    ("cfg2", 100_000, 10_000, 64, 4096, 50, 20_000),
    ("d256_direct", 100_000, 66_000, 256, 1024, 50, 5_000),
])
def test_published_shape_step_matches_oracle(name, users_total, Nn, d, B, L, shard):
    """Two whole steps: transitions exact, Q-values / lse / target within 1e-3 (P3), loss rtol 1e-3, every gradient
    segment normwise <= 5e-3, parameters after Adam <= 1e-3 (P4)."""
    (d_off, d_items, d_rew), (off, items, rew) = _bench_shard(users_total, Nn, shard)
    m = _oracle_model(Nn, d)
    lay = m.layout
    core = CQLCore(Nn, CQLHyper(d=d, window=L, batch=B, seed=0), device=DEV)
    core.load_flat(m.theta, m.target)
    core.set_log(d_off, d_items, d_rew)
    loss = torch.zeros(1, device=DEV)
    core.forward_backward(loss)
    v = core.views()
    g = core.grads.cpu().numpy()
    pos = O.sample_positions(0, 0, 0, B, int(off[-1]))
    users, tpos = O.positions_to_transitions(pos, off)
    with np.errstate(all="ignore"):
        out = O.loss_and_grads(lay, m.theta, m.target, off, items, rew, users, tpos, L, 0.99, 1.0, fast=True)
    # ---- integer / index work: exact
    assert np.array_equal(v["users"].cpu().numpy(), users) and np.array_equal(v["tpos"].cpu().numpy(), tpos)
    assert np.array_equal(v["act"].cpu().numpy(), items[off[users] + tpos])
    # ---- forward (P3)
    for nm, ref in (("hb_s", out.hb_s), ("hb_sn", out.hb_sn)):
        got = bf16_to_np(v[nm])
        assert np.mean(got != ref) < 2e-3, nm            # rare one-ulp flips of a bf16 rounding
    np.testing.assert_allclose(v["q_a"].cpu().numpy(), out.q_a, atol=1e-3)
    np.testing.assert_allclose(v["lse"].cpu().numpy(), out.lse, atol=1e-3)
    np.testing.assert_allclose(v["q_targ"].cpu().numpy(), out.q_targ, atol=1e-3)
    np.testing.assert_allclose(v["y"].cpu().numpy(), out.y, atol=1e-3)
    # arg-max per row (P3): a row that differs from the oracle must have picked a near-tie (<= 1e-4 below the maximum)
    bad = O.argmax_margin_violations(v["a_star"].cpu().numpy(), out.a_star, out.qn_max, out.hb_sn,
                                     lay.view(O.shadow(m.theta), "E_out"), lay.view(m.theta, "b_out"),
                                     hb_got=bf16_to_np(v["hb_sn"]))
    assert bad.size == 0, bad
    assert abs(loss.item() - out.loss) < 1e-3 * abs(out.loss)
    # ---- backward: per segment, normwise
    assert rel_err(v["dH"].cpu().numpy(), out.dH) < 5e-3
    for nm in SEGS:
        assert rel_err(lay.view(g, nm), lay.view(out.grads, nm)) < 5e-3, nm
    mask = np.ones(lay.total, bool)
    for nm in SEGS:
        mask[lay.off[nm]: lay.off[nm] + int(np.prod(lay.shape(nm)))] = False
    assert np.all(g[mask] == 0) and np.all(lay.view(g, "E_in")[Nn] == 0)      # padding / PAD row: no gradient
    # ---- Adam + Polyak
    core.apply_update()
    with np.errstate(all="ignore"):
        O.adam_ema_step(m.theta, out.grads, m.m, m.v, m.target, 1, 1e-3)
    m.step = 1
    assert rel_err(core.theta.cpu().numpy(), m.theta) < 1e-3
    assert rel_err(core.target.cpu().numpy(), m.target) < 1e-3
    assert rel_err(core.adam_m.cpu().numpy(), m.m) < 5e-3
    # the displacement of the first Adam step is lr * sign(g) wherever |g| >> eps: sign agreement on the strong rows
    strong = np.abs(out.grads) > 1e-6
    assert np.mean(np.sign(core.adam_m.cpu().numpy()[strong]) == np.sign(out.grads[strong])) > 0.999
    # ---- second step through the pipelined driver
    l2 = core.train(1).cpu().numpy()
    with np.errstate(all="ignore"):
        ref2 = O.train_steps(m, off, items, rew, 1, B, L, seed=0, fast=True)
    np.testing.assert_allclose(l2, ref2, rtol=1e-3)
    assert rel_err(core.theta.cpu().numpy(), m.theta) < 1e-3


@pytest.mark.parametrize("rows,Nn,d", [(4096, 100_000, 128), (4096, 10_000, 64), (512, 66_000, 256)])
def test_published_shape_qhead_dyadic_bit_exact(lib, rows, Nn, d):
    """P2 at the published catalogue sizes: on dyadic operands the streamed max, the argmax (ties -> smallest id) and
    Q(s, a) are bit-identical to the oracle; logsumexp within 1e-4."""
    Hb, Eb, b = qhead_inputs(rows, Nn, d, True, rows + Nn)
    Q = O.qvalues(Hb, Eb, b)
    nb = int(lib.cqlrec_qhead_ws_bytes(rows, Nn, d))
    ws = ws_bytes_tensor(nb)
    dH, dE, db = bf16_dev(Hb), bf16_dev(Eb), dev(b)
    vmax = torch.empty(rows, dtype=torch.float32, device=DEV)
    imax = torch.empty(rows, dtype=torch.int32, device=DEV)
    lse = torch.empty(rows, dtype=torch.float32, device=DEV)
    N.check(lib.cqlrec_qhead_fwd(ptr(dH), rows, ptr(dE), ptr(db), Nn, d, N.QHEAD_ARGMAX, ptr(ws), nb, ptr(vmax),
                                 ptr(imax), None, stream()))
    N.check(lib.cqlrec_qhead_fwd(ptr(dH), rows, ptr(dE), ptr(db), Nn, d, N.QHEAD_LSE, ptr(ws), nb, ptr(lse), None,
                                 None, stream()))
    act = np.random.default_rng(1).integers(0, Nn, rows).astype(np.int32)
    q_a = torch.empty(rows, dtype=torch.float32, device=DEV)
    N.check(lib.cqlrec_gather_dot(ptr(dH), ptr(dE), ptr(db), ptr(dev(act)), rows, d, ptr(q_a), stream()))
    sync()
    assert np.array_equal(vmax.cpu().numpy(), Q.max(1))
    assert np.array_equal(imax.cpu().numpy(), O.argmax_rows(Q))
    assert np.array_equal(q_a.cpu().numpy(), Q[np.arange(rows), act])
    lse_ref = O.logsumexp_rows(Q)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref, rtol=0, atol=1e-4 * max(1.0, np.abs(lse_ref).max()))


@pytest.mark.parametrize("Nn,d", [(100_000, 128), (10_000, 64)])
@pytest.mark.parametrize("dyadic", [True, False])
def test_published_shape_topk(lib, Nn, d, dyadic):
    """Top-10 with seen filtering for 256 users at the published catalogue sizes: dyadic -> ids, order and scores
    bit-identical; random -> sets equal outside a 1e-4 margin around the k-th score (P3)."""
    n_users, k = 256, 10
    idx, val, cnt, idx_ref, val_ref, Q = topk_case(lib, n_users, Nn, d, k, dyadic, Nn + 17, True)
    assert np.array_equal(cnt, np.isfinite(val_ref).sum(1))
    if dyadic:
        assert np.array_equal(idx, idx_ref)
        assert np.array_equal(val, val_ref)
        return
    excluded = 0
    for u in range(n_users):
        kth = val_ref[u, -1]
        for j in set(idx[u]) ^ set(idx_ref[u]):
            assert abs(Q[u, j] - kth) < 1e-4
            excluded += 1
        np.testing.assert_allclose(val[u], Q[u, idx[u]], atol=1e-3)
        assert np.all(np.diff(val[u]) <= 0)
    assert excluded <= 6


def test_cfg5_shard_shape_properties():
    """cfg5's per-GPU shape (N = 1 000 000 items, d = 256, B = 4096, L = 50): 64-bit row indexing of the 512 MB
    tables, the d=256 kernels at scale.  Checked against a chunked fp32 torch restatement of S4/S5 on the device
    (bf16-valued operands, fp32 accumulate -- the definition the oracle implements) and through properties of the
    gradient that hold at any size: sum_j g_b_out[j] = alpha + sum_b coef_b;  sum_j g_E_out[j] = sum_b (alpha/B +
    coef_b) hb_b;  sampled gradient rows, including the last catalogue rows;  Adam bit-exact on the head and the tail
    of the flat buffers given the device's own gradients."""
    Nn, d, B, L, shard = 1_000_000, 256, 4096, 50, 20_000
    off, items, rew = synth_log_device(1_250_000, Nn, seed=12345, device=DEV, user_lo=0, user_hi=shard)
    assert int(items.max()) > Nn - 2000          # the log really reaches the end of the tables
    core = CQLCore(Nn, CQLHyper(d=d, window=L, batch=B, seed=0), device=DEV)
    lay = core.layout
    b_out = core.segment(core.theta, "b_out")
    b_out.copy_(torch.randn(Nn, generator=torch.Generator().manual_seed(3)).to(DEV) * 0.05)
    core.target.copy_(core.theta)
    core.refresh_shadows()
    core.set_log(off, items, rew)
    theta0 = {nm: core.theta[s].clone() for nm, s in (("head", slice(0, 1 << 20)), ("tail", slice(-(1 << 20), None)))}
    loss = torch.zeros(1, device=DEV)
    core.forward_backward(loss)
    v = core.views()
    # ---- transitions: exact (integer work; the sampler is pinned bit-exact by test_sampler_bit_exact)
    off_h, items_h = off.cpu().numpy(), items.cpu().numpy()
    pos = O.sample_positions(0, 0, 0, B, int(off_h[-1]))
    users, tpos = O.positions_to_transitions(pos, off_h)
    assert np.array_equal(v["users"].cpu().numpy(), users) and np.array_equal(v["tpos"].cpu().numpy(), tpos)
    act = torch.as_tensor(items_h[off_h[users] + tpos].astype(np.int64)).to(DEV)
    assert torch.equal(v["act"].to(torch.int64), act)
    # ---- forward against the chunked restatement
    E = core.segment(core.theta_b, "E_out").float()                     # bf16-valued fp32, 1 GB
    hb_s, hb_sn = v["hb_s"].float(), v["hb_sn"].float()
    lse_ref = torch.empty(B, device=DEV)
    amax_ref = torch.empty(B, dtype=torch.int64, device=DEV)
    vmax_ref = torch.empty(B, device=DEV)
    for lo in range(0, B, 256):
        Q = hb_s[lo: lo + 256] @ E.T + b_out
        lse_ref[lo: lo + 256] = torch.logsumexp(Q.double(), 1).float()
        Qn = hb_sn[lo: lo + 256] @ E.T + b_out
        vmax_ref[lo: lo + 256], amax_ref[lo: lo + 256] = Qn.max(1)
        # every row: the chosen item's score is within 1e-4 of the row maximum (P3 margin rule, per row)
        got = v["a_star"][lo: lo + 256].to(torch.int64)
        assert torch.all(Qn.gather(1, got[:, None])[:, 0] >= vmax_ref[lo: lo + 256] - 1e-4)
    del Q, Qn
    torch.testing.assert_close(v["lse"], lse_ref, rtol=0, atol=1e-3)
    q_a_ref = (hb_s * E[act]).sum(1) + b_out[act]
    torch.testing.assert_close(v["q_a"], q_a_ref, rtol=0, atol=1e-3)
    # ---- gradient identities
    g = core.grads
    gE, gb = core.segment(g, "E_out"), core.segment(g, "b_out")
    coef = v["coef"].double()
    alpha, s = 1.0, 1.0 / B
    assert abs(gb.double().sum().item() - (alpha + coef.sum().item())) < 2e-4
    col = ((s + coef)[:, None] * hb_s.double()).sum(0)
    assert rel_err(gE.double().sum(0).cpu().numpy(), col.cpu().numpy()) < 5e-3
    # sampled rows of g_E_out / g_b_out: first and last catalogue rows, tile and slice edges, the batch's actions
    rng = np.random.default_rng(0)
    rows = np.unique(np.concatenate([np.arange(0, 130), np.arange(Nn - 130, Nn), rng.integers(0, Nn, 256),
                                     act.cpu().numpy()[:128], [65535, 65536, 65537, 499_999, 500_000]]))
    rows_t = torch.as_tensor(rows).to(DEV)
    P = torch.exp((hb_s @ E[rows_t].T + b_out[rows_t]) - v["lse"][:, None])          # [B, rows]
    Pb = P.to(torch.bfloat16).float()
    onehot = (act[:, None] == rows_t[None, :]).float() * v["coef"][:, None]
    gE_ref = s * (Pb.T @ hb_s) + onehot.T @ hb_s
    gb_ref = s * P.sum(0) + onehot.sum(0)
    assert rel_err(gE[rows_t].cpu().numpy(), gE_ref.cpu().numpy()) < 5e-3
    assert rel_err(gb[rows_t].cpu().numpy(), gb_ref.cpu().numpy()) < 1e-3
    # g_E_in touches exactly the window rows of the batch (and never the PAD row)
    gEin = core.segment(g, "E_in")
    touched = torch.nonzero(gEin.abs().sum(1) > 0)[:, 0].cpu().numpy()
    win = set()
    for u, t in zip(users[:512], tpos[:512]):
        win.update(items_h[off_h[u] + max(0, t - L): off_h[u] + t].tolist())
    assert win <= set(touched.tolist()) and Nn not in touched
    # ---- Adam on the head and the tail of the flat buffers: bit-exact given the device's gradients
    g_head, g_tail = g[: 1 << 20].cpu().numpy().copy(), g[-(1 << 20):].cpu().numpy().copy()
    core.apply_update()
    for nm, sl, gg in (("head", slice(0, 1 << 20), g_head), ("tail", slice(-(1 << 20), None), g_tail)):
        th = theta0[nm].cpu().numpy().copy()
        tg = th.copy()
        mm, vv = np.zeros_like(th), np.zeros_like(th)
        with np.errstate(all="ignore"):
            O.adam_ema_step(th, gg, mm, vv, tg, 1, 1e-3)
        assert np.array_equal(core.theta[sl].cpu().numpy(), th), nm
        assert np.array_equal(core.target[sl].cpu().numpy(), tg), nm
        assert np.array_equal(bf16_to_np(core.theta_b[sl]), O.bf16_round(th)), nm
    assert torch.count_nonzero(core.grads).item() == 0
    # ---- top-K at N = 1M, d = 256: margin rule against the restatement
    nu, k = 64, 10
    uu = torch.arange(nu, dtype=torch.int32, device=DEV)
    hb = core.encode(off, items, uu)
    idx, val, cnt = core.score_topk(hb, k)
    E = core.segment(core.theta_b, "E_out").float()
    Q = hb.float() @ E.T + core.segment(core.theta, "b_out")
    rv, ri = torch.sort(Q, dim=1, descending=True, stable=True)
    assert torch.all(cnt == k)
    for u in range(nu):
        kth = rv[u, k - 1].item()
        for j in set(idx[u].tolist()) ^ set(ri[u, :k].tolist()):
            assert abs(Q[u, j].item() - kth) < 1e-4
        torch.testing.assert_close(val[u], Q[u, idx[u].long()], rtol=0, atol=1e-3)


@pytest.mark.parametrize("d,chunk", [(128, 1536), (128, 700), (64, 4000)])
def test_encode_topk_pipelined_equals_two_calls(d, chunk):
    """core.encode_topk (seen bitmap of a chunk built on a side stream while the chunk is encoded, several chunks
    through one workspace) returns exactly what encode() followed by score_topk() returns."""
    U, Nn, L, k = 4000, 5000, 20, 10
    off, items, rew = synth_log_device(U, Nn, seed=777, device=DEV)
    core = CQLCore(Nn, CQLHyper(d=d, window=L, batch=256, seed=0), device=DEV)
    core.set_log(off, items, rew)
    core.train(3)
    users = torch.randperm(U, generator=torch.Generator().manual_seed(5))[:3500].to(DEV).to(torch.int32)
    rows = torch.repeat_interleave(torch.arange(U, device=DEV), off[1:] - off[:-1])
    seen = items[torch.argsort(rows * Nn + items.to(torch.int64))].contiguous()
    hb = core.encode(off, items, users)
    ref = core.score_topk(hb, k, seen=(off, seen), seen_rows=users)
    for _ in range(2):            # twice: the second pass reuses streams and events of the first
        got = core.encode_topk(off, items, users, k, seen=(off, seen), chunk=chunk)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    got = core.encode_topk(off, items, users, k, chunk=chunk)          # no filter
    ref = core.score_topk(hb, k)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)

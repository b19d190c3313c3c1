"""End-to-end GPU parity of the training step and of CQLCore against the oracle (P4)."""
import numpy as np
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd.core import CQLCore, CQLHyper

from helpers import DEV, bf16_to_np, rel_err, small_log

pytestmark = pytest.mark.gpu


def _make(U, Nn, d, B, L, dyadic=False, seed=3, alpha=1.0):
    off, items, rew = small_log(U=U, N=Nn, seed=seed, mean_len=14, max_len=45)
    m = O.OracleModel.create(Nn, d, seed=7, dyadic=dyadic)
    rng = np.random.default_rng(5)
    if not dyadic:
        for nm in ("b_out", "b1", "b2"):
            m.layout.view(m.theta, nm)[:] = (rng.standard_normal(m.layout.shape(nm)) * 0.05).astype(np.float32)
        m.target[:] = m.theta + (rng.standard_normal(m.theta.shape) * 0.01).astype(np.float32) * (m.theta != 0)
    hyper = CQLHyper(d=d, window=L, batch=B, seed=11, alpha=alpha)
    core = CQLCore(Nn, hyper, device=DEV)
    core.load_flat(m.theta, m.target)
    core.set_log(off, items, rew)
    return m, core, (off, items, rew)


@pytest.mark.parametrize("U,Nn,d,B,L", [(64, 257, 64, 64, 5), (300, 1000, 128, 256, 8), (200, 4099, 128, 96, 50),
                                        (100, 513, 256, 128, 10)])
def test_step_intermediates_and_grads(U, Nn, d, B, L):
    m, core, (off, items, rew) = _make(U, Nn, d, B, L)
    lay = m.layout
    assert int(core.layout.total) == lay.total
    for nm in ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2"):
        assert core.layout.offset(nm) == lay.off[nm]
    loss = torch.zeros(1, device=DEV)
    core.forward_backward(loss)
    v = core.views()
    pos = O.sample_positions(11, 0, 0, B, int(off[-1]))
    users, tpos = O.positions_to_transitions(pos, off)
    out = O.loss_and_grads(lay, m.theta, m.target, off, items, rew, users, tpos, L, 0.99, 1.0)
    assert np.array_equal(v["users"].cpu().numpy(), users) and np.array_equal(v["tpos"].cpu().numpy(), tpos)
    np.testing.assert_allclose(v["h0_s"].cpu().numpy(), out.h0_s, rtol=1e-5, atol=1e-6)
    # bf16 state vectors: identical up to one bf16 ulp on rare rounding ties of an fp32 sum
    for name, ref in (("hb_s", out.hb_s), ("hb_sn", out.hb_sn)):
        got = bf16_to_np(v[name])
        assert np.mean(got != ref) < 2e-3
        np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(v["q_a"].cpu().numpy(), out.q_a, atol=1e-3)          # P3 |dQ| <= 1e-3
    np.testing.assert_allclose(v["lse"].cpu().numpy(), out.lse, atol=1e-3)
    np.testing.assert_allclose(v["q_targ"].cpu().numpy(), out.q_targ, atol=1e-3)
    np.testing.assert_allclose(v["y"].cpu().numpy(), out.y, atol=1e-3)
    bad = O.argmax_margin_violations(v["a_star"].cpu().numpy(), out.a_star, out.qn_max, out.hb_sn,
                                     lay.view(O.shadow(m.theta), "E_out"), lay.view(m.theta, "b_out"),
                                     hb_got=bf16_to_np(v["hb_sn"]))
    assert bad.size == 0, bad                                                    # per row, P3 margin rule
    assert abs(loss.item() - out.loss) < 1e-3 * abs(out.loss)
    assert rel_err(v["dH"].cpu().numpy(), out.dH) < 5e-3
    assert rel_err(v["dh0"].cpu().numpy(), out.dh0) < 5e-3
    g = core.grads.cpu().numpy()
    for nm in ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2"):
        assert rel_err(lay.view(g, nm), lay.view(out.grads, nm)) < 5e-3, nm
    # padding and the PAD row never receive gradient
    mask = np.ones(lay.total, bool)
    for nm in ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2"):
        mask[lay.off[nm]: lay.off[nm] + int(np.prod(lay.shape(nm)))] = False
    assert np.all(g[mask] == 0) and np.all(lay.view(g, "E_in")[Nn] == 0)


def test_dyadic_forward_bit_exact():
    """P2: on dyadic parameters the whole forward (h0, h, Q-values, argmax) is exact."""
    m, core, (off, items, rew) = _make(80, 513, 64, 64, 4, dyadic=True)
    core.forward_backward(None)
    v = core.views()
    pos = O.sample_positions(11, 0, 0, 64, int(off[-1]))
    users, tpos = O.positions_to_transitions(pos, off)
    out = O.loss_and_grads(m.layout, m.theta, m.target, off, items, rew, users, tpos, 4, 0.99, 1.0)
    assert np.array_equal(v["h0_s"].cpu().numpy(), out.h0_s)
    assert np.array_equal(bf16_to_np(v["hb_s"]), out.hb_s)
    assert np.array_equal(bf16_to_np(v["hb_sn"]), out.hb_sn)
    assert np.array_equal(v["q_a"].cpu().numpy(), out.q_a)
    assert np.array_equal(v["a_star"].cpu().numpy(), out.a_star)
    assert np.array_equal(v["q_targ"].cpu().numpy(), out.q_targ)


@pytest.mark.parametrize("U,Nn,d,B,L,steps", [(300, 1000, 128, 256, 8, 6), (64, 257, 64, 64, 5, 10)])
def test_training_trajectory(U, Nn, d, B, L, steps):
    """P4: loss after n Adam steps within 1e-3 relative; parameters normwise within 1e-3."""
    m, core, (off, items, rew) = _make(U, Nn, d, B, L)
    losses = core.train(steps).cpu().numpy()
    ref = O.train_steps(m, off, items, rew, steps, B, L, seed=11)
    np.testing.assert_allclose(losses, ref, rtol=1e-3)
    assert core.step == steps
    th = core.theta.cpu().numpy()
    assert rel_err(th, m.theta) < 1e-3
    assert rel_err(core.target.cpu().numpy(), m.target) < 1e-3
    # update direction itself (theta moved by ~steps*lr): compare the displacement normwise
    m0 = O.OracleModel.create(Nn, d, seed=7)
    assert np.count_nonzero(core.grads.cpu().numpy()) == 0          # re-zeroed by the fused update
    assert np.array_equal(bf16_to_np(core.theta_b), O.bf16_round(th))


def test_pipelined_steps_match_step_by_step():
    """cqlrec_train_steps (Adam halves under the backward, next prologue under the item-side Adam, double-buffered
    step vectors) computes the same steps as fwd_bwd + update one at a time in strict program order."""
    from replay_cql_amd import _native as N
    U, Nn, d, B, L, steps = 300, 1000, 128, 256, 8, 7
    _, a, _ = _make(U, Nn, d, B, L)
    _, b, _ = _make(U, Nn, d, B, L)
    la = torch.zeros(steps, device=DEV)
    a.train_steps(3, la)
    a.train_steps(steps - 3, la[3:])
    lb = torch.zeros(steps, device=DEV)
    N.check(N.load().cqlrec_set_concurrency(0))
    try:
        for i in range(steps):
            b.forward_backward(lb[i:i + 1])
            b.apply_update()
    finally:
        N.check(N.load().cqlrec_set_concurrency(1))
    torch.cuda.synchronize()
    assert a.step == b.step == steps
    assert torch.equal(la, lb)
    # the step is deterministic (no float atomics): bit for bit
    assert torch.equal(a.theta, b.theta) and torch.equal(a.target, b.target)
    assert np.count_nonzero(a.grads.cpu().numpy()) == 0
    # the two-pass item-side Adam leaves the same shadows as the one-pass kernel
    assert np.array_equal(bf16_to_np(a.theta_b), O.bf16_round(a.theta.cpu().numpy()))
    assert np.array_equal(bf16_to_np(a.target_b), O.bf16_round(a.target.cpu().numpy()))
    # views of a given step stay addressable by parity
    v = a.views(steps - 1)
    pos = O.sample_positions(11, steps - 1, 0, B, int(a._csr[0][-1]))
    assert np.array_equal(v["tpos"].cpu().numpy(), O.positions_to_transitions(pos, a._csr[0].cpu().numpy())[1])


@pytest.mark.parametrize("U,Nn,d,B,L", [(300, 1000, 128, 256, 8), (200, 66000, 64, 128, 6),
                                          (200, 66000, 128, 128, 6), (200, 70001, 128, 192, 6)])
def test_steps_are_bit_reproducible(U, Nn, d, B, L):
    """No float atomics anywhere in the step (sorted segmented sums for both scatters, ordered slab reductions), so the
    same steps give the same bits: run to run, and pipelined (side streams, cross-step overlap) vs strict program
    order on one stream -- which also makes this the sharpest race detector of the suite."""
    from replay_cql_amd import _native as N
    steps = 6
    cores = [_make(U, Nn, d, B, L)[1] for _ in range(4)]
    cores[0].train_steps(steps)
    cores[1].train_steps(2)
    cores[1].train_steps(steps - 2)
    cores[2].train_steps(steps, phased=True)             # the data-parallel phase path on one rank
    N.check(N.load().cqlrec_set_concurrency(0))
    try:
        for _ in range(steps):
            cores[3].forward_backward(None)
            cores[3].apply_update()
    finally:
        N.check(N.load().cqlrec_set_concurrency(1))
    torch.cuda.synchronize()
    ref = cores[3]
    for i, c in enumerate(cores[:3]):
        for name in ("theta", "target", "adam_m", "adam_v"):
            assert torch.equal(getattr(c, name), getattr(ref, name)), (i, name)
        assert torch.equal(c.theta_b, ref.theta_b) and torch.equal(c.target_b, ref.target_b), i


def test_large_catalogue_direct_output_steps():
    """Catalogues of >= 65 536 items: the item-side backward kernel runs in its direct-output mode (one block per 128
    items writes its gradient rows itself, on top of the scattered one-hot part).  Pipelined steps == step by step,
    and the oracle's trajectory."""
    from replay_cql_amd import _native as N
    U, Nn, d, B, L, steps = 200, 66000, 64, 128, 6, 4
    m, a, (off, items, rew) = _make(U, Nn, d, B, L)
    _, b, _ = _make(U, Nn, d, B, L)
    la = torch.zeros(steps, device=DEV)
    a.train_steps(steps, la)                       # pipelined
    lb = torch.zeros(steps, device=DEV)
    N.check(N.load().cqlrec_set_concurrency(0))
    try:
        for i in range(steps):                     # strict program order
            b.forward_backward(lb[i:i + 1])
            b.apply_update()
    finally:
        N.check(N.load().cqlrec_set_concurrency(1))
    torch.cuda.synchronize()
    assert torch.equal(la, lb)
    for name in ("theta", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert np.count_nonzero(a.grads.cpu().numpy()) == 0
    assert np.array_equal(bf16_to_np(a.theta_b), O.bf16_round(a.theta.cpu().numpy()))
    assert np.array_equal(bf16_to_np(a.target_b), O.bf16_round(a.target.cpu().numpy()))
    ref = O.train_steps(m, off, items, rew, steps, B, L, seed=11)
    np.testing.assert_allclose(la.cpu().numpy(), ref, rtol=1e-3)
    assert rel_err(a.theta.cpu().numpy(), m.theta) < 1e-3


def test_core_predict_matches_oracle():
    U, Nn, d, L, k = 150, 2000, 128, 10, 10
    m, core, (off, items, rew) = _make(U, Nn, d, 128, L)
    users = np.arange(U, dtype=np.int32)
    d_off, d_items = core._csr[0], core._csr[1]
    hb = core.encode(d_off, d_items, torch.as_tensor(users).to(DEV))
    # seen CSR = items sorted inside each user's row (same offsets)
    seen = items.copy()
    for u in range(U):
        seen[off[u]: off[u + 1]] = np.sort(seen[off[u]: off[u + 1]])
    d_seen = torch.as_tensor(seen).to(DEV)
    idx, val, cnt = core.score_topk(hb, k, seen=(d_off, d_seen), chunk=64)
    ridx, rval, rcnt, rhb = O.predict_topk(m.layout, m.theta, off, items, users, k, L, filter_seen=True)
    got_hb = bf16_to_np(hb)
    assert np.mean(got_hb != rhb) < 2e-3
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    Q = O.qvalues(rhb, O.bf16_round(m.layout.view(m.theta, "E_out")), m.layout.view(m.theta, "b_out"))
    bad = 0
    for u in range(U):
        assert not set(idx[u]) & set(items[off[u]: off[u + 1]])      # nothing seen is recommended
        for j in set(idx[u]) ^ set(ridx[u]):
            assert abs(Q[u, j] - rval[u, -1]) < 2e-3
            bad += 1
        np.testing.assert_allclose(val[u], Q[u, idx[u]], atol=2e-3)
    assert bad <= 6
    # pairs
    pi = np.random.default_rng(0).integers(0, Nn, U).astype(np.int32)
    ps = core.pair_scores(hb, torch.as_tensor(pi).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(ps, Q[np.arange(U), pi], atol=2e-3)


def test_full_ranking_beyond_the_fused_k_limit():
    """k > 2048 (full-ranking requests, e.g. metrics over the whole catalogue): per-part complete rankings merged per
    user.  Dyadic parameters -> scores are exact, so ids AND order must equal the oracle's (score desc, id asc)."""
    U, Nn, d, L, k = 12, 5000, 64, 6, 3000
    m, core, (off, items, rew) = _make(U, Nn, d, 64, L, dyadic=True)
    users = np.arange(U, dtype=np.int32)
    d_off, d_items = core._csr[0], core._csr[1]
    hb = core.encode(d_off, d_items, torch.as_tensor(users).to(DEV))
    seen = items.copy()
    for u in range(U):
        seen[off[u]: off[u + 1]] = np.sort(seen[off[u]: off[u + 1]])
    idx, val, cnt = core.score_topk(hb, k, seen=(d_off, torch.as_tensor(seen).to(DEV)))
    ridx, rval, rcnt, _ = O.predict_topk(m.layout, m.theta, off, items, users, k, L, filter_seen=True)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(val.cpu().numpy(), rval)


def test_phased_step_equals_fused_step():
    """the data-parallel phase split (forward / backward_items / backward_rest / update_range x3, side-stream sort)
    computes the same step as the fused driver."""
    m, core_a, (off, items, rew) = _make(300, 1000, 128, 256, 8)
    _, core_b, _ = _make(300, 1000, 128, 256, 8)
    la = core_a.train(5, phased=False).cpu().numpy()
    lb = core_b.train(5, phased=True).cpu().numpy()
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    assert core_a.step == core_b.step == 5
    assert torch.equal(core_a.theta, core_b.theta) and torch.equal(core_a.target, core_b.target)
    assert torch.count_nonzero(core_b.grads).item() == 0
    assert torch.equal(core_b.theta_b.float(), core_b.theta.to(torch.bfloat16).float())


@pytest.mark.parametrize("phased", [False, True])
def test_early_item_side_launch_changes_no_bit(phased):
    """The long dE_out kernel launched behind the fused forward (default, in the pipelined single-rank driver and -- through
    cqlrec_train_step_forward_early_items -- in the data-parallel step loop) against the old order, behind the loss
    (CQL_EARLY_DE=0): losses, parameters, Adam moments, target net and shadows bit-identical after 6 steps + one single
    step.  The knob is read once per process: two child processes (tools/train_digest.py)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    outs = []
    for knob in ("1", "0"):
        env = dict(os.environ, CQL_EARLY_DE=knob)
        cmd = [sys.executable, str(root / "tools" / "train_digest.py"), "--d", "128", "--items", "20000", "--steps", "6"]
        if phased:
            cmd.append("--phased")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["losses"] == outs[1]["losses"]
    assert outs[0]["state_sha256"] == outs[1]["state_sha256"]

"""CPU tests of the oracle: committed golden vectors, an independent autograd cross-check of its analytic gradient,
known answers for the integer pieces.  (The reference holds no vector for this path: parity unpinned)"""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import cql_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
CASES = ["tiny_dyadic", "small_random", "small_dyadic", "medium_random", "survey_medium_random", "survey_medium_dyadic"]


def _load(name):
    z = np.load(GOLD / f"{name}.npz")
    U, Nn, d, L, B, steps, dyadic, ls = [int(x) for x in z["case"]]
    off, items, rew = O.build_csr(z["log_user"], z["log_item"], z["log_ts"], z["log_rel"], U)
    return z, (U, Nn, d, L, B, steps, bool(dyadic)), (off, items, rew)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    z, (U, Nn, d, L, B, steps, dyadic), (off, items, rew) = _load(name)
    m = O.OracleModel.create(Nn, d, seed=7, dyadic=dyadic)
    assert np.array_equal(m.theta[:: max(1, m.layout.total // 257)][:257], z["theta0_probe"])
    # top-K of the initial parameters: on dyadic parameters ids, order and scores are exact (P2)
    nu0 = min(U, 256)
    idx0, val0, cnt0, _ = O.predict_topk(m.layout, m.theta, off, items, np.arange(nu0), min(10, Nn), L, filter_seen=True)
    assert np.array_equal(cnt0, z["topk0_cnt"])
    if dyadic:
        assert np.array_equal(idx0, z["topk0_idx"]) and np.array_equal(val0, z["topk0_val"])
    else:
        np.testing.assert_allclose(val0, z["topk0_val"], rtol=1e-5, atol=1e-6)
    pos = O.sample_positions(11, 0, 0, B, int(off[-1]))
    users, tpos = O.positions_to_transitions(pos, off)
    assert np.array_equal(users, z["users"]) and np.array_equal(tpos, z["tpos"])
    out = O.loss_and_grads(m.layout, m.theta, m.target, off, items, rew, users, tpos, L, 0.99, 1.0)
    tol = dict(rtol=0, atol=0) if dyadic else dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.q_a, z["q_a"], **tol)
    np.testing.assert_allclose(out.q_targ, z["q_targ"], **tol)
    assert np.array_equal(out.a_star, z["a_star"])
    np.testing.assert_allclose(out.lse, z["lse"], rtol=1e-5, atol=1e-6)
    assert abs(out.loss - float(z["loss0"])) < 1e-6 * abs(float(z["loss0"]))
    norms = [np.linalg.norm(m.layout.view(out.grads, n)) for n in ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2")]
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=1e-4)
    losses = O.train_steps(m, off, items, rew, steps, B, L, seed=11)
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    np.testing.assert_allclose(m.theta[:: max(1, m.layout.total // 257)][:257], z["theta_probe"], rtol=1e-4, atol=1e-6)
    idx, val, cnt, _ = O.predict_topk(m.layout, m.theta, off, items, np.arange(U), min(10, Nn), L, filter_seen=True)
    assert np.array_equal(cnt, z["topk_cnt"])
    np.testing.assert_allclose(val, z["topk_val"], rtol=1e-4, atol=1e-5)


def test_fast_gather_equals_reference_gather():
    _, (U, Nn, d, L, B, steps, dyadic), (off, items, rew) = _load("small_dyadic")
    m = O.OracleModel.create(Nn, d, seed=7, dyadic=True)
    E = O.bf16_round(m.layout.view(m.theta, "E_in"))
    rng = np.random.default_rng(0)
    users = rng.integers(0, U, 200)
    ends = rng.integers(0, 10**6, 200) % (off[users + 1] - off[users] + 1)
    a, la = O.gather_pool(E, off, items, users, ends, L)
    b, lb = O.gather_pool_fast(E, off, items, users, ends, L)
    assert np.array_equal(a, b) and np.array_equal(la, lb)


def test_oracle_gradient_matches_autograd():
    """Independent pin of the analytic backward: torch autograd on the same loss with straight-through bf16 rounding.
    The only intended difference is the bf16 rounding of softmax probabilities inside the two gradient GEMMs."""
    def rb(x):
        return x + (x.detach().to(torch.bfloat16).to(torch.float32) - x.detach())
    U, Nn, d, L, B = 64, 257, 16, 5, 48
    u, i, t, r = O.synth_log(U, Nn, seed=1, mean_len=12, max_len=40)
    off, items, rew = O.build_csr(u, i, t, r, U)
    m = O.OracleModel.create(Nn, d)
    rng = np.random.default_rng(0)
    lay = m.layout
    for nm in ("b_out", "b1", "b2"):
        lay.view(m.theta, nm)[:] = rng.standard_normal(lay.shape(nm)).astype(np.float32) * 0.1
    m.target[:] = m.theta + rng.standard_normal(m.theta.shape).astype(np.float32) * 0.01
    pos = O.sample_positions(3, 0, 0, B, int(off[-1]))
    us, tp = O.positions_to_transitions(pos, off)
    gamma, alpha = 0.99, 0.7
    out = O.loss_and_grads(lay, m.theta, m.target, off, items, rew, us, tp, L, gamma, alpha)
    th = torch.tensor(m.theta, requires_grad=True)
    tg = torch.tensor(m.target)

    def V(flat, nm):
        shp = lay.shape(nm)
        return flat[lay.off[nm]: lay.off[nm] + int(np.prod(shp))].reshape(shp)

    def enc(flat, users, ends):
        Ein, W1, W2 = rb(V(flat, "E_in")), rb(V(flat, "W1")), rb(V(flat, "W2"))
        hs = []
        for uu, e in zip(users, ends):
            ln = min(int(e), L)
            if ln == 0:
                hs.append(torch.zeros(d))
                continue
            base = int(off[uu]) + int(e)
            hs.append(Ein[torch.tensor(items[base - ln: base].astype(np.int64))].sum(0) / ln)
        h0 = torch.stack(hs)
        z = torch.relu(rb(h0) @ W1.T + V(flat, "b1"))
        return rb(rb(z) @ W2.T + V(flat, "b2"))
    us64, tp64 = us.astype(np.int64), tp.astype(np.int64)
    hs, hn = enc(th, us64, tp64), enc(th, us64, tp64 + 1)
    with torch.no_grad():
        ht = enc(tg, us64, tp64 + 1)
    Eo, bo = rb(V(th, "E_out")), V(th, "b_out")
    Qs = hs @ Eo.T + bo
    act = torch.tensor(items[off[us64] + tp64].astype(np.int64))
    qa, lse = Qs[torch.arange(B), act], torch.logsumexp(Qs, 1)
    with torch.no_grad():
        astar = (hn @ Eo.T + bo).argmax(1)
        Eot = V(tg, "E_out").to(torch.bfloat16).float()
        qt = (ht * Eot[astar]).sum(1) + V(tg, "b_out")[astar]
        done = torch.tensor((tp64 == off[us64 + 1] - off[us64] - 1).astype(np.float32))
        y = torch.tensor(rew[off[us64] + tp64]) + gamma * (1 - done) * qt
    loss = (0.5 * (qa - y) ** 2 + alpha * (lse - qa)).mean()
    loss.backward()
    g = th.grad.numpy()
    assert abs(out.loss - loss.item()) < 1e-5 * abs(loss.item())
    assert np.array_equal(out.a_star, astar.numpy())
    for nm in ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2"):
        a, b = lay.view(out.grads, nm), lay.view(g, nm)
        assert np.linalg.norm(a - b) <= 3e-4 * np.linalg.norm(b), nm


def test_bf16_round_known_answers():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -2.5, 3.1415927, 0.0, 65504.0], dtype=np.float32)
    # 1 + 2^-8 is a tie -> even (1.0); 1 + 3*2^-9 rounds up to 1 + 2^-7; 1 + 3*2^-8 tie -> even (1 + 2^-6)
    exp = np.array([1.0, 1.0, 1.0078125, 1.015625, -2.5, 3.140625, 0.0, 65536.0], dtype=np.float32)
    assert np.array_equal(O.bf16_round(x), exp)
    assert np.array_equal(O.bf16_round(x), torch.tensor(x).to(torch.bfloat16).float().numpy())
    r = np.random.default_rng(0).standard_normal(100000).astype(np.float32)
    assert np.array_equal(O.bf16_round(r), torch.tensor(r).to(torch.bfloat16).float().numpy())
    assert np.array_equal(O.bf16_from_bits(O.bf16_bits(r)), O.bf16_round(r))


def test_sampler_properties():
    nnz = 12345
    p = O.sample_positions(5, 3, 0, 4096, nnz)
    assert p.min() >= 0 and p.max() < nnz
    assert np.array_equal(p[100:200], O.sample_positions(5, 3, 100, 100, nnz))     # slots are independent of batch split
    assert not np.array_equal(p, O.sample_positions(5, 4, 0, 4096, nnz))
    assert abs(p.mean() / nnz - 0.5) < 0.02
    assert O._mix64(0) == 0xE220A8397B1DCDAF                                       # splitmix64 known answer


def test_topk_tie_rule_and_seen():
    s = np.array([[1.0, 3.0, 3.0, 2.0, 3.0], [0.0, 0.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    idx, val = O.topk_rows(s, 3)
    assert idx.tolist() == [[1, 2, 4], [0, 1, 2]]
    lay = O.Layout.make(5, 64)
    th = np.zeros(lay.total, np.float32)
    lay.view(th, "b_out")[:] = [5, 4, 3, 2, 1]
    off = np.array([0, 2, 2], dtype=np.int64)
    items = np.array([0, 1], dtype=np.int32)
    idx, val, cnt, _ = O.predict_topk(lay, th, off, items, np.array([0, 1]), 4, 3, filter_seen=True)
    assert idx.tolist() == [[2, 3, 4, -1], [0, 1, 2, 3]] and cnt.tolist() == [3, 4]


def test_adam_matches_torch_adam():
    rng = np.random.default_rng(0)
    n = 1000
    theta = rng.standard_normal(n).astype(np.float32)
    p = torch.tensor(theta.copy(), requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-3)
    m, v, target = np.zeros(n, np.float32), np.zeros(n, np.float32), theta.copy()
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        p.grad = torch.tensor(g)
        opt.step()
        O.adam_ema_step(theta, g, m, v, target, t, 1e-3)
        np.testing.assert_allclose(theta, p.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_layout_is_aligned_and_counts_params():
    lay = O.Layout.make(100_000, 128)
    assert all(o % O.SEG_ALIGN == 0 for o in lay.off.values()) and lay.total % O.SEG_ALIGN == 0
    assert lay.n_params == (2 * 100_000 + 1) * 128 + 100_000 + 2 * 128 * 128 + 2 * 128

"""Generates tests/golden/*.npz from the oracle (PARITY UNPINNED: the reference has no CQL path and no vectors for
it -- SURVEY.md 8(c) -- so these fixtures pin the build's own CPU restatement against regressions, and are what the
GPU box checks the HIP path against at the fixture sizes).   python tests/golden/make_golden.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import cql_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = {
    # name: (U, N, d, L, B, steps, dyadic, log_seed)
    "tiny_dyadic": (4, 4, 64, 2, 8, 2, True, 1),
    "small_random": (64, 257, 64, 5, 32, 4, False, 2),
    "small_dyadic": (64, 257, 64, 5, 32, 1, True, 2),
    "medium_random": (300, 1000, 128, 8, 64, 3, False, 3),
    # the SURVEY 8(c) "medium" fixture: U=2048, N=10007, d=64, L=50
    "survey_medium_random": (2048, 10007, 64, 50, 256, 3, False, 4),
    "survey_medium_dyadic": (2048, 10007, 64, 50, 256, 1, True, 4),
}


def make(name):
    U, Nn, d, L, B, steps, dyadic, ls = CASES[name]
    big = U >= 2048      # long histories, so that the L=50 window is really full for most states
    u, i, t, r = O.synth_log(U, Nn, seed=ls, mean_len=6 if U < 10 else (40 if big else 14), min_len=3,
                             max_len=120 if big else 45)
    off, items, rew = O.build_csr(u, i, t, r, U)
    m = O.OracleModel.create(Nn, d, seed=7, dyadic=dyadic)
    stride = max(1, m.layout.total // 257)
    theta0_probe = m.theta[::stride][:257].copy()
    k = min(10, Nn)
    # top-K of the INITIAL parameters (dyadic cases: ids, order and scores are exact) for the first 256 users
    nu0 = min(U, 256)
    idx0, val0, cnt0, _ = O.predict_topk(m.layout, m.theta, off, items, np.arange(nu0), k, L, filter_seen=True)
    pos = O.sample_positions(11, 0, 0, B, int(off[-1]))
    users, tpos = O.positions_to_transitions(pos, off)
    out = O.loss_and_grads(m.layout, m.theta, m.target, off, items, rew, users, tpos, L, 0.99, 1.0)
    losses = O.train_steps(m, off, items, rew, steps, B, L, seed=11)
    idx, val, cnt, _ = O.predict_topk(m.layout, m.theta, off, items, np.arange(U), k, L, filter_seen=True)
    lay = m.layout
    # Trained parameters, as what the scoring pass depends on (so that the top-K after training can be checked to the
    # 1e-3 / 1e-4 rules on the FIXTURE's parameters instead of on parameters the checked path trained itself, which
    # agree to 1e-3 only): the bf16 shadow as a signed difference of bit patterns against the initial shadow (a few
    # ulps after `steps` Adam steps: compresses to a fraction of a byte per parameter) + the fp32 biases in full.
    th0 = O.init_params(lay, 7, dyadic)
    b_delta = (O.bf16_bits(m.theta).astype(np.int32) - O.bf16_bits(th0).astype(np.int32)).astype(np.int16)
    assert np.array_equal((O.bf16_bits(th0).astype(np.int32) + b_delta).astype(np.uint16), O.bf16_bits(m.theta))
    np.savez_compressed(
        OUT / f"{name}.npz",
        case=np.array([U, Nn, d, L, B, steps, int(dyadic), ls]),
        log_user=u.astype(np.int32), log_item=i.astype(np.int32), log_ts=t.astype(np.int32), log_rel=r.astype(np.float32),
        theta0_probe=theta0_probe, topk0_idx=idx0, topk0_val=val0, topk0_cnt=cnt0,
        users=users, tpos=tpos, q_a=out.q_a, lse=out.lse, a_star=out.a_star, q_targ=out.q_targ, y=out.y,
        qn_max=out.qn_max, hb_sn_bits=O.bf16_bits(out.hb_sn),
        trained_shadow_delta=b_delta, trained_b_out=lay.view(m.theta, "b_out").copy(),
        trained_b1=lay.view(m.theta, "b1").copy(), trained_b2=lay.view(m.theta, "b2").copy(),
        loss0=np.float64(out.loss), grad_norms=np.array([np.linalg.norm(lay.view(out.grads, n)) for n in
                                                         ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2")]),
        losses=np.array(losses), theta_sum=np.float64(m.theta.astype(np.float64).sum()),
        theta_probe=m.theta[::stride][:257].copy(),
        topk_idx=idx, topk_val=val, topk_cnt=cnt)


if __name__ == "__main__":
    for n in CASES:
        make(n)
        print("wrote", n)

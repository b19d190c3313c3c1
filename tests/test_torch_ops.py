"""torch.ops.cqlrec.* (the hot path "exposed as PyTorch-ROCm custom ops"): the registration shim over the
C ABI.  CPU: the library loads and registers every op with the documented schema.  GPU: every op reproduces the ctypes
path bit for bit (same kernels, same stream) -- so every parity statement made through the C ABI holds for the ops."""
import numpy as np
import pytest
import torch

from oracle import cql_oracle as O
from replay_cql_amd import _native as N
from replay_cql_amd import torch_ops

DEV = "cuda:0"


def test_ops_are_registered_with_their_schemas():
    ops = torch_ops.load()
    for name in torch_ops.OPS:
        assert hasattr(ops, name), name
    assert str(ops.score_topk.default._schema).startswith(
        "cqlrec::score_topk(Tensor H_b, Tensor E_b, Tensor b, int k, Tensor? item_ids=None")
    assert "Tensor(a!) theta" in str(ops.fused_adam_ema.default._schema)
    # no CPU kernels: the product path fails loudly off the GPU
    z = torch.zeros(2, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        ops.qhead_gather_dot(z, z, torch.zeros(2), torch.zeros(2, dtype=torch.int32))


def _bf(x):
    return torch.as_tensor(O.bf16_bits(np.asarray(x, np.float32)).astype(np.int16)).to(DEV).view(torch.bfloat16).contiguous()


@pytest.mark.gpu
def test_ops_match_the_ctypes_path_bit_for_bit():
    from replay_cql_amd.core import CQLCore, CQLHyper
    ops, lib = torch_ops.load(), N.load()
    rng = np.random.default_rng(0)
    U, Nn, d, L, B, k = 300, 5003, 128, 12, 256, 10
    u, i, t, r = O.synth_log(U, Nn, seed=3, mean_len=20, max_len=60)
    off, items, rew = O.build_csr(u, i, t, r, U)
    core = CQLCore(Nn, CQLHyper(d=d, window=L, batch=B, seed=1), device=DEV)
    core.set_log(off, items, rew)
    d_off, d_items, _ = core._csr
    lay = core.layout
    E_in_b, E_out_b = core.segment(core.theta_b, "E_in"), core.segment(core.theta_b, "E_out")
    b_out = core.segment(core.theta, "b_out")
    b_out.copy_(torch.randn(Nn, device=DEV) * 0.1)
    users = torch.arange(U, dtype=torch.int32, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    # ---- gather_pool_fwd
    h0, h0b = ops.gather_pool_fwd(E_in_b, d_off, d_items, users, None, 0, L)
    r0, r0b = torch.empty_like(h0), torch.empty_like(h0b)
    N.check(lib.cqlrec_gather_pool_fwd(E_in_b.data_ptr(), d_off.data_ptr(), d_items.data_ptr(), users.data_ptr(), None, 0, U,
                                       L, d, r0.data_ptr(), r0b.data_ptr(), None, s))
    assert torch.equal(h0, r0) and torch.equal(h0b.view(torch.int16), r0b.view(torch.int16))
    # ---- gather_pool_bwd (deterministic sorted form)
    dh0 = torch.randn(U, d, device=DEV)
    g = ops.gather_pool_bwd(dh0, d_off, d_items, users, None, 0, L, Nn)
    ws_b = int(lib.cqlrec_gather_pool_bwd_ws_bytes(U, L, d))
    ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    g_ref = torch.zeros(Nn + 1, d, device=DEV)
    N.check(lib.cqlrec_gather_pool_bwd_sorted(dh0.data_ptr(), d_off.data_ptr(), d_items.data_ptr(), users.data_ptr(), None, 0, U,
                                              L, d, Nn, ws.data_ptr(), ws_b, g_ref.data_ptr(), s))
    assert torch.equal(g, g_ref) and g.abs().sum() > 0
    # ---- Q-head forward / backward / gather-dot on the encoder output
    hb = core.encode(d_off, d_items, users)
    lse, nlse2 = ops.qhead_lse_fwd(hb, E_out_b, b_out)
    vmax, imax = ops.qhead_argmax_fwd(hb, E_out_b, b_out)
    ws_b = int(lib.cqlrec_qhead_ws_bytes(U, Nn, d))
    ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    r_lse, r_nl, r_v = (torch.empty(U, device=DEV) for _ in range(3))
    r_i = torch.empty(U, dtype=torch.int32, device=DEV)
    N.check(lib.cqlrec_qhead_fwd(hb.data_ptr(), U, E_out_b.data_ptr(), b_out.data_ptr(), Nn, d, N.QHEAD_LSE, ws.data_ptr(), ws_b,
                                 r_lse.data_ptr(), None, r_nl.data_ptr(), s))
    N.check(lib.cqlrec_qhead_fwd(hb.data_ptr(), U, E_out_b.data_ptr(), b_out.data_ptr(), Nn, d, N.QHEAD_ARGMAX, ws.data_ptr(),
                                 ws_b, r_v.data_ptr(), r_i.data_ptr(), None, s))
    assert torch.equal(lse, r_lse) and torch.equal(nlse2, r_nl) and torch.equal(vmax, r_v) and torch.equal(imax, r_i)
    act = torch.as_tensor(rng.integers(0, Nn, U).astype(np.int32)).to(DEV)
    coef = torch.randn(U, device=DEV) * 0.01
    dH, gE, gb = ops.qhead_lse_bwd(hb, nlse2, coef, act, E_out_b, b_out, 1.0 / U)
    ws_b = int(lib.cqlrec_qhead_bwd_ws_bytes(U, Nn, d))
    ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    r_dH, r_gE, r_gb = torch.empty_like(dH), torch.empty_like(gE), torch.empty_like(gb)
    N.check(lib.cqlrec_qhead_bwd(hb.data_ptr(), nlse2.data_ptr(), coef.data_ptr(), act.data_ptr(), U, E_out_b.data_ptr(),
                                 b_out.data_ptr(), Nn, d, float(np.float32(1.0 / U)), ws.data_ptr(), ws_b, r_dH.data_ptr(),
                                 r_gE.data_ptr(), r_gb.data_ptr(), s))
    assert torch.equal(dH, r_dH)
    # the two-pass ABI entry scatters the one-hot part with float atomics: same sums up to their order
    torch.testing.assert_close(gE, r_gE, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(gb, r_gb, rtol=1e-5, atol=1e-7)
    q = ops.qhead_gather_dot(hb, E_out_b, b_out, act)
    assert torch.equal(q, core.pair_scores(hb, act))
    # ---- score_topk with seen filtering: the ctypes path of CQLCore
    seen = torch.as_tensor(np.concatenate([np.concatenate([np.sort(items[off[x]: off[x + 1]]) for x in range(U)]), [0]])
                           .astype(np.int32)).to(DEV)
    idx, val, cnt = ops.score_topk(hb, E_out_b, b_out, k, None, d_off, seen, users)
    ridx, rval, rcnt = core.score_topk(hb, k, seen=(d_off, seen), seen_rows=users)
    assert torch.equal(idx, ridx) and torch.equal(val, rval) and torch.equal(cnt, rcnt)
    idx2, _, _ = ops.score_topk(hb, E_out_b, b_out, k)
    assert torch.equal(idx2, core.score_topk(hb, k)[0])
    # ---- fused_adam_ema: in place, equal to the oracle's expression order bit for bit
    n = 64 * 257
    th = torch.randn(n, device=DEV)
    gr = torch.randn(n, device=DEV) * 0.01
    m_, v_ = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    tg = th.clone()
    thb, tgb = torch.empty(n, dtype=torch.bfloat16, device=DEV), torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ref = [x.cpu().numpy().copy() for x in (th, gr, m_, v_, tg)]
    step_size, sqrt_bc2 = O.adam_scalars(1, 1e-3, 0.9, 0.999)
    ops.fused_adam_ema(th, gr, m_, v_, tg, thb, tgb, float(step_size), float(sqrt_bc2), 0.9, 0.999, 1e-8, 0.005, True)
    with np.errstate(all="ignore"):
        O.adam_ema_step(ref[0], ref[1], ref[2], ref[3], ref[4], 1, 1e-3)
    assert np.array_equal(th.cpu().numpy(), ref[0]) and np.array_equal(m_.cpu().numpy(), ref[2])
    assert np.array_equal(v_.cpu().numpy(), ref[3]) and np.array_equal(tg.cpu().numpy(), ref[4])
    assert np.array_equal(thb.float().cpu().numpy(), O.bf16_round(ref[0])) and torch.count_nonzero(gr).item() == 0
    # ---- argument checks surface as Python RuntimeError (TORCH_CHECK)
    with pytest.raises(RuntimeError, match="BFloat16"):
        ops.qhead_lse_fwd(hb.float(), E_out_b, b_out)
    with pytest.raises(RuntimeError, match="k must be positive"):
        ops.score_topk(hb, E_out_b, b_out, 0)

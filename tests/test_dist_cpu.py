"""World-size-2 data-parallel logic on CPU (gloo): user sharding + SUM all-reduce of the flat gradient buffer +
identical Adam on every rank == one process summing the per-shard gradients.  The gradient math is the oracle's (no
GPU here); the exchange step is the product's replay_cql_amd.dist code path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cql_oracle as O
from replay_cql_amd import dist as PD

U, NI, D_, L, B, STEPS = 120, 300, 64, 6, 32, 3


def _data():
    u, i, t, r = O.synth_log(U, NI, seed=4, mean_len=12, max_len=40)
    return O.build_csr(u, i, t, r, U)


def _shard(off, items, rew, lo, hi):
    a, b = int(off[lo]), int(off[hi])
    return off[lo: hi + 1] - off[lo], items[a:b], rew[a:b]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w, pg = PD.init_from_env("gloo")
    assert (r, w) == (rank, world)
    off, items, rew = _data()
    lo, hi = PD.shard_range(U, rank, world)
    so, si, sr = _shard(off, items, rew, lo, hi)
    m = O.OracleModel.create(NI, D_, seed=7)
    nnz = int(so[-1])
    losses = []
    for step in range(STEPS):
        pos = O.sample_positions(0, step, rank * B, B, nnz)
        users, tpos = O.positions_to_transitions(pos, so)
        out = O.loss_and_grads(m.layout, m.theta, m.target, so, si, sr, users, tpos, L, 0.99, 1.0, grad_scale_batch=B * world)
        g = torch.from_numpy(out.grads)
        PD.allreduce_sum_(g, pg, bucket_elems=10_000 if step % 2 else 0)      # bucketed and single-shot forms
        O.adam_ema_step(m.theta, g.numpy(), m.m, m.v, m.target, step + 1, 1e-3)
        losses.append(PD.max_over_ranks(out.loss, "cpu", pg))
    q.put((rank, m.theta.copy(), losses))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_users():
    for world in (1, 2, 3, 8):
        rs = [PD.shard_range(1_000_003, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == 1_000_003
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        PD.shard_range(10, 2, 2)


def test_two_rank_data_parallel_equals_summed_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])          # replicas stay bit-identical without any broadcast
    # single-process reference: sum of the per-shard gradients, same Adam
    off, items, rew = _data()
    m = O.OracleModel.create(NI, D_, seed=7)
    for step in range(STEPS):
        g = np.zeros_like(m.theta)
        for rank in range(world):
            lo, hi = PD.shard_range(U, rank, world)
            so, si, sr = _shard(off, items, rew, lo, hi)
            pos = O.sample_positions(0, step, rank * B, B, int(so[-1]))
            users, tpos = O.positions_to_transitions(pos, so)
            g += O.loss_and_grads(m.layout, m.theta, m.target, so, si, sr, users, tpos, L, 0.99, 1.0,
                                  grad_scale_batch=B * world).grads
        O.adam_ema_step(m.theta, g, m.m, m.v, m.target, step + 1, 1e-3)
    np.testing.assert_allclose(res[0][1], m.theta, rtol=1e-5, atol=1e-7)


def _rs_ag_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w, pg = PD.init_from_env("gloo")
    n = 96
    region = torch.arange(world * n, dtype=torch.float32) * (rank + 1)       # rank r contributes (r+1) * [0, 1, 2, ...]
    mine = torch.empty(n)
    PD.reduce_scatter_sum(mine, region, pg)
    full = torch.zeros(world * n)
    PD.all_gather_into(full, mine * 2, pg)
    # epoch length agreed from the GLOBAL log size (CQL.fit_arrays): unequal local sizes, one answer on every rank
    assert PD.sum_over_ranks(float(1000 + 7 * rank), "cpu", pg) == float(sum(1000 + 7 * r_ for r_ in range(world)))
    q.put((rank, mine.numpy().copy(), full.numpy().copy()))
    dist.destroy_process_group()


def test_reduce_scatter_and_all_gather_helpers_two_ranks():
    """the exchange steps of the row-sharded optimizer (gloo has no reduce-scatter: the helper falls back to all-reduce +
    own slice, the same sums)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rs_ag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 96
    total = np.arange(world * n, dtype=np.float32) * sum(r + 1 for r in range(world))
    for rank, mine, full in res:
        assert np.array_equal(mine, total[rank * n: (rank + 1) * n])
        assert np.array_equal(full, 2 * total)


def test_shard_plan_world8_with_remainder_rows():
    """Row-sharded optimizer at world = 8 and N mod 8 != 0 (the cfg5 variant's arithmetic, never run on 8 GPUs here):
    own slices tile the two regions, the three replicated tails tile the rest, nothing overlaps, nothing is left out."""
    for n_items, d in ((1003, 64), (100_000, 128), (1_000_003, 256), (7, 64), (8, 128)):
        lay = O.Layout.make(n_items, d)
        plans = [PD.shard_plan(lay.off["E_in"], lay.off["E_out"], lay.off["W1"], lay.total, n_items, d, 8, r)
                 for r in range(8)]
        cover = np.zeros(lay.total, dtype=np.int32)
        p0 = plans[0]
        assert p0["n"] == (n_items // 8) * d
        for key in ("in", "out"):
            reg = p0[key + "_region"]
            assert reg[1] - reg[0] == 8 * p0["n"]
            for r, p in enumerate(plans):
                assert p[key + "_region"] == reg
                lo, hi = p[key + "_own"]
                assert (lo, hi) == (reg[0] + r * p0["n"], reg[0] + (r + 1) * p0["n"])
                cover[lo:hi] += 1
        for key in ("tail_in", "tail_out", "tail_enc"):
            lo, hi = p0[key]
            assert hi >= lo
            cover[lo:hi] += 1
        assert np.all(cover == 1)
        # the replicated E_in tail holds the last N mod 8 rows, the PAD row and the alignment padding
        assert p0["tail_in"][1] - p0["tail_in"][0] >= ((n_items % 8) + 1) * d
    with pytest.raises(ValueError):
        PD.shard_plan(0, 10, 20, 30, 8, 1, 8, 8)


def _plan8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    r, w, pg = PD.init_from_env("gloo")
    n_items, d = 1003, 64
    lay = O.Layout.make(n_items, d)
    P = PD.shard_plan(lay.off["E_in"], lay.off["E_out"], lay.off["W1"], lay.total, n_items, d, world, rank)
    g = torch.from_numpy(np.random.default_rng(100 + rank).standard_normal(lay.total).astype(np.float32))
    shadow = torch.zeros(lay.total, dtype=torch.bfloat16)
    for key in ("in", "out"):          # reduce-scatter -> "Adam" on the own rows (here: bf16 of twice the sum) -> all-gather
        reg, own = P[key + "_region"], P[key + "_own"]
        mine = torch.empty(P["n"])
        PD.reduce_scatter_sum(mine, g[reg[0]: reg[1]], pg)
        stage = (2 * mine).to(torch.bfloat16)
        PD.all_gather_into(shadow[reg[0]: reg[1]], stage, pg)
    for key in ("tail_in", "tail_out", "tail_enc"):
        lo, hi = P[key]
        if hi > lo:
            dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=pg)
            shadow[lo:hi] = (2 * g[lo:hi]).to(torch.bfloat16)
    q.put((rank, shadow.view(torch.int16).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_exchange_eight_ranks_gloo():
    """The exchange of the row-sharded optimizer rehearsed at world = 8 on the CPU backend, N mod 8 = 3: every rank ends
    with the same full shadow, equal to the single-process result."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lay = O.Layout.make(1003, 64)
    tot = np.zeros(lay.total, dtype=np.float32)
    for r in range(world):          # gloo's all-reduce sums in rank order on every rank (ring of 8: not guaranteed);
        tot += np.random.default_rng(100 + r).standard_normal(lay.total).astype(np.float32)
    ref = torch.from_numpy(2 * tot).to(torch.bfloat16).float().numpy()
    for rank, bits in res:
        assert np.array_equal(bits, res[0][1])                       # identical on all ranks
        got = torch.from_numpy(bits).view(torch.bfloat16).float().numpy()
        np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=1e-6)   # one bf16 ulp: the sum order may differ

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

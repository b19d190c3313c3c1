"""Two data-parallel ranks sharing ONE GPU (gloo backend; RCCL refuses two ranks on one device) exercise the whole
multi-rank path of CQLCore on the real kernels: user sharding, rank-offset sampling, phased step with asynchronous
gradient all-reduce, ranged Adam.  Checked against the oracle's summed per-shard gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import cql_oracle as O

pytestmark = pytest.mark.gpu

U, NI, D_, L, B, STEPS = 200, 1000, 128, 8, 128, 4


def _data(n_items=NI):
    u, i, t, r = O.synth_log(U, n_items, seed=8, mean_len=14, max_len=40)
    return O.build_csr(u, i, t, r, U)


def _shard(off, items, rew, lo, hi):
    a, b = int(off[lo]), int(off[hi])
    return off[lo: hi + 1] - off[lo], items[a:b], rew[a:b]


def _worker(rank, world, port, theta0, q, shard=False, n_items=NI, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CQL_DP_EXCHANGE=exchange)
    import torch.distributed as dist
    from replay_cql_amd import dist as PD
    from replay_cql_amd.core import CQLCore, CQLHyper
    r, w, pg = PD.init_from_env("gloo")
    torch.cuda.set_device(0)
    off, items, rew = _data(n_items)
    lo, hi = PD.shard_range(U, rank, world)
    so, si, sr = _shard(off, items, rew, lo, hi)
    core = CQLCore(n_items, CQLHyper(d=D_, window=L, batch=B, seed=5), device="cuda:0", rank=rank, world=world,
                   process_group=pg, shard_optimizer=shard)
    core.load_flat(theta0)
    core.set_log(so, si, sr)
    losses = core.train(STEPS)            # world > 1 -> phased step with async all-reduce
    shadow = core.theta_b.view(torch.int16).cpu().numpy().copy()
    core.sync_full_state()                # no-op unless the optimizer is row-sharded
    res = (rank, core.theta.cpu().numpy(), losses.cpu().numpy(), shadow, core.adam_v.cpu().numpy(),
           core.target.cpu().numpy())
    # the step loop allocates nothing once its persistent buffers exist: three more steps, allocation
    # counters of the caching allocator unchanged
    scratch = torch.zeros(4, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    core.train_steps(3, scratch)
    torch.cuda.synchronize()
    st1 = torch.cuda.memory_stats()
    q.put(res + ((st1["allocation.all.allocated"] - st0["allocation.all.allocated"],
                  st1["num_alloc_retries"] - st0["num_alloc_retries"]),))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, theta0, shard=False, n_items=NI, exchange="allreduce"):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, theta0, q, shard, n_items, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("n_items", [NI, NI + 1])          # N divisible by the world size / one replicated remainder row
def test_row_sharded_optimizer_equals_replicated(n_items):
    """SURVEY 8(e), cfg5 variant: reduce-scatter of the gradients, Adam on this rank's rows of E_in / E_out only,
    all-gather of the bf16 shadows.  Same sums, same Adam arithmetic -> after gathering the shards the masters, moments
    and targets equal the replicated data-parallel run bit for bit, and the shadows every rank computes with are
    identical on all ranks."""
    u, i, t, r = O.synth_log(U, n_items, seed=8, mean_len=14, max_len=40)
    m = O.OracleModel.create(n_items, D_, seed=7)
    theta0 = m.theta.copy()
    # the workers rebuild the log from the module constants: patch the item count through the argument only
    rep = _run(2, theta0, shard=False, n_items=n_items)
    shd = _run(2, theta0, shard=True, n_items=n_items)
    assert np.array_equal(shd[0][3], shd[1][3])                  # bf16 shadows identical on both ranks
    assert np.array_equal(shd[0][3], rep[0][3])                  # ... and equal to the replicated run's
    for k in (1, 4, 5):                                          # theta, adam_v, target after sync_full_state()
        assert np.array_equal(shd[0][k], shd[1][k]) and np.array_equal(shd[0][k], rep[0][k]), k
    assert np.array_equal(shd[0][2], rep[0][2])                  # same losses
    # allocation-free step loops.  The replicated loop: nothing at all.  The sharded loop: nothing of its
    # own either (persistent gradient shards, staging and pack buffers) -- what may remain under THIS backend is gloo's
    # internal flattening of a device all-gather (one temporary per all-gather, 4 per step, absent under RCCL where
    # all_gather_into_tensor writes in place); anything beyond that is a regression.
    for rk in rep:
        assert rk[6] == (0, 0), rk[6]
    for rk in shd:
        assert rk[6][1] == 0 and rk[6][0] in (0, 4 * 3), rk[6]


def test_replicated_exchange_as_reduce_scatter_plus_all_gather():
    """CQL_DP_EXCHANGE=rsag: the gradient sum of the replicated variant travels as reduce-scatter into a persistent
    1/W shard + all-gather back.  Over gloo the helper's fallback forms the same sums as the all-reduce,
    so the run must equal the all-reduce run bit for bit; replicas identical; no allocation in the loop."""
    m = O.OracleModel.create(NI, D_, seed=7)
    theta0 = m.theta.copy()
    a = _run(2, theta0, exchange="allreduce")
    b = _run(2, theta0, exchange="rsag")
    for k in (1, 2, 3, 4, 5):
        assert np.array_equal(b[0][k], b[1][k]) or k == 2, k
        assert np.array_equal(a[0][k], b[0][k]), k
    # nothing of the loop's own; under gloo the device all-gather of each of the three regions may keep one temporary
    assert a[0][6] == (0, 0) and a[1][6] == (0, 0)
    for rk in b:
        assert rk[6][1] == 0 and rk[6][0] in (0, 3 * 3), rk[6]


def test_two_ranks_one_gpu_match_oracle():
    world = 2
    m = O.OracleModel.create(NI, D_, seed=7)
    theta0 = m.theta.copy()
    res = _run(world, theta0)
    assert np.array_equal(res[0][1], res[1][1])            # replicas bit-identical after identical Adam
    assert np.allclose(res[0][2], res[1][2])               # the reduced loss is the same on both ranks
    off, items, rew = _data()
    ref_losses = []
    from replay_cql_amd import dist as PD
    for step in range(STEPS):
        g = np.zeros_like(m.theta)
        loss = 0.0
        for rank in range(world):
            lo, hi = PD.shard_range(U, rank, world)
            so, si, sr = _shard(off, items, rew, lo, hi)
            pos = O.sample_positions(5, step, rank * B, B, int(so[-1]))
            users, tpos = O.positions_to_transitions(pos, so)
            out = O.loss_and_grads(m.layout, m.theta, m.target, so, si, sr, users, tpos, L, 0.99, 1.0,
                                   grad_scale_batch=B * world)
            g += out.grads
            loss += out.loss
        O.adam_ema_step(m.theta, g, m.m, m.v, m.target, step + 1, 1e-3)
        ref_losses.append(loss)
    np.testing.assert_allclose(res[0][2], ref_losses, rtol=1e-3)
    assert np.linalg.norm(res[0][1] - m.theta) < 1e-3 * np.linalg.norm(m.theta)


def _fit_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from replay_cql_amd import dist as PD
    from replay_cql_amd.cql import CQL
    r, w, pg = PD.init_from_env("gloo")
    torch.cuda.set_device(0)
    off, items, rew = _data()
    cut = int(U * 0.62)                       # UNEQUAL shards: a per-rank epoch length would differ between the ranks
    lo, hi = (0, cut) if rank == 0 else (cut, U)
    m = CQL(embedding_dim=D_, window=L, batch_size=B, epochs=3, valid_split_size=0.2, patience=0, factor=0.5, seed=5,
            device="cuda:0")
    m.set_distributed(rank, world, pg)
    m.fit_arrays(*_shard(off, items, rew, lo, hi), NI)
    q.put((rank, m.core.theta.cpu().numpy(), m.train_losses, m.valid_losses, m.best_epoch, m.core.hyper.lr,
           int(_shard(off, items, rew, lo, hi - int((hi - lo) * 0.2))[0][-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_arrays_data_parallel_unequal_shards():
    """CQL.fit_arrays under data parallelism -- the epoch length comes from the GLOBAL log size (all ranks
    issue the same number of all-reduces), the validation loss, the plateau cut and the best-epoch choice are agreed
    across ranks: replicas end bit-identical, with identical bookkeeping."""
    import math
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res
    assert a[6] != b[6]                                           # the local training logs really differ in size
    steps = 3 * math.ceil((a[6] + b[6]) / (B * world))
    assert len(a[2]) == len(b[2]) == steps
    assert np.array_equal(a[1], b[1])                             # replicas bit-identical
    assert np.array_equal(a[3], b[3]) and a[4] == b[4] and a[5] == b[5]
    assert np.allclose(a[2], b[2])                                # the reduced loss


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver's scaling run invokes it): the parent --
    which never touches the GPU -- starts one child per rank, rank 0 prints the ONE JSON line, the exit code is the
    children's worst.  Rehearsed with both ranks on this one GPU over gloo (RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(CQL_DIST_BACKEND="gloo", CQL_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--config", "cfg2", "--steps", "10",
                        "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 10 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["dp_variant"] in ("sharded", "allreduce")
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]

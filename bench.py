#!/usr/bin/env python3
"""bench.py -- CQL hot-path throughput on MI355X (BASELINE.json metric: CQL train-steps/sec + top-K users/sec on a
1M-user x 100K-item synthetic log).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full CQL training step over one batch of B=4096 transitions per GPU (sample, window gather, encoder,
full-catalog Q-head LSE + argmax, double-Q TD target, CQL loss, backward, [RCCL gradient all-reduce], Adam + Polyak).
Weak scaling: per-GPU batch fixed, users sharded by rank, no data-path collective besides the gradient all-reduce.
`value` = (N * K steps of 4096 transitions) / max-over-ranks time, inputs resident in HBM.  After the timed training
region the all-users top-K pass (K=10, seen items filtered) is timed on a user block and reported in `topk`.

The oracle (oracle/) is used ONLY for the cpu_baseline leg (rank 0, N=1), never in the measured path."""
from __future__ import annotations

import argparse
import json
import os

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a data-parallel rank has five that must
# run concurrently (main, item side, forward branch, RCCL, torch's copy stream): ask for more before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def _self_launch_if_needed():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the
    launcher.  It starts N fresh children of this same script -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, rendezvous on 127.0.0.1 -- BEFORE torch is imported, so the parent never touches the
    GPU (no exec from a process that has initialised HIP).  Rank 0 inherits stdout and prints the ONE JSON line; the
    parent exits with the worst child return code, and stops the remaining ranks (by their exact PIDs) as soon as one
    has failed, so a crashed rank cannot leave the others waiting in a collective."""
    if "WORLD_SIZE" in os.environ:
        return
    n, argv = 1, sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                print(f"[bench] rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in sorted(live):
                    procs[o].terminate()
        time.sleep(0.05)
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:      # a rank that ignores SIGTERM (stuck in a collective)
            pr.kill()
    sys.exit(worst)


if __name__ == "__main__":
    _self_launch_if_needed()

import torch  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[2] (the configuration the metric is quoted on) and configs[1]
    "cfg3": dict(users=1_000_000, items=100_000, d=128, window=50, batch=4096, k=10, topk_users=131_072),
    "cfg2": dict(users=100_000, items=10_000, d=64, window=50, batch=4096, k=10, topk_users=65_536),
    "tiny": dict(users=2_000, items=1_000, d=64, window=10, batch=256, k=10, topk_users=2_000),
    # BASELINE.json configs[0] shape (MovieLens-1M: 6 040 users, 3 883 items, ~836 K events -> long histories)
    "cfg1": dict(users=6_040, items=3_883, d=64, window=50, batch=4096, k=10, topk_users=6_040,
                 mean_len=92.0, sigma=0.9, max_len=2000),
    # BASELINE.json configs[4] as ONE GPU of the 8 sees it: the full 1 M-item, d=256 tables (512 MB each: beyond the
    # Infinity Cache, the honest HBM case for the window gather) and 1/8 of the 10 M users
    "cfg5shard": dict(users=1_250_000, items=1_000_000, d=256, window=50, batch=4096, k=10, topk_users=16_384),
}
WORKLOAD = {"cfg3": "BASELINE.json configs[2]", "cfg2": "BASELINE.json configs[1]", "cfg1": "BASELINE.json configs[0] shape",
            "cfg5shard": "BASELINE.json configs[4], one GPU's share (users / 8, full tables)", "tiny": "tiny"}
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-topk", action="store_true")
    ap.add_argument("--topk-chunk", type=int, default=None,
                    help="users per launch of the top-K scoring kernel (default: the config's topk_users)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--dp-variant", default="auto", choices=["auto", "sharded", "allreduce"],
                    help="ranks > 1: row-sharded optimizer (reduce-scatter / own-row Adam / bf16 all-gather), replicated "
                         "Adam behind a gradient all-reduce, or auto = sharded behind ONE probe step whose outcome all "
                         "ranks agree on (recorded in config.dp_variant / dp_probe)")
    ap.add_argument("--serial", action="store_true",
                    help="no intra-step concurrency for the whole run (per-kernel durations then match rocprofv3)")
    return ap.parse_args()


def cpu_baseline(cfg, off, items, rew, budget_s=20.0):
    """Oracle (numpy restatement, multi-threaded BLAS) on the host cores: identical algorithm, identical batch.
    Returns the baseline object and what the parity check needs (initial parameters, per-step losses, top-K)."""
    import numpy as np
    from oracle import cql_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:  # pragma: no cover
        cores = os.cpu_count() or 1
    m = O.OracleModel.create(cfg["items"], cfg["d"], seed=7)
    theta0 = m.theta.copy()
    losses = []
    t0, n = time.perf_counter(), 0
    while n < 4 and (n == 0 or time.perf_counter() - t0 < budget_s):
        with np.errstate(all="ignore"):
            losses += O.train_steps(m, off, items, rew, 1, cfg["batch"], cfg["window"], seed=0, fast=True)
        n += 1
    dt = time.perf_counter() - t0
    # top-K on a small user sample
    nu = min(256, len(off) - 1)
    t1 = time.perf_counter()
    tk = O.predict_topk(m.layout, m.theta, off, items, np.arange(nu), cfg["k"], cfg["window"], filter_seen=True, fast=True)
    dtk = time.perf_counter() - t1
    base = {"value": n / dt, "unit": "train-steps/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} oracle train steps (numpy, B={cfg['batch']}, N={cfg['items']}, d={cfg['d']}) on a "
                      f"{len(off) - 1}-user shard of the same synthetic log; top-K on {nu} users",
            "topk_users_per_s": nu / dtk}
    return base, {"theta0": theta0, "losses": losses, "n": n, "nu": nu, "topk": tk[:3]}


def parity_check(cfg, dev, d_off, d_items, d_rew, ref):
    """The HIP path on the SAME shard, initial parameters, seed and steps as the oracle's cpu_baseline leg: per-step
    losses (rtol 1e-3, P4) and the top-K lists of the trained model (P3 margin rule, 2e-3 because the two parameter
    sets are 1e-3 apart after training).  Outside the timed region; reported, and `ok` summarises it."""
    import numpy as np
    from replay_cql_amd.core import CQLCore, CQLHyper
    from replay_cql_amd.data import sorted_seen
    core = CQLCore(cfg["items"], CQLHyper(d=cfg["d"], window=cfg["window"], batch=cfg["batch"], seed=0), device=dev)
    core.load_flat(ref["theta0"])
    core.set_log(d_off, d_items, d_rew)
    g_losses = core.train(ref["n"]).cpu().numpy().astype(np.float64)
    o_losses = np.asarray(ref["losses"], dtype=np.float64)
    rel = np.abs(g_losses - o_losses) / np.abs(o_losses)
    nu, k = ref["nu"], cfg["k"]
    users = torch.arange(nu, dtype=torch.int32, device=dev)
    seen = torch.as_tensor(np.concatenate([sorted_seen(d_off.cpu().numpy(), d_items.cpu().numpy()), [0]])
                           .astype(np.int32)).to(dev)
    hb = core.encode(d_off, d_items, users)
    idx, val, cnt = (t.cpu().numpy() for t in core.score_topk(hb, k, seen=(d_off, seen), seen_rows=users))
    ridx, rval, rcnt = ref["topk"]
    same, worst, viol = 0, 0.0, 0
    for u in range(nu):
        got, want = dict(zip(idx[u], val[u])), dict(zip(ridx[u], rval[u]))
        same += set(got) == set(want)
        for j in got.keys() & want.keys():
            worst = max(worst, abs(float(got[j]) - float(want[j])))
        for j in got.keys() - want.keys():
            viol += abs(float(got[j]) - float(rval[u, -1])) >= 2e-3
        for j in want.keys() - got.keys():
            viol += abs(float(want[j]) - float(val[u, -1])) >= 2e-3
    ok = bool(rel.max() < 1e-3 and viol == 0 and worst < 2e-3 and np.array_equal(cnt, rcnt))
    return {"ok": ok, "steps": int(ref["n"]), "gpu_losses": g_losses.tolist(), "oracle_losses": o_losses.tolist(),
            "loss_max_rel_err": float(rel.max()), "loss_rtol": 1e-3,
            "topk_users": int(nu), "topk_sets_identical": int(same), "topk_margin_violations": int(viol),
            "topk_max_abs_score_diff": worst, "topk_margin": 2e-3,
            "note": "oracle = this repo's CPU restatement (parity unpinned by the reference: it has no CQL path)"}


INFINITY_CACHE_BYTES = 256 << 20


def hbm_roofline(alg_bytes, ms, table_bytes, **extra):
    """HBM roofline of one launch from ALGORITHMIC bytes.  `frac` is only a statement about HBM when the kernel's table
    cannot sit in the 256 MiB Infinity Cache: for a resident table the algorithmic rate may exceed the HBM peak (rows
    come from cache), so frac is null there and profiles/r02_hbm_traffic.json holds the measured FETCH_SIZE/WRITE_SIZE
    evidence on tables of 512 MB - 1 GiB (gather: 5.6 TB/s Zipf, 7.5 TB/s uniform; Adam: 4.9 TB/s, traffic = algorithmic)."""
    ach = alg_bytes / (ms * 1e-3) / 1e9
    resident = table_bytes < INFINITY_CACHE_BYTES
    r = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
         "frac": None if resident else min(ach / PEAK_HBM_GBS, 1.0), "traffic": None, "avg_ms": ms,
         "algorithmic_bytes": int(alg_bytes), "table_bytes": int(table_bytes), "infinity_cache_resident": resident,
         "big_table_evidence": "profiles/r02_hbm_traffic.json"}
    r.update(extra)
    return r



def main():
    args = parse()
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:      # (a bare `--gpus N` was turned into N ranks by _self_launch_if_needed above)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU with gloo, to exercise the N > 1 code path
    backend = os.environ.get("CQL_DIST_BACKEND", "nccl")
    if os.environ.get("CQL_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    # the library's streams are created BEFORE torch.distributed creates RCCL's (cqlrec_runtime_init, include/cqlrec.h:
    # streams created late share the default stream's hardware queue and the step loses its concurrency)
    from replay_cql_amd import _native as N
    if not os.environ.get("CQL_SKIP_EARLY_INIT"):
        N.runtime_init()
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a bounded timeout: a rank that fails on its own (before a collective its peers already sit in) makes the job
        # FAIL after this long instead of hanging it
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("CQL_DIST_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=tmo)
        pg = dist.group.WORLD

    from replay_cql_amd.core import CQLCore, CQLHyper
    from replay_cql_amd.data import synth_log_device

    lib = N.load()
    U, NI, d, L, B, K = cfg["users"], cfg["items"], cfg["d"], cfg["window"], cfg["batch"], cfg["k"]
    lo, hi = rank * U // world, (rank + 1) * U // world
    gen_kw = {k_: cfg[k_] for k_ in ("mean_len", "sigma", "max_len") if k_ in cfg}

    def gen_log():
        return synth_log_device(U, NI, seed=12345, device=dev, user_lo=lo, user_hi=hi, **gen_kw)
    # Order of the set-up (nothing of it is timed): the model first -- its parameters are drawn on the host and uploaded --
    # and the log last, generated on the device right in front of the warm-up steps.  (It does not shorten the ramp: the
    # first ~30 steps of a process run 3-12 % slower than the steady state whatever the GPU did before them -- idle, 150 ms
    # of GEMMs, a predict pass: tools/step_ramp.py, profiles/r03_step_ramp.json -- so `--steps 20 --warmup 5` reads ~5 %
    # below `--steps 200 --warmup 20`; README quotes the former, the driver's condition.)
    # Ranks > 1, two exchange patterns with the same bytes on the links: "sharded" = reduce-scatter of the
    # gradients, Adam on this rank's rows only, all-gather of the bf16 shadows (1/W of the Adam traffic per GPU);
    # "allreduce" = replicated Adam behind a gradient all-reduce.  The variant is chosen by --dp-variant and the run
    # FAILS if it does not work -- except under "auto", where one probe step of the sharded variant is tried first and
    # ALL ranks agree (MAX all-reduce of a failure flag) whether to keep it; the outcome is part of the JSON line.
    want = args.dp_variant if world > 1 else "single"
    if os.environ.get("CQL_SHARD_OPTIMIZER") == "0":       # round-1 knob, kept
        want = "allreduce" if world > 1 else want
    dp_probe = None

    def make_core(shard, log=None):
        c_ = CQLCore(NI, CQLHyper(d=d, window=L, batch=B, seed=0), device=dev, rank=rank, world=world, process_group=pg,
                     shard_optimizer=shard)
        if log is not None:
            c_.set_log(*log)
        return c_
    if want == "auto":
        off, items, rew = gen_log()
        import torch.distributed as dist
        failed, why = 0.0, ""
        try:
            probe = make_core(True, (off, items, rew))
            probe.train_steps(1)
            torch.cuda.synchronize()
        except Exception as exc:  # pragma: no cover - needs a multi-GPU RCCL job
            failed, why = 1.0, repr(exc)
            print(f"[bench] rank {rank}: row-sharded optimizer probe step failed: {why}", file=sys.stderr, flush=True)
        flag = torch.tensor([failed], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        want = "allreduce" if float(flag.item()) > 0 else "sharded"
        dp_probe = {"sharded_probe_failed_on_some_rank": bool(float(flag.item()) > 0), "rank0_error": why}
        probe = None                                        # the probe step is discarded: the timed run starts fresh
        del off, items, rew
        torch.cuda.empty_cache()
    core = make_core(want == "sharded")
    off, items, rew = gen_log()
    core.set_log(off, items, rew)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    loss_buf = torch.zeros(args.warmup + args.steps + 1, device=dev)
    issue_s = [0.0]

    def run(n, base):
        # <=64 steps per call, pipelined across steps: one library call (single rank: the step loop is in C++,
        # cqlrec_train_steps) or the phased data-parallel loop around the two asynchronous gradient all-reduces
        t_issue = time.perf_counter()
        for lo_ in range(0, n, 64):
            core.train_steps(min(64, n - lo_), loss_buf[base + lo_:])
        issue_s[0] += time.perf_counter() - t_issue      # host time spent ENQUEUEING (no sync inside train_steps)

    if args.serial:
        N.check(lib.cqlrec_set_concurrency(0), "set_concurrency")
    run(args.warmup, 0)
    barrier()
    # Inside the timed region only the kernel the `roofline` object reports on is bracketed with HIP events: every event pair costs two barrier packets on its stream, and bracketing all ~35
    # launches of a step slows the step by ~14 % (measured).  All phases are bracketed in the serialised pass below.
    if not args.no_prof:
        dom = 1 << N.PHASES.index("qhead_bwd_de")
        N.check(lib.cqlrec_prof_select(0xFFFFFFFF if args.serial else dom), "prof_select")
        N.check(lib.cqlrec_prof_enable(1), "prof_enable")
    issue_s[0] = 0.0
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    host_issue_ms_per_step = 1e3 * issue_s[0] / max(args.steps, 1)
    barrier()
    dt = time.perf_counter() - t0
    phases_timed = N.prof_read() if not args.no_prof else {}
    N.check(lib.cqlrec_prof_enable(0), "prof_enable")
    N.check(lib.cqlrec_prof_select(0xFFFFFFFF), "prof_select")
    # Per-kernel durations for the roofline: inside the timed region independent kernels overlap on side streams, so a
    # kernel's event-bracketed duration there includes what it shared the chip with.  A short serialised pass in the
    # same process (same data, same shapes, not part of `value`) gives each kernel's own launch duration.
    phases = phases_timed
    n_prof_steps = args.steps
    if not args.no_prof and not args.serial:
        n_prof_steps = 10
        N.check(lib.cqlrec_set_concurrency(0), "set_concurrency")
        spare = torch.zeros(n_prof_steps + 1, device=dev)
        core.train_step(spare[:1])
        barrier()
        N.check(lib.cqlrec_prof_enable(1), "prof_enable")
        for i in range(n_prof_steps):
            core.train_step(spare[i + 1: i + 2])
        barrier()
        phases = N.prof_read()
        N.check(lib.cqlrec_prof_enable(0), "prof_enable")
        N.check(lib.cqlrec_set_concurrency(1), "set_concurrency")
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = loss_buf[args.warmup: args.warmup + args.steps].cpu().tolist()

    # ---- top-K scoring pass (second half of the metric) --------------------------------------------------------
    topk = None
    if not args.no_topk:
        # all users of this rank's shard, in chunks of cfg["topk_users"] (one launch of the scoring kernel each)
        nu = hi - lo
        tk_chunk = min(args.topk_chunk or cfg["topk_users"], nu)
        users = torch.arange(nu, dtype=torch.int32, device=dev)
        # seen lists = items sorted inside each user's row (input preparation, untimed)
        rows = torch.repeat_interleave(torch.arange(hi - lo, device=dev), off[1:] - off[:-1])
        seen_items = items[torch.argsort(rows * NI + items.to(torch.int64))].contiguous()
        del rows
        core.encode_topk(off, items, users[:1024], K, seen=(off, seen_items), chunk=tk_chunk)      # warm-up
        barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            idx, val, cnt = core.encode_topk(off, items, users, K, seen=(off, seen_items), chunk=tk_chunk)
        barrier()
        dtk = time.perf_counter() - t1
        # per-kernel durations from one more, event-bracketed pass (not part of `value`: the brackets cost ~8 % here)
        tk_ph = {}
        if not args.no_prof:
            N.check(lib.cqlrec_prof_enable(1), "prof_enable")
            core.encode_topk(off, items, users, K, seen=(off, seen_items), chunk=tk_chunk)
            barrier()
            tk_ph = N.prof_read()
            N.check(lib.cqlrec_prof_enable(0), "prof_enable")
        tk_reps = 1
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dtk], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtk = float(t.item())
        topk = {"metric": "top-K users/sec", "value": world * reps * nu / dtk, "unit": "users/s", "k": K,
                "users_per_rank": nu, "filter_seen": True, "ms_per_pass": 1e3 * dtk / reps,
                # the whole pass (encode + seen bitmap + scoring/selection + list merge) against the bf16 MFMA peak
                "pass_mfma_frac": 2.0 * nu * NI * d / (dtk / reps) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
        if tk_ph.get("topk_tilemax", (0, 0))[1]:
            ms = tk_ph["topk_tilemax"][0] / tk_ph["topk_tilemax"][1]
            launches_per_pass = tk_ph["topk_tilemax"][1] / tk_reps
            fl = 2.0 * nu * NI * d / launches_per_pass
            fused = (d == 128 and K <= 16)
            # `frac` = the PASS-level fraction (encode + seen bitmap + scoring/selection + merge, all users of the shard);
            # the scoring kernel's own launches are reported beside it
            pass_tf = 2.0 * nu * NI * d / (dtk / reps) / 1e12
            topk["roofline"] = {"kernel": ("qtopk4_kernel<128>" if tk_chunk >= 512 * 160 else "qtopk2_kernel<128>") +
                                " (scores + on-chip top-k selection)" if fused
                                else "qstream_kernel<TOPK> / <TILEMAX>", "bound": "mfma",
                                "achieved": pass_tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": pass_tf / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                                "scope": "whole pass: 2*U*N*d / ms_per_pass",
                                "kernel_avg_ms": ms, "kernel_launches_per_pass": launches_per_pass,
                                "kernel_frac": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
            topk["select_ms_per_pass"] = tk_ph["topk_select"][0] / tk_reps
        if tk_ph.get("gather_fwd", (0, 0))[1]:
            # the window gather at a size that fills the chip: one launch over all `nu` users of the scoring pass
            # (the training step's gathers cover only B = 4096 states and are launch/latency bound)
            g_n = tk_ph["gather_fwd"][1]
            g_ms = tk_ph["gather_fwd"][0] / g_n
            lens_u = (off[1: nu + 1] - off[:nu]).clamp(max=L).float().mean().item()
            gb = (nu * (lens_u * d * 2 + lens_u * 4) + nu * d * 2) / g_n      # rows + indices in, bf16 state out, per launch
            topk["roofline_gather"] = hbm_roofline(gb, g_ms, table_bytes=NI * d * 2, states_per_launch=nu // g_n,
                                                   note="runs on the side stream beside the scoring kernel of the previous chunk")

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    out = {
        "metric": "CQL train-steps/sec", "value": world * args.steps / dt, "unit": "train-steps/s (B=4096 transitions per step and GPU)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{WORKLOAD[args.config]}: synthetic {U} users x {NI} items, CQL d={d}, L={L}, "
                               f"B={B}/GPU, bf16 MFMA + fp32 accumulate",
                   "users": U, "items": NI, "d": d, "window": L, "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": f"dp{world} (users sharded by rank, RCCL gradient " +
                                  ("reduce-scatter, row-sharded Adam, bf16 all-gather)" if core.shard_optimizer else "all-reduce)"), "k": K,
                   "dp_variant": want, "dp_probe": dp_probe},
        "transitions_per_sec": world * args.steps * B / dt,
        # time outside kernels: what the host (rank 0) spent enqueueing a step (launches, collectives' issue, Python);
        # it runs ahead of the device, so it only bounds the step when it exceeds ms_per_step
        "host_issue_ms_per_step": host_issue_ms_per_step,
        "loss_first_last": [losses[0], losses[-1]] if losses else None,
    }
    if phases:
        out["kernel_ms_per_step"] = {p: round(ms / n_prof_steps, 4) for p, (ms, n) in phases.items() if n}
        out["kernel_ms_note"] = ("serialised pass of %d steps after the timed region (intra-step concurrency off); "
                                 "in the timed region independent kernels overlap" % n_prof_steps) if not args.serial \
            else "timed region (run with --serial)"
        if phases_timed is not phases:
            out["kernel_ms_per_step_overlapped"] = {p: round(ms / args.steps, 4) for p, (ms, n) in phases_timed.items() if n}
        # Q-head kernels of a step: the fused forward of branch A (phase "qhead_lse": logsumexp AND the softmax-weighted
        # item sum, i.e. the forward GEMM + the dH GEMM of SURVEY 8(d) in one catalogue pass), the argmax of branch B, the
        # item-side backward (dE_out).  "qhead_bwd_dh" only appears when the two-pass ABI entry point is used.
        qk = {p: phases[p][0] / phases[p][1] for p in ("qhead_lse", "qhead_argmax", "qhead_bwd_dh", "qhead_bwd_de")
              if phases[p][1]}
        gemms = {"qhead_lse": 2, "qhead_argmax": 1, "qhead_bwd_dh": 1, "qhead_bwd_de": 1}   # algorithmic GEMMs per launch
        dom = max(qk, key=qk.get)
        flops = 2.0 * B * NI * d          # algorithmic flops of ONE Q-head GEMM (8*B*N*d per step = 4 GEMMs)
        ach = gemms[dom] * flops / (qk[dom] * 1e-3) / 1e12
        kname = {"qhead_lse": ("qfwd2_kernel<128>" if d == 128 else "qfwd3_kernel<256>" if d == 256 else "qstream_kernel<QM_LSE_DH>")
                 + " (fused forward: lse + softmax-weighted item sum)",
                 "qhead_bwd_de": ("qde2_kernel<128>" if (d == 128 and B % 64 == 0) else "qde3_kernel<256>" if (d == 256 and B % 32 == 0)
                                  else "qde_kernel") + " (item-side backward)",
                 "qhead_argmax": "qargmax2_kernel<256>" if d == 256 else "qstream_kernel<QM_ARGMAX>"}.get(dom, dom)
        out["roofline"] = {"kernel": kname, "phase": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "avg_ms": qk[dom],
                           "algorithmic_flops_per_launch": gemms[dom] * flops}
        # `traffic`: HBM bytes per launch from the PMC counters -- collected in separate rocprofv3 --pmc passes (they cannot
        # be read from inside this process) and committed with their provenance; only valid for the shape they were taken on
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))          # a malformed file is an error, not a silent null
            hit = [e for e in pmc["entries"] if e["config"] == {"batch": B, "items": NI, "d": d} and dom in e["kernels"]]
            if hit:
                out["roofline"]["traffic"] = hit[0]["kernels"][dom]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_note"] = pmc["source"] + "; " + pmc["correction"]
            else:
                out["roofline"]["traffic_note"] = "profiles/pmc_traffic.json holds no counters for this shape/kernel"
        else:
            out["roofline"]["traffic_note"] = "profiles/pmc_traffic.json not found"
        out["roofline_qhead_kernels"] = {
            p: {"avg_ms": round(ms, 4), "algorithmic_tflops": round(gemms[p] * flops / (ms * 1e-3) / 1e12, 1),
                "frac": round(gemms[p] * flops / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)} for p, ms in qk.items()}
        if phases_timed is not phases and phases_timed.get(dom, (0, 0))[1]:
            # the same kernel bracketed inside the timed region, where it shares the chip with the state-side backward
            out["roofline"]["avg_ms_timed_region_overlapped"] = phases_timed[dom][0] / phases_timed[dom][1]
        q_ms = sum(qk.values())
        out["roofline_qhead_step"] = {"flops_per_step": 4 * flops, "ms_per_step": q_ms,
                                      "achieved": 4 * flops / (q_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                      "frac": 4 * flops / (q_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
        # HBM-bound companions (algorithmic bytes)
        lens = (off[1:] - off[:-1]).clamp(max=L).float().mean().item()
        g_bytes = B * (lens * d * 2 + lens * 4) + B * d * 4
        if phases["gather_fwd"][1]:
            g_ms = phases["gather_fwd"][0] / phases["gather_fwd"][1]
            out["roofline_gather"] = hbm_roofline(g_bytes, g_ms, table_bytes=NI * d * 2, states_per_launch=B)
        if phases["adam"][1]:
            a_ms = phases["adam"][0] / phases["adam"][1]
            a_bytes = int(core.layout.total) * 44
            out["roofline_adam"] = hbm_roofline(a_bytes, a_ms, table_bytes=int(core.layout.total) * 20)
    if topk:
        out["topk"] = topk
    if world == 1 and not args.no_cpu_baseline:
        import numpy as np
        n_shard = min(hi - lo, 20_000)
        o_h = off[: n_shard + 1].cpu().numpy()
        nz = int(o_h[-1])
        cb, ref = cpu_baseline(cfg, o_h, items[:nz].cpu().numpy(), rew[:nz].cpu().numpy())
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = out["value"] / cb["value"]
        out["parity_check"] = parity_check(cfg, dev, off[: n_shard + 1].contiguous(), items[:nz].contiguous(),
                                           rew[:nz].contiguous(), ref)
    print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

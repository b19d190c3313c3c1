#!/usr/bin/env python3
"""Where the wall time of a cfg3 train step goes (un-profiled, host clock around synchronised loops):
forward only / forward+backward / full step, with and without intra-step concurrency.

    python tools/phase_timing.py [--reps 200]
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N  # noqa: E402
from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--trace-only", action="store_true", help="just run 3 x train_steps(32) (rocprofv3 target)")
    a = ap.parse_args()
    lib = N.load()
    dev = torch.device("cuda:0")
    off, items, rew = synth_log_device(a.users, a.items, seed=12345, device=dev)
    core = CQLCore(a.items, CQLHyper(d=a.d, window=50, batch=a.batch, seed=0), device=dev)
    core.set_log(off, items, rew)
    for _ in range(20):
        core.train_step()
    torch.cuda.synchronize()
    if a.trace_only:
        for _ in range(3):
            core.train_steps(32)
        torch.cuda.synchronize()
        return
    c = core._train_ctx()
    s = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3

    def fwd():
        N.check(lib.cqlrec_train_step_forward(C.byref(c), core.step, None, s))

    def fwd_bwd():
        N.check(lib.cqlrec_train_step_fwd_bwd(C.byref(c), core.step, None, s))

    def upd():
        N.check(lib.cqlrec_train_step_update(C.byref(c), core.step, s))

    def full():
        fwd_bwd()
        upd()

    lb = torch.zeros(1, device=dev)
    print(f"core.train_step(): {timed(lambda: core.train_step(lb)):.3f} ms", flush=True)
    a.reps, keep = 8, a.reps
    t64s = sorted(timed(lambda: core.train_steps(64)) / 64 for _ in range(5))
    a.reps = keep
    print(f"core.train_steps(64): min {t64s[0]:.4f}  median {t64s[2]:.4f} ms/step (5 x 512 steps)", flush=True)
    a.reps, keep = 8, a.reps
    tph = sorted(timed(lambda: core.train_steps(64, phased=True)) / 64 for _ in range(3))
    a.reps = keep
    print(f"core.train_steps(64, phased=True) [the data-parallel step loop, one rank, no collective]: "
          f"min {tph[0]:.4f} ms/step", flush=True)
    N.check(lib.cqlrec_debug_marks_enable(1))
    core.train_steps(64)
    torch.cuda.synchronize()
    ms = (C.c_float * 12)()
    N.check(lib.cqlrec_debug_marks_read(ms))
    N.check(lib.cqlrec_debug_marks_enable(0))
    names = ("loss", "dH", "dE_out", "enc+gather bwd", "adam E_in", "adam E_out", "next prologue", "next LSE", "next loss", "enc dx", "gather bwd",
             "next sample+sort")
    print("schedule marks of one pipelined step (us after the loss): " +
          "  ".join(f"{n}={1e3 * ms[i]:.0f}" for i, n in enumerate(names)), flush=True)
    for conc in (1, 0):
        N.check(lib.cqlrec_set_concurrency(conc))
        print(f"concurrency={conc}: forward {timed(fwd):.3f} ms   fwd+bwd {timed(fwd_bwd):.3f} ms   "
              f"update {timed(upd):.3f} ms   full {timed(full):.3f} ms", flush=True)


if __name__ == "__main__":
    main()

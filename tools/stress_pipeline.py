#!/usr/bin/env python3
"""Race check at full size: the pipelined step driver (cross-step overlap, side streams, double-buffered step vectors)
and the pipelined data-parallel phase path (one rank) against strict program order on ONE stream, same seed, cfg3
shapes.  The step has no float atomics (sorted segmented sums, ordered slab reductions), so all three must agree BIT
FOR BIT after hundreds of steps -- any race, however rare, breaks that.

    python tools/stress_pipeline.py [--steps 300]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N  # noqa: E402
from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    lib = N.load()
    dev = torch.device("cuda:0")
    off, items, rew = synth_log_device(a.users, a.items, seed=12345, device=dev)

    def make():
        c = CQLCore(a.items, CQLHyper(d=a.d, window=50, batch=a.batch, seed=3), device=dev)
        c.set_log(off, items, rew)
        return c
    A, B = make(), make()
    assert torch.equal(A.theta, B.theta)
    la = A.train(a.steps)                               # pipelined, chunks of 64
    lp = torch.zeros(a.steps, device=dev)
    B2 = make()
    B2.train_steps(a.steps, lp, phased=True)            # the data-parallel phase path on one rank, pipelined as well
    N.check(lib.cqlrec_set_concurrency(0))
    lb = torch.zeros(a.steps, device=dev)
    for i in range(a.steps):
        B.forward_backward(lb[i:i + 1])
        B.apply_update()
    N.check(lib.cqlrec_set_concurrency(1))
    torch.cuda.synchronize()
    ok = True
    for name, l, core in (("train_steps", la, A), ("phased", lp, B2)):
        same = all(torch.equal(getattr(core, n), getattr(B, n)) for n in ("theta", "target", "adam_m", "adam_v", "theta_b",
                                                                          "target_b")) and torch.equal(l, lb)
        ok = ok and same and bool(torch.isfinite(l).all())
        print(f"{name:12s} vs strict program order over {a.steps} steps: "
              f"{'bit-identical parameters, optimizer state and losses' if same else 'MISMATCH'}"
              f" (max |dtheta| {(core.theta - B.theta).abs().max().item():.2e})", flush=True)
    print(f"loss first/last: {lb[0].item():.4f} -> {lb[-1].item():.4f}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

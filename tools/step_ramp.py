#!/usr/bin/env python3
"""How long after start-up does a cfg3 training step reach its steady time?  Times consecutive groups of G steps
(sync between groups) from a cold start -- the question behind 'bench.py --steps 20 --warmup 5 vs --steps 200
--warmup 20' (VERDICT r2 weak #8).   python tools/step_ramp.py [G] [groups] [idle_ms_before]"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 40
IDLE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
PRESPIN = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0     # ms of unrelated GPU work (bf16 GEMMs) right before the steps
PRIME = int(sys.argv[5]) if len(sys.argv) > 5 else 0             # 1: one forward + backward (gradients zeroed again) first
dev = torch.device("cuda:0")
off, items, rew = synth_log_device(1_000_000, 100_000, seed=12345, device=dev)
core = CQLCore(100_000, CQLHyper(d=128, window=50, batch=4096, seed=0), device=dev)
core.set_log(off, items, rew)
torch.cuda.synchronize()
if PRIME:
    t0 = time.perf_counter()
    core.forward_backward()
    core.grads.zero_()
    torch.cuda.synchronize()
    print(json.dumps({"prime_ms": 1e3 * (time.perf_counter() - t0)}))
if IDLE:
    time.sleep(IDLE / 1e3)
if PRESPIN:
    a_ = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < PRESPIN:
        for _ in range(10):
            a_ @ a_
        torch.cuda.synchronize()
out = []
for g in range(NG):
    t0 = time.perf_counter()
    core.train_steps(G)
    torch.cuda.synchronize()
    out.append(round(1e3 * (time.perf_counter() - t0) / G, 4))
print(json.dumps({"group": G, "idle_ms": IDLE, "prespin_ms": PRESPIN, "prime": PRIME, "ms_per_step_by_group": out}))

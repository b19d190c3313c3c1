#!/usr/bin/env python3
"""Order-of-events probe (r3): tokens executed in order, then 150 timed cfg3 training steps.
   tokens: tiny (one small kernel on the default stream) | work (torch log generation, 1 M users) | core (CQLCore: creates
   the library's streams) | tstream (a torch.cuda.Stream(), used once) | sync"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
core = log = None
for tok in sys.argv[1:]:
    if tok == "tiny":
        torch.zeros(64, device=dev).add_(1)
        torch.cuda.synchronize()
    elif tok == "work":
        log = synth_log_device(1_000_000, 100_000, seed=12345, device=dev)
    elif tok == "core":
        core = CQLCore(100_000, CQLHyper(d=128, window=50, batch=4096, seed=0), device=dev)
    elif tok == "tstream":
        s_ = torch.cuda.Stream()
        with torch.cuda.stream(s_):
            torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
    elif tok == "sync":
        torch.cuda.synchronize()
core.set_log(*log)
core.train_steps(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    core.train_steps(50)
torch.cuda.synchronize()
print(json.dumps({"order": sys.argv[1:], "ms_per_step": round(1e3 * (time.perf_counter() - t0) / 150, 4)}))

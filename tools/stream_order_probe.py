#!/usr/bin/env python3
"""Does a predict pass BEFORE the first training step slow training down?  (r3: bench.py with the predict warm-up in front
of the training leg ran 1.34 instead of 0.69 ms/step -- the step's internal streams lost their concurrency.)
    python tools/stream_order_probe.py MODE     MODE = base | predict_first | train1_predict | torchstream_first | torchstream_after_core"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda:0")
if mode == "torchstream_first":        # a torch stream pool BEFORE the library's streams exist (the caller's mistake)
    s0_ = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s0_):
        torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize()
NU = 200_000
off, items, rew = synth_log_device(1_000_000, 100_000, seed=12345, device=dev, user_lo=0, user_hi=NU)
core = CQLCore(100_000, CQLHyper(d=128, window=50, batch=4096, seed=0), device=dev)
core.set_log(off, items, rew)
users = torch.arange(NU, dtype=torch.int32, device=dev)
rows = torch.repeat_interleave(torch.arange(NU, device=dev), off[1:] - off[:-1])
seen = items[torch.argsort(rows * 100_000 + items.to(torch.int64))].contiguous()
torch.cuda.synchronize()
if mode == "train1_predict":
    core.train_steps(1)
    torch.cuda.synchronize()
if mode in ("torchstream_first", "torchstream_after_core"):
    s_ = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s_):
        torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize()
if mode in ("predict_first", "train1_predict"):
    core.encode_topk(off, items, users, 10, seen=(off, seen), chunk=65536)
    torch.cuda.synchronize()
core.train_steps(40)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    core.train_steps(50)
torch.cuda.synchronize()
print(json.dumps({"mode": mode, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / 150, 4)}))

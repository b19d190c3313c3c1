#!/usr/bin/env python3
"""Train a few steps and print a digest of everything the step leaves behind (losses as bit patterns, SHA-256 of theta,
Adam moments and target net) -- for bit-for-bit A/B runs of build or environment knobs in separate processes.

    python tools/train_digest.py --d 128 --items 20000 --users 3000 --steps 6 [--phased]
"""
import argparse
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--items", type=int, default=20_000)
    ap.add_argument("--users", type=int, default=3_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--phased", action="store_true", help="the data-parallel step loop (one rank, no collective)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    off, items, rew = synth_log_device(a.users, a.items, seed=12345, device=dev)
    core = CQLCore(a.items, CQLHyper(d=a.d, window=20, batch=a.batch, seed=3), device=dev)
    core.set_log(off, items, rew)
    losses = torch.zeros(a.steps, device=dev)
    core.train_steps(a.steps, losses, phased=True if a.phased else None)
    one = torch.zeros(1, device=dev)
    core.train_step(one)              # the single-step entry point as well
    torch.cuda.synchronize()
    lv = np.concatenate([losses.float().cpu().numpy().ravel(), one.cpu().numpy().ravel()]).astype(np.float32)
    h = hashlib.sha256()
    for t in (core.theta, core.adam_m, core.adam_v, core.target, core.theta_b, core.target_b):
        h.update(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    print(json.dumps({"losses": [int(x) for x in lv.view(np.uint32)], "state_sha256": h.hexdigest()}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Predict pass (encode + seen filter + top-K) of a briefly trained model over the first USERS users of the cfg3 log and a
SHA-256 of everything it returns -- for bit-for-bit A/B runs of build or environment knobs in separate processes.

    python tools/topk_digest.py [--users 150205] [--items 100000] [--train 30] [--chunk 131072]
"""
import argparse
import hashlib
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd.core import CQLCore, CQLHyper  # noqa: E402
from replay_cql_amd.data import synth_log_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2 * 65_536 + 19_133)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--train", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, NN = a.users, a.items
    off, items, rew = synth_log_device(1_000_000, NN, seed=12345, device=dev, user_lo=0, user_hi=n)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), off[1:] - off[:-1])
    seen = items[torch.argsort(rows * NN + items.to(torch.int64))].contiguous()
    core = CQLCore(NN, CQLHyper(d=128, window=50, batch=4096, seed=0), device=dev)
    core.set_log(off, items, rew)
    core.train(a.train)
    users = torch.arange(n, dtype=torch.int32, device=dev)
    idx, val, cnt = core.encode_topk(off, items, users, a.k, seen=(off, seen), chunk=a.chunk)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (idx, val, cnt):
        h.update(t.cpu().contiguous().view(torch.uint8).numpy().tobytes())
    print(json.dumps({"sha256": h.hexdigest(), "users": n, "cnt_min": int(cnt.min()), "idx0": idx[0].tolist()}))


if __name__ == "__main__":
    main()

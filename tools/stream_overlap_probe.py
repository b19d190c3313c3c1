#!/usr/bin/env python3
"""Do the library's aux streams run concurrently with torch's default stream?  Order of events as argv:
   tokens: init (cqlrec_runtime_init) | work (200 ms of torch ops on the default stream) | alloc (1 GiB torch.empty + free)
   e.g.  python tools/stream_overlap_probe.py init work      vs      work init"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from replay_cql_amd import _native as N  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for tok in sys.argv[1:]:
    if tok == "init":
        N.runtime_init()
    elif tok == "work":
        a = torch.arange(50_000_000, device=dev, dtype=torch.int64)
        for _ in range(20):
            a = (a * 3 + 1) % 1000003
        torch.cuda.synchronize()
        del a
    elif tok == "alloc":
        x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        del x
    elif tok == "sort":
        a = torch.randint(0, 1 << 30, (20_000_000,), device=dev)
        torch.sort(a)
        torch.cuda.synchronize()
    elif tok == "tstream":
        s_ = torch.cuda.Stream()
        with torch.cuda.stream(s_):
            torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
N.runtime_init()
s0, s1 = N.aux_stream(0), N.aux_stream(1)
CY = 2_000_000


def timed(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in streams:
        if st is None:
            torch.cuda._sleep(CY)
        else:
            with torch.cuda.stream(st):
                torch.cuda._sleep(CY)
    torch.cuda.synchronize()
    return round(1e6 * (time.perf_counter() - t0))


timed([None, s0, s1])
print(json.dumps({"order": sys.argv[1:], "one": timed([None]), "null+aux0": timed([None, s0]), "aux0+aux1": timed([s0, s1]),
                  "all3": timed([None, s0, s1])}))

"""Top-K pass timing: python tools/topk_bench.py [--users 65536] [--items 100000] [--d 128] [--train 0|35] [--k 10]
Run once per CQL_TOPK_FUSED setting (the knob is read once per process)."""
import argparse, os, sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N
from replay_cql_amd.core import CQLCore, CQLHyper
from replay_cql_amd.data import synth_log_device
ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=65536)
ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--train", type=int, default=0)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
U, NI = a.users, a.items
off, items, rew = synth_log_device(U, NI, device=dev)
core = CQLCore(NI, CQLHyper(d=a.d, window=50, batch=4096), device=dev)
core.set_log(off, items, rew)
if a.train:
    core.train(a.train)
torch.cuda.synchronize()
users = torch.arange(U, dtype=torch.int32, device=dev)
rows = torch.repeat_interleave(torch.arange(U, device=dev), off[1:] - off[:-1])
seen_items = items[torch.argsort(rows * NI + items.to(torch.int64))].contiguous()
hb = core.encode(off, items, users)
lib = N.load()
def t(fn, n=a.reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def phases(fn):
    N.check(lib.cqlrec_prof_enable(1), "prof"); fn(); torch.cuda.synchronize()
    ph = N.prof_read(); N.check(lib.cqlrec_prof_enable(0), "prof")
    return {k: (round(v[0], 4), v[1]) for k, v in ph.items() if v[1]}
tag = "fused=%s" % os.environ.get("CQL_TOPK_FUSED", "1")
for name, fn in (("no seen", lambda: core.score_topk(hb, a.k, chunk=U)),
                 ("seen", lambda: core.score_topk(hb, a.k, seen=(off, seen_items), chunk=U)),
                 ("encode+seen (pipelined)", lambda: core.encode_topk(off, items, users, a.k, seen=(off, seen_items),
                                                                      chunk=U))):
    ms = t(fn)
    print(f"{tag} U={U} N={NI} d={a.d} k={a.k} train={a.train} {name}: {ms:.3f} ms  {U / ms * 1e-3:.2f} M users/s  "
          f"MFMA frac {2.0 * U * NI * a.d / (ms * 1e-3) / 2.5e15:.3f}  phases {phases(fn)}", flush=True)

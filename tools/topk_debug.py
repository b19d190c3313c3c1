import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N
from replay_cql_amd.core import CQLCore, CQLHyper
from replay_cql_amd.data import synth_log_device
dev = torch.device("cuda:0")
U, NI = 100_000, 100_000
off, items, rew = synth_log_device(U, NI, device=dev)
core = CQLCore(NI, CQLHyper(d=128, window=50, batch=4096), device=dev)
core.set_log(off, items, rew)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 35
core.train(steps); torch.cuda.synchronize()
nu = 16384
users = torch.arange(nu, dtype=torch.int32, device=dev)
rows = torch.repeat_interleave(torch.arange(U, device=dev), off[1:] - off[:-1])
seen_items = items[torch.argsort(rows * NI + items.to(torch.int64))].contiguous()
hb = core.encode(off, items, users)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("no seen   ms", t(lambda: core.score_topk(hb, 10)))
print("with seen ms", t(lambda: core.score_topk(hb, 10, seen=(off, seen_items))))
hr = (torch.randn(nu, 128, device=dev) * 0.5).to(torch.bfloat16)
print("random h, no seen ms", t(lambda: core.score_topk(hr, 10)))
print("random h, seen    ms", t(lambda: core.score_topk(hr, 10, seen=(off, seen_items))))
i1, v1, c1 = core.score_topk(hb, 10)
print("top vals user0", v1[0].tolist(), i1[0].tolist())
print("distinct top1 items", i1[:, 0].unique().numel(), "hb std", hb.float().std().item())

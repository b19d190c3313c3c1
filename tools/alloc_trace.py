#!/usr/bin/env python3
"""Which lines of the data-parallel step loop allocate device memory?  Two gloo ranks on one GPU, allocator history of
rank 0 over three steps (python frames of every alloc event).   python tools/alloc_trace.py [sharded]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, shard):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from replay_cql_amd import dist as PD
    from replay_cql_amd.core import CQLCore, CQLHyper
    from replay_cql_amd.data import synth_log_device
    r, w, pg = PD.init_from_env("gloo")
    torch.cuda.set_device(0)
    off, items, rew = synth_log_device(400, 1000, seed=1, device="cuda:0", user_lo=rank * 200, user_hi=(rank + 1) * 200)
    core = CQLCore(1000, CQLHyper(d=128, window=8, batch=128, seed=5), device="cuda:0", rank=rank, world=world,
                   process_group=pg, shard_optimizer=shard)
    core.set_log(off, items, rew)
    scratch = torch.zeros(8, device="cuda:0")
    core.train_steps(2, scratch)
    torch.cuda.synchronize()
    if rank == 0:
        torch.cuda.memory._record_memory_history(max_entries=10000)
    core.train_steps(3, scratch)
    torch.cuda.synchronize()
    if rank == 0:
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
        n = 0
        for tr in snap["device_traces"]:
            for ev in tr:
                if ev["action"] == "alloc":
                    n += 1
                    fr = [f"{Path(f['filename']).name}:{f['line']}:{f['name']}" for f in ev.get("frames", [])
                          if "site-packages" not in f["filename"] and "dist-packages" not in f["filename"]][:4]
                    fr2 = [f"{Path(f['filename']).name}:{f['line']}:{f['name']}" for f in ev.get("frames", [])][:6]
                    print(f"alloc {ev['size']:>10} B  ours={fr}  top={fr2}")
        print("allocs over 3 steps:", n)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, len(sys.argv) > 1 and sys.argv[1] == "sharded"), nprocs=2, join=True)

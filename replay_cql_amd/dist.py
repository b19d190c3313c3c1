"""Data-parallel plumbing (one process per GPU; torch.distributed backend "nccl" = RCCL over xGMI on ROCm, "gloo" on
CPU for tests).  The path shards by user: every rank samples transitions from its own CSR shard, the
only exchange step is the SUM all-reduce of the flat fp32 gradient buffer between the two halves of the step;
Adam then runs identically on every rank, so parameters stay bit-identical across ranks without a broadcast."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch


def shard_range(n_users: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous user_idx range of `rank` (transitions never cross users, so shards are independent)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * n_users // world, (rank + 1) * n_users // world


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as torch.distributed.run sets them)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return rank, world, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    if torch.cuda.is_available():
        # the library's streams before RCCL's (cqlrec_runtime_init, include/cqlrec.h): on the device the caller selected
        from . import _native as N
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            N.runtime_init()
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, dist.group.WORLD


def allreduce_sum_(flat: torch.Tensor, group=None, bucket_elems: int = 0) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient buffer.  bucket_elems > 0 splits it into fixed-size buckets
    (async ops, waited together) so that RCCL can pipeline reduce-scatter / all-gather phases over the xGMI links."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if bucket_elems <= 0 or bucket_elems >= flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[lo: lo + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    return flat


def max_over_ranks(value: float, device, group=None) -> float:
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value: float, device, group=None) -> float:
    """SUM of a host scalar over the ranks (identity when not distributed): epoch length from the global log size,
    validation loss agreed across ranks (CQL.fit_arrays)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    dev = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


def reduce_scatter_sum(out: torch.Tensor, region: torch.Tensor, group=None, async_op: bool = False):
    """out = this rank's 1/W slice of the SUM over ranks of `region` (numel = W * out.numel()).  RCCL: one
    reduce-scatter.  gloo has none: all-reduce the region and keep the own slice (same sums; used by the tests that run
    several ranks on the CPU backend)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return dist.reduce_scatter_tensor(out, region, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    dist.all_reduce(region, op=dist.ReduceOp.SUM, group=group)
    r, n = dist.get_rank(group), out.numel()
    out.copy_(region[r * n: (r + 1) * n])
    return None


def all_gather_into(full: torch.Tensor, mine: torch.Tensor, group=None, async_op: bool = False):
    """full[r*n:(r+1)*n] = rank r's `mine` for every r (n = mine.numel()).  `mine` must not alias `full` (the callers
    pass a persistent staging buffer).  No device allocation on either backend: gloo gathers straight into views."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(full, mine, group=group, async_op=async_op)
    # gloo (tests): the flat form too where this torch's gloo has it (no temporary of the region's size on the device),
    # else the list form into views
    global _GLOO_FLAT_ALLGATHER
    if _GLOO_FLAT_ALLGATHER is not False:
        try:
            dist.all_gather_into_tensor(full, mine, group=group)
            _GLOO_FLAT_ALLGATHER = True
            return None
        except (RuntimeError, NotImplementedError):
            if _GLOO_FLAT_ALLGATHER is True:
                raise
            _GLOO_FLAT_ALLGATHER = False
    w, n = dist.get_world_size(group), mine.numel()
    dist.all_gather([full[r * n: (r + 1) * n] for r in range(w)], mine, group=group)
    return None


_GLOO_FLAT_ALLGATHER = None


def shard_plan(off_E_in: int, off_E_out: int, off_W1: int, total: int, n_items: int, d: int, world: int, rank: int):
    """Element ranges of the flat parameter buffer under the row-sharded optimizer (SURVEY 8(e), cfg5 variant):
    R = n_items // world rows of E_in and of E_out per rank; what does not divide -- the last n_items - world * R rows,
    the PAD row and alignment padding of E_in, b_out, the encoder -- stays replicated ("tails", all-reduced).
    Pure arithmetic (tested on the CPU for world = 8 and n_items mod 8 != 0)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    n = (n_items // world) * d
    return {"n": n,
            "in_region": (off_E_in, off_E_in + world * n), "in_own": (off_E_in + rank * n, off_E_in + (rank + 1) * n),
            "out_region": (off_E_out, off_E_out + world * n), "out_own": (off_E_out + rank * n, off_E_out + (rank + 1) * n),
            "tail_in": (off_E_in + world * n, off_E_out), "tail_out": (off_E_out + world * n, off_W1),
            "tail_enc": (off_W1, total)}

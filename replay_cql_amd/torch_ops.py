"""torch.ops.cqlrec.* -- the PyTorch-ROCm custom ops SURVEY 8(b) lists, registered by libcqlrec_torch.so
(csrc/torch_ops.cpp: a TORCH_LIBRARY shim that allocates outputs / scratch with torch and calls the C ABI of
include/cqlrec.h on the current HIP stream):

    gather_pool_fwd, gather_pool_bwd, qhead_lse_fwd, qhead_argmax_fwd, qhead_lse_bwd, qhead_gather_dot, score_topk,
    fused_adam_ema

`load()` makes them available; there is no fallback implementation -- a missing library raises."""
from __future__ import annotations

from pathlib import Path

OPS = ("gather_pool_fwd", "gather_pool_bwd", "qhead_lse_fwd", "qhead_argmax_fwd", "qhead_lse_bwd", "qhead_gather_dot",
       "score_topk", "fused_adam_ema")
_LIB = Path(__file__).resolve().parent / "libcqlrec_torch.so"
_loaded = False


def load():
    """Register the ops (idempotent) and return the `torch.ops.cqlrec` namespace."""
    global _loaded
    import torch
    from . import _native as N
    if not _loaded:
        N.load()                                    # libcqlrec.so first: torch's HIP runtime, then the kernels
        if not _LIB.exists():
            raise N.CqlrecError(f"{_LIB} is missing: build it with `python -m replay_cql_amd.build`")
        torch.ops.load_library(str(_LIB))
        _loaded = True
    return torch.ops.cqlrec

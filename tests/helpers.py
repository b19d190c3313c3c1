"""Shared test helpers: device buffers for the C ABI, oracle model <-> device state."""
from __future__ import annotations

import numpy as np
import torch

from oracle import cql_oracle as O
from replay_cql_amd import _native as N

DEV = "cuda:0"
_KEEP = []   # device tensors handed to the C ABI as raw pointers must outlive the (asynchronous) call


def keep(t):
    _KEEP.append(t)
    return t


def release_kept():
    _KEEP.clear()


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return keep(t.to(DEV).contiguous())


def bf16_dev(x_f32: np.ndarray) -> torch.Tensor:
    """fp32 numpy (any values) -> device bf16 tensor holding oracle-rounded values."""
    bits = O.bf16_bits(np.asarray(x_f32, dtype=np.float32)).astype(np.int16)
    return keep(torch.as_tensor(bits).to(DEV).view(torch.bfloat16).contiguous())


def bf16_to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


def ws_bytes_tensor(nbytes: int) -> torch.Tensor:
    return torch.empty(int(nbytes), dtype=torch.uint8, device=DEV)


def small_log(U=64, N=257, seed=1, mean_len=12, max_len=40):
    u, i, t, r = O.synth_log(U, N, seed=seed, mean_len=mean_len, max_len=max_len)
    return O.build_csr(u, i, t, r, U)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

"""Arrow-batch ingest / egress of the hot path.

The reference moves data across the model boundary as Spark DataFrames; with `spark.sql.execution.arrow.pyspark.enabled`
(replay/session_handler.py:47) a `toPandas()` (replay/models/neuromf.py:332) or `createDataFrame(pandas)` is a stream of
Arrow record batches that pandas then re-materialises column by column.  Here the batches themselves are the interface:
columns are copied once into pinned host buffers and sent to the GPU with one asynchronous copy per column; results
come back as one pinned device->host copy per column and are wrapped (zero-copy) as a RecordBatch with REC_SCHEMA.

Schemas = replay/constants.py:16-31 (IntegerType -> int32, TimestampType -> timestamp[us], DoubleType -> float64)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence, Union

import numpy as np
import pyarrow as pa
import torch

LOG_SCHEMA = pa.schema([("user_idx", pa.int32()), ("item_idx", pa.int32()), ("timestamp", pa.timestamp("us")),
                        ("relevance", pa.float64())])
REC_SCHEMA = pa.schema([("user_idx", pa.int32()), ("item_idx", pa.int32()), ("relevance", pa.float64())])

Batches = Union[pa.Table, pa.RecordBatch, Iterable[pa.RecordBatch]]

_TORCH_OF = {"user_idx": torch.int32, "item_idx": torch.int32, "timestamp": torch.int64, "relevance": torch.float64}


def _as_batches(data: Batches) -> Sequence[pa.RecordBatch]:
    if isinstance(data, pa.RecordBatch):
        return [data]
    if isinstance(data, pa.Table):
        return data.to_batches()
    out = list(data)
    for b in out:
        if not isinstance(b, pa.RecordBatch):
            raise ValueError(f"expected pyarrow.RecordBatch, got {type(b)}")
    return out


def _column_numpy(col: pa.Array, name: str) -> np.ndarray:
    """One Arrow column as a numpy array of the hot path's dtype (no copy when the Arrow type already matches)."""
    if col.null_count:
        raise ValueError(f"column {name} contains nulls")
    t = col.type
    if name in ("user_idx", "item_idx"):
        if not pa.types.is_integer(t):
            raise ValueError(f"column {name} must be an integer column, got {t}")
        if t != pa.int32():
            col = col.cast(pa.int32())                # safe cast: raises on overflow
        return col.to_numpy(zero_copy_only=False)
    if name == "relevance":
        if not (pa.types.is_floating(t) or pa.types.is_integer(t)):
            raise ValueError(f"column relevance must be numeric, got {t}")
        if t != pa.float64():
            col = col.cast(pa.float64())
        return col.to_numpy(zero_copy_only=False)
    # timestamp: any Arrow timestamp / date unit, integers, or floats -- only the ORDER matters (S2)
    if pa.types.is_timestamp(t) or pa.types.is_date(t) or pa.types.is_time(t) or pa.types.is_duration(t):
        a = col.to_numpy(zero_copy_only=False)
        return a.view(np.int64) if a.dtype.itemsize == 8 else a.astype("datetime64[us]").view(np.int64)
    if pa.types.is_integer(t):
        return (col if t == pa.int64() else col.cast(pa.int64())).to_numpy(zero_copy_only=False)
    if pa.types.is_floating(t):
        from .data import timestamp_key
        return timestamp_key(col.to_numpy(zero_copy_only=False))
    raise ValueError(f"column timestamp has unsupported type {t}")


def columns_to_device(data: Batches, device, columns: Sequence[str] = ("user_idx", "item_idx", "timestamp", "relevance")
                      ) -> Dict[str, torch.Tensor]:
    """RecordBatches -> one device tensor per requested column (pinned staging, asynchronous copies on the current
    stream).  Missing `timestamp` / `relevance` columns are returned as None (a log without them is legal for
    predict: replay/models/base_rec.py:1220-1257 only requires user_idx / item_idx)."""
    batches = _as_batches(data)
    n = sum(b.num_rows for b in batches)
    dev = torch.device(device)
    out: Dict[str, Optional[torch.Tensor]] = {}
    names = set(batches[0].schema.names) if batches else set()
    for name in columns:
        if name not in names:
            if name in ("user_idx", "item_idx"):
                raise ValueError(f"log has no column {name}")
            out[name] = None
            continue
        host = torch.empty(n, dtype=_TORCH_OF[name], pin_memory=(dev.type == "cuda" and n > 0))
        dst = host.numpy()
        lo = 0
        for b in batches:
            a = _column_numpy(b.column(b.schema.get_field_index(name)), name)
            dst[lo: lo + len(a)] = a
            lo += len(a)
        out[name] = host.to(dev, non_blocking=True)
    return out


def ids_to_device(ids, column: str, device) -> Optional[torch.Tensor]:
    """users / items argument of predict: Arrow batches or table with that column, an Arrow array, or an iterable of
    ids -> sorted unique int64 device tensor (the role of _get_ids, replay/models/base_rec.py:542-558)."""
    if ids is None:
        return None
    if isinstance(ids, (pa.Table, pa.RecordBatch)):
        arr = ids.column(column)
        arr = arr.combine_chunks() if isinstance(arr, pa.ChunkedArray) else arr
        a = arr.to_numpy(zero_copy_only=False)
    elif isinstance(ids, (pa.Array, pa.ChunkedArray)):
        a = (ids.combine_chunks() if isinstance(ids, pa.ChunkedArray) else ids).to_numpy(zero_copy_only=False)
    elif torch.is_tensor(ids):
        return torch.unique(ids.to(device=device, dtype=torch.int64))
    else:
        a = np.asarray(list(ids) if not isinstance(ids, np.ndarray) else ids)
    return torch.unique(torch.as_tensor(np.ascontiguousarray(a).astype(np.int64)).to(device))


def recs_to_arrow(users: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, cnt: torch.Tensor) -> pa.RecordBatch:
    """Device top-k block ([n] user ids, [n x k] item ids / scores, [n] valid counts) -> RecordBatch with REC_SCHEMA:
    compacted on the device, one pinned D2H copy per column, wrapped without a further copy."""
    n, k = idx.shape if idx.dim() == 2 else (0, 0)
    if n == 0 or k == 0:
        return pa.RecordBatch.from_arrays([pa.array([], pa.int32()), pa.array([], pa.int32()),
                                           pa.array([], pa.float64())], schema=REC_SCHEMA)
    keep = torch.arange(k, device=idx.device)[None, :] < cnt[:, None]
    cols = (users.to(torch.int32)[:, None].expand(n, k)[keep], idx[keep], val[keep].to(torch.float64))
    host = []
    for c in cols:
        h = torch.empty(c.shape, dtype=c.dtype, pin_memory=c.is_cuda)
        h.copy_(c, non_blocking=True)
        host.append(h)
    if idx.is_cuda:
        torch.cuda.current_stream(idx.device).synchronize()          # the only host wait of the egress
    return pa.RecordBatch.from_arrays([pa.array(h.numpy()) for h in host], schema=REC_SCHEMA)

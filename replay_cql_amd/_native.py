"""ctypes binding of libcqlrec.so (the C ABI in include/cqlrec.h).

There is NO fallback: if the HIP library is missing or an entry point is absent, loading raises.  The product path
never routes through oracle/ or any CPU implementation."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step driver runs four
# streams concurrently (the caller's + three internal), the predict pass and the data-parallel loop one more each: ask
# for eight -- effective when this module is imported before the first HIP call of the process (importing torch makes
# none); an explicit setting of the user's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# CQLREC_LIB: another build of the same library (A/B-ing kernel variants with tools/); default: the in-tree build
_LIB_PATH = Path(os.environ["CQLREC_LIB"]).resolve() if os.environ.get("CQLREC_LIB") else \
    Path(__file__).resolve().parent / "libcqlrec.so"
_lib: Optional[C.CDLL] = None

ABI_VERSION = 2
TOPK_ALL, TOPK_SEEN, TOPK_SCORE, TOPK_SEEN_BESIDE = 0, 1, 2, 3
QHEAD_LSE = 1
QHEAD_ARGMAX = 2

vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double


class Layout(C.Structure):
    _fields_ = [
        ("n_items", i64), ("d", i32), ("reserved", i32),
        ("off_E_in", i64), ("off_E_out", i64), ("off_b_out", i64), ("off_W1", i64), ("off_b1", i64),
        ("off_W2", i64), ("off_b2", i64), ("total", i64),
    ]

    SEGMENTS = ("E_in", "E_out", "b_out", "W1", "b1", "W2", "b2")

    def offset(self, name: str) -> int:
        return int(getattr(self, "off_" + name))

    def shape(self, name: str):
        n, d = int(self.n_items), int(self.d)
        return {"E_in": (n + 1, d), "E_out": (n, d), "b_out": (n,), "W1": (d, d), "b1": (d,), "W2": (d, d),
                "b2": (d,)}[name]


class TrainCtx(C.Structure):
    _fields_ = [
        ("layout", Layout),
        ("offsets", vp), ("items", vp), ("rewards", vp), ("n_users", i64),
        ("theta", vp), ("grads", vp), ("adam_m", vp), ("adam_v", vp), ("target", vp), ("theta_b", vp),
        ("target_b", vp),
        ("batch", i32), ("window", i32), ("world", i32), ("rank", i32),
        ("gamma", f64), ("alpha", f64), ("lr", f64), ("beta1", f64), ("beta2", f64), ("eps", f64), ("tau", f64),
        ("seed", u64),
        ("ws", vp), ("ws_bytes", i64),
    ]


class TrainViews(C.Structure):
    _fields_ = [(n, vp) for n in (
        "users", "tpos", "act", "a_star", "rew", "done", "q_a", "lse", "q_targ", "y", "coef", "dH", "dh0", "h0_s",
        "hb_s", "hb_sn", "hb_tn")]


# name -> (restype, argtypes); every symbol include/cqlrec.h declares
SIGNATURES = {
    "cqlrec_abi_version": (i32, []),
    "cqlrec_last_error": (C.c_char_p, []),
    "cqlrec_layout_make": (i32, [i64, i32, C.POINTER(Layout)]),
    "cqlrec_sample_transitions": (i32, [vp, vp, vp, i64, u64, u64, u64, i32, vp, vp, vp, vp, vp, vp]),
    "cqlrec_gather_pool_fwd": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, vp]),
    "cqlrec_gather_pool_bwd": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp]),
    "cqlrec_gather_pool_bwd_ws_bytes": (i64, [i64, i32, i32]),
    "cqlrec_gather_pool_bwd_prepare": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, i64, vp, i64, vp]),
    "cqlrec_gather_pool_bwd_apply": (i32, [vp, i64, i32, i32, i64, vp, i64, vp, vp]),
    "cqlrec_gather_pool_bwd_sorted": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i64, vp, i64, vp, vp]),
    "cqlrec_linear_bf16": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "cqlrec_encoder_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "cqlrec_encoder_bwd_ws_bytes": (i64, [i64, i32]),
    "cqlrec_encoder_bwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, vp]),
    "cqlrec_qhead_ws_bytes": (i64, [i64, i64, i32]),
    "cqlrec_qhead_fwd": (i32, [vp, i64, vp, vp, i64, i32, i32, vp, i64, vp, vp, vp, vp]),
    "cqlrec_gather_dot": (i32, [vp, vp, vp, vp, i64, i32, vp, vp]),
    "cqlrec_td_loss": (i32, [vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp]),
    "cqlrec_qhead_bwd_ws_bytes": (i64, [i64, i64, i32]),
    "cqlrec_qhead_bwd": (i32, [vp, vp, vp, vp, i64, vp, vp, i64, i32, f32, vp, i64, vp, vp, vp, vp]),
    "cqlrec_qhead_bwd_items": (i32, [vp, vp, vp, vp, i64, vp, vp, i64, i32, f32, vp, i64, vp, vp, vp]),
    "cqlrec_qhead_bwd_states": (i32, [vp, vp, vp, vp, i64, vp, vp, i64, i32, f32, vp, i64, vp, vp]),
    "cqlrec_adam_ema": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, i32, vp]),
    "cqlrec_cast_bf16": (i32, [vp, vp, i64, vp]),
    "cqlrec_topk_ws_bytes": (i64, [i64, i64, i32, i32]),
    "cqlrec_score_topk": (i32, [vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, i64, vp, vp, vp, vp]),
    "cqlrec_topk_seen_form": (i32, [vp, i64, i64, i32, i32, C.POINTER(C.c_int32), vp]),
    "cqlrec_score_topk_phase": (i32, [vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, i64, vp, vp, vp, i32, vp]),
    "cqlrec_train_ws_bytes": (i64, [i32, i64, i32, i32]),
    "cqlrec_train_step_fwd_bwd": (i32, [C.POINTER(TrainCtx), u64, vp, vp]),
    "cqlrec_train_step_update": (i32, [C.POINTER(TrainCtx), u64, vp]),
    "cqlrec_train_step_forward": (i32, [C.POINTER(TrainCtx), u64, vp, vp]),
    "cqlrec_train_step_forward_after": (i32, [C.POINTER(TrainCtx), u64, vp, vp, vp]),
    "cqlrec_train_step_forward_early_items": (i32, [C.POINTER(TrainCtx), u64, vp, vp, vp, vp]),
    "cqlrec_train_step_backward_items": (i32, [C.POINTER(TrainCtx), u64, vp]),
    "cqlrec_train_step_backward_rest": (i32, [C.POINTER(TrainCtx), u64, vp]),
    "cqlrec_train_step_update_range": (i32, [C.POINTER(TrainCtx), u64, i64, i64, vp]),
    "cqlrec_set_concurrency": (i32, [i32]),
    "cqlrec_runtime_init": (i32, []),
    "cqlrec_runtime_probe_count": (i32, []),
    "cqlrec_aux_stream": (vp, [i32]),
    "cqlrec_qhead_fused_ws_bytes": (i64, [i64, i64, i32]),
    "cqlrec_qhead_fwd_lse_dh": (i32, [vp, i64, vp, vp, i64, i32, vp, i64, vp, vp, vp]),
    "cqlrec_qhead_dh_finish": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, C.c_float, vp, vp]),
    "cqlrec_train_steps": (i32, [C.POINTER(TrainCtx), u64, i32, vp, vp]),
    "cqlrec_train_views_get": (i32, [C.POINTER(TrainCtx), u64, C.POINTER(TrainViews)]),
    "cqlrec_build_csr_ws_bytes": (i64, [i64]),
    "cqlrec_build_csr": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp]),
    "cqlrec_eval_topk_ws_bytes": (i64, [i64, i32]),
    "cqlrec_eval_topk": (i32, [vp, i64, i32, vp, vp, vp, C.POINTER(i32), i32, vp, i64, vp, vp, vp]),
    "cqlrec_prof_enable": (i32, [i32]),
    "cqlrec_prof_select": (i32, [C.c_uint32]),
    "cqlrec_debug_marks_enable": (i32, [i32]),
    "cqlrec_debug_marks_read": (i32, [C.POINTER(C.c_float)]),
    "cqlrec_prof_read": (i32, [C.POINTER(C.c_double), C.POINTER(i64)]),
}

PHASES = ("sample", "gather_fwd", "encoder_fwd", "qhead_lse", "qhead_argmax", "qhead_bwd_dh", "qhead_bwd_de",
          "qhead_small", "encoder_bwd", "gather_bwd", "adam", "topk_tilemax", "topk_select")


def prof_read():
    """{phase: (summed kernel ms, launches)} since the last read (synchronises the recorded events)."""
    ms = (C.c_double * len(PHASES))()
    cnt = (i64 * len(PHASES))()
    check(load().cqlrec_prof_read(ms, cnt), "prof_read")
    return {p: (float(ms[i]), int(cnt[i])) for i, p in enumerate(PHASES)}


class CqlrecError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library or raise -- never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be loaded first: it bundles its own libamdhip64, and libcqlrec.so has to bind to that same HIP
    # runtime instance (one runtime per process -- torch's stream handles and device pointers are only valid there).
    import torch  # noqa: F401  pylint: disable=import-outside-toplevel,unused-import
    if not _LIB_PATH.exists():
        raise CqlrecError(
            f"{_LIB_PATH} is missing: build it with `python -m replay_cql_amd.build` (hipcc --offload-arch=gfx950). "
            "replay_cql_amd has no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # pragma: no cover
            raise CqlrecError(f"{_LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype, fn.argtypes = res, args
    if lib.cqlrec_abi_version() != ABI_VERSION:
        raise CqlrecError(f"libcqlrec ABI {lib.cqlrec_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


AUX_STREAMS = 2
_aux_cache: dict = {}


def runtime_init() -> None:
    """Create the library's streams for the CURRENT device now (cqlrec_runtime_init: why the order of stream creation in
    a process decides whether the training step keeps its concurrency).  CQLCore calls this on construction; a
    data-parallel job calls it right after torch.cuda.set_device and BEFORE torch.distributed.init_process_group, whose
    RCCL streams come out of torch's 64-stream pool."""
    check(load().cqlrec_runtime_init(), "runtime_init")


def aux_stream(index: int):
    """Library-owned side stream `index` of the current device as a torch stream object (torch.cuda.ExternalStream):
    side work of the host driver goes there instead of onto a torch.cuda.Stream(), whose first construction creates a
    pool of 64 HIP streams."""
    import torch  # pylint: disable=import-outside-toplevel
    runtime_init()
    dev = torch.cuda.current_device()
    key = (dev, int(index))
    if key not in _aux_cache:
        ptr = load().cqlrec_aux_stream(int(index))
        if not ptr:
            raise CqlrecError(f"no aux stream {index}")
        _aux_cache[key] = torch.cuda.ExternalStream(ptr, device=torch.device("cuda", dev))
    return _aux_cache[key]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().cqlrec_last_error().decode("utf-8", "replace")
        raise CqlrecError(f"{what or 'cqlrec'} failed (rc={rc}): {msg}")


def make_layout(n_items: int, d: int) -> Layout:
    lay = Layout()
    check(load().cqlrec_layout_make(int(n_items), int(d), C.byref(lay)), "layout_make")
    return lay

"""The C-ABI library loads on a CPU-only box and exports every symbol include/cqlrec.h declares; host-only entry points
(layout, workspace sizes, argument validation) behave.  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import pytest

from oracle import cql_oracle as O
from replay_cql_amd import _native as N
from replay_cql_amd import build as B

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    B.build(verbose=False)
    return N.load()


def _declared():
    text = (ROOT / "include" / "cqlrec.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cqlrec_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared()
    assert len(declared) >= 20
    nm = subprocess.run(["nm", "-D", "--defined-only", str(N.lib_path())], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (cqlrec_[a-z0-9_]+)", nm))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(N.SIGNATURES), sorted(set(declared) ^ set(N.SIGNATURES))
    assert lib.cqlrec_abi_version() == N.ABI_VERSION


def test_library_holds_gfx950_code_only():
    """the offload bundle of the .so carries exactly one device target: gfx950 (no multi-arch / compat builds)"""
    s = subprocess.run(["strings", str(N.lib_path())], capture_output=True, text=True).stdout
    targets = set(re.findall(r"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", s))
    assert targets == {"gfx950"}, targets


@pytest.mark.parametrize("n,d", [(4, 64), (257, 64), (10007, 128), (100000, 128), (1000000, 256)])
def test_layout_matches_oracle(lib, n, d):
    lay = N.make_layout(n, d)
    ref = O.Layout.make(n, d)
    assert int(lay.total) == ref.total
    for nm in N.Layout.SEGMENTS:
        assert lay.offset(nm) == ref.off[nm] and lay.shape(nm) == ref.shape(nm)


def test_workspace_queries_are_host_only(lib):
    assert lib.cqlrec_qhead_ws_bytes(8192, 100000, 128) > 0
    assert lib.cqlrec_qhead_bwd_ws_bytes(4096, 100000, 128) >= 100000 * 128 * 4
    assert lib.cqlrec_encoder_bwd_ws_bytes(4096, 128) > 4096 * 128 * 4
    assert lib.cqlrec_topk_ws_bytes(1000, 100000, 128, 10) >= 3125 * 1000 * 4
    a = lib.cqlrec_train_ws_bytes(4096, 100000, 128, 50)
    assert a > lib.cqlrec_qhead_bwd_ws_bytes(4096, 100000, 128)
    assert lib.cqlrec_train_ws_bytes(8192, 100000, 128, 50) > a
    assert lib.cqlrec_gather_pool_bwd_ws_bytes(4096, 50, 128) > 4096 * 50 * 16


def test_argument_validation_without_gpu(lib):
    """validation happens on the host before any HIP call"""
    with pytest.raises(N.CqlrecError, match="unsupported"):
        N.make_layout(10, 100)
    with pytest.raises(N.CqlrecError, match="out of range"):
        N.make_layout(0, 64)
    with pytest.raises(N.CqlrecError, match="NULL"):
        N.check(lib.cqlrec_adam_ema(None, None, None, None, None, None, None, 64, 1e-3, 1.0, 0.9, 0.999, 1e-8, 0.005, 1, None))
    with pytest.raises(N.CqlrecError, match="multiple of 4"):
        buf = (C.c_float * 8)()
        p = C.addressof(buf)
        N.check(lib.cqlrec_adam_ema(p, p, p, p, p, p, p, 6, 1e-3, 1.0, 0.9, 0.999, 1e-8, 0.005, 1, None))
    with pytest.raises(N.CqlrecError, match="ctx is NULL"):
        N.check(lib.cqlrec_train_step_update(None, 0, None))
    ctx = N.TrainCtx()
    with pytest.raises(N.CqlrecError, match="CSR pointers"):
        N.check(lib.cqlrec_train_step_fwd_bwd(C.byref(ctx), 0, None, None))


def test_core_refuses_to_run_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from replay_cql_amd.core import CQLCore
    with pytest.raises(N.CqlrecError, match="no CPU path"):
        CQLCore(100)


def test_product_path_never_imports_oracle():
    for f in (ROOT / "replay_cql_amd").glob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_host_layer_under_address_sanitizer():
    """SURVEY 5 / VERDICT r2 missing #6: the host C++ side of the library (argument checks, error plumbing, size, split and
    layout arithmetic of every entry point) built with -fsanitize=address and driven without a GPU
    (replay_cql_amd/build.py::build_asan, tests/asan/host_driver.cpp: host-only objects, no device code -- GPU ASan is
    not available on the pool).  Pass = exit code 0, the driver's "ok", no AddressSanitizer report."""
    import subprocess
    from replay_cql_amd import build as B
    exe = B.build_asan()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:verbosity=1", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert "AddressSanitizer Init done" in r.stderr, "the driver is not an ASan build"
    assert r.returncode == 0 and "host_driver: ok" in r.stdout, r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "ERROR: LeakSanitizer" not in r.stderr, r.stderr[-3000:]

"""Build recipe for libcqlrec.so (HIP, gfx950 only).  `python -m replay_cql_amd.build` or __graft_entry__.build().

The library is built IN-TREE (replay_cql_amd/libcqlrec.so) so that it travels with the repo snapshot to the GPU box;
it is git-ignored.  hipcc cross-compiles for gfx950 without a GPU present."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libcqlrec.so"
TORCH_LIB = PKG / "libcqlrec_torch.so"      # torch.ops.cqlrec.* registration shim over the C ABI (csrc/torch_ops.cpp)
SOURCES = ["misc.hip", "qhead.hip", "qhead_de.hip", "qhead_de2.hip", "qhead_fwd2.hip", "qhead_topk2.hip", "topk.hip", "gbwd.hip", "prep.hip", "train.hip"]
# misc.hip holds the Adam kernel whose expression order is normative: no fma contraction anywhere in that file
EXTRA = {
    "misc.hip": ["-ffp-contract=off"],
    # MFMA results straight into VGPRs (gfx950 has a unified file): no v_accvgpr_read per accumulator register
    "qhead.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_de.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_de2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_topk2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_fwd2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale() -> bool:
    if not LIB.exists() or not TORCH_LIB.exists():
        return True
    t = min(LIB.stat().st_mtime, TORCH_LIB.stat().st_mtime)
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "cqlrec.h", Path(__file__)]
    return any(p.stat().st_mtime > t for p in deps)


def build_torch_ops(verbose: bool = True) -> Path:
    """libcqlrec_torch.so: TORCH_LIBRARY(cqlrec) over the C ABI.  Host C++ only (no kernels): g++ against torch's headers,
    linked to libcqlrec.so (rpath $ORIGIN) and to the torch libraries of THIS interpreter."""
    import torch
    from torch.utils import cpp_extension as ce
    tlib = Path(torch.__file__).resolve().parent / "lib"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include"]
    cmd += [str(CSRC / "torch_ops.cpp"), "-o", str(TORCH_LIB), f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
            f"-L{PKG}", "-lcqlrec", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return TORCH_LIB


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = common + EXTRA.get(src, []) + ["-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    build_torch_ops(verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

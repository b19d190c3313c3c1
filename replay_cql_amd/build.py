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
SOURCES = ["misc.hip", "qhead.hip", "qhead_de.hip", "qhead_de2.hip", "qhead_de3.hip", "qhead_argmax2.hip", "qhead_fwd2.hip", "qhead_fwd3.hip", "qhead_topk2.hip", "qhead_topk4.hip", "topk.hip", "gbwd.hip", "prep.hip", "train.hip"]
# misc.hip holds the Adam kernel whose expression order is normative: no fma contraction anywhere in that file
EXTRA = {
    "misc.hip": ["-ffp-contract=off"],
    # MFMA results straight into VGPRs (gfx950 has a unified file): no v_accvgpr_read per accumulator register
    "qhead.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_de.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_de2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_de3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_argmax2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_topk2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_topk4.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_fwd2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "qhead_fwd3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
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


def build_asan(verbose: bool = False) -> Path:
    """AddressSanitizer build of the HOST side of the library + its driver (tests/asan/host_driver.cpp):
    `hipcc --cuda-host-only -fsanitize=address` -- host code only (argument checks, error plumbing, size / split
    arithmetic, launch preparation), no device code at all (GPU ASan / xnack+ code objects are not available on the
    pool; a kernel launch from this build would fail, and the driver makes none).  Returns the driver executable.
    A CPU test (tests/test_abi.py::test_host_layer_under_address_sanitizer) runs it."""
    hipcc = _hipcc()
    out = PKG / "build" / "asan"
    out.mkdir(parents=True, exist_ok=True)
    lib, exe = out / "libcqlrec_asan.so", out / "host_driver"
    driver = PKG.parent / "tests" / "asan" / "host_driver.cpp"
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "cqlrec.h", driver, Path(__file__)]
    if lib.exists() and exe.exists() and all(p.stat().st_mtime <= min(lib.stat().st_mtime, exe.stat().st_mtime) for p in deps):
        return exe
    flags = [f"--offload-arch={ARCH}", "--cuda-host-only", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address",
             "-fPIC", "-std=c++17", "-Wno-unused-function"]

    def compile_one(src: str) -> Path:
        obj = out / (src + ".o")
        cmd = [hipcc] + flags + ["-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, stderr=None if verbose else subprocess.DEVNULL)
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # a host-only object still refers to its translation unit's device code object (__hip_fatbin_<hash>, registered with
    # the runtime when the library is loaded, read at the first launch): zero-filled stand-ins of our OWN device code,
    # which this build never launches
    undef = subprocess.run(["nm", "-u"] + [str(o) for o in objs], check=True, capture_output=True, text=True).stdout
    syms = sorted({ln.split()[-1] for ln in undef.splitlines() if "__hip_fatbin_" in ln})
    stubs = out / "fatbin_stubs.cpp"
    stubs.write_text("// generated by replay_cql_amd/build.py::build_asan\n" + "".join(
        f'extern "C" {{ __attribute__((aligned(4096))) extern const char {s}[4096]; const char {s}[4096] = {{0}}; }}\n'
        for s in syms))
    clangxx = str(Path(hipcc).resolve().parent.parent / "lib" / "llvm" / "bin" / "clang++")
    if not Path(clangxx).exists():
        clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    rocm_lib = "/opt/rocm/lib"
    subprocess.run([clangxx, "-shared", "-fPIC", "-O1", "-g", "-fsanitize=address", "-std=c++17", "-o", str(lib)] +
                   [str(o) for o in objs] + [str(stubs), f"-L{rocm_lib}", "-lamdhip64", f"-Wl,-rpath,{rocm_lib}"], check=True)
    subprocess.run([clangxx, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-std=c++17", str(driver), "-o",
                    str(exe), f"-L{out}", "-lcqlrec_asan", "-Wl,-rpath,$ORIGIN", f"-L{rocm_lib}", "-lamdhip64",
                    f"-Wl,-rpath,{rocm_lib}", "-lpthread"], check=True)
    return exe


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    else:
        print(build(force="--force" in sys.argv))

"""Build libparsy_amd.so (host C++ + HIP kernels for gfx950) in-tree with hipcc.

`python -m parsy_bench_amd.build` or `parsy_bench_amd.build.build_native()`.
The library lands next to this file so it travels with the source tree (the GPU
box receives in-tree .so files; a JIT cache would not).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libparsy_amd.so"
OBJ = PKG / "build"

HOST_SOURCES = ["inspector.cpp", "gen.cpp", "ordering.cpp", "capi_host.cpp", "schedule.cpp", "dist.cpp", "capi_dist.cpp"]
HIP_SOURCES = ["executor.hip", "chol_kernels.hip", "trsv_kernels.hip", "trsv_sub_kernels.hip", "capi_exec.hip", "mg.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP executor cannot be built (no CPU fallback exists)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_native(force: bool = False, verbose: bool = False, extra_flags=(), out: Path | None = None,
                 objdir: Path | None = None) -> Path:
    """extra_flags/out/objdir exist for diagnostic builds (tools/); the product is the default."""
    global OBJ, LIB
    hipcc = _hipcc()
    saved = (OBJ, LIB)
    if objdir is not None:
        OBJ = Path(objdir)
    if out is not None:
        LIB = Path(out)
    try:
        return _build(hipcc, force, verbose, list(extra_flags))
    finally:
        OBJ, LIB = saved


def _build(hipcc, force, verbose, extra_flags) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "parsy_amd.h"]
    common = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-I", str(PKG.parent / "include"),
              *extra_flags]
    jobs = []
    for src in HOST_SOURCES + HIP_SOURCES:
        sp = CSRC / src
        if not sp.exists():
            continue
        obj = OBJ / (src + ".o")
        if src.endswith(".hip"):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-munsafe-fp-atomics", *common, "-c", str(sp), "-o", str(obj)]
        else:
            cmd = [hipcc, "-x", "c++", *common, "-c", str(sp), "-o", str(obj)]
        if force or _stale(obj, [sp, *headers]):
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = sorted(str(p) for p in OBJ.glob("*.o"))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *objs])
    if LIB.name == "libparsy_amd.so":
        build_drivers(hipcc, run, force)
    return LIB


DRIVERS = PKG / "drivers"


def build_drivers(hipcc=None, run=None, force: bool = False):
    """choleskyTest / choleskyTest03 / triangularTest: the reference's example drivers over the C ABI;
    makingLowerHalf."""
    hipcc = hipcc or _hipcc()
    if run is None:
        def run(cmd):
            subprocess.run(cmd, check=True)
    outs = []
    for name in ("choleskyTest", "choleskyTest03", "triangularTest"):
        src, out = DRIVERS / f"{name}.cpp", DRIVERS / f"{name}.bin"
        if force or _stale(out, [src, DRIVERS / "mtx_io.hpp", DRIVERS / "verify.hpp", LIB, PKG.parent / "include" / "parsy_amd.h"]):
            run([hipcc, "-x", "c++", "-O2", "-std=c++17", str(src), "-x", "none", "-o", str(out), str(LIB),
                 f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/.."])
        outs.append(out)
    # the input converter of the reference's workflow (examples/MakingLowerHalf.cpp): plain host C++
    src, out = DRIVERS / "makingLowerHalf.cpp", DRIVERS / "makingLowerHalf.bin"
    if force or _stale(out, [src]):
        run([hipcc, "-x", "c++", "-O2", "-std=c++17", str(src), "-x", "none", "-o", str(out)])
    outs.append(out)
    return outs


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))

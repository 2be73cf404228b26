"""Builds dpu_olap_amd/lib/libolap_gpu.so (the C-ABI library of include/olap_gpu.h)
from the hand-written HIP sources in dpu_olap_amd/csrc with hipcc for gfx950.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libolap_gpu.so")
SOURCES = ["ctx.hip", "stage.hip", "scan_ops.hip", "join.hip", "gen.hip", "arrow_c.hip", "group.hip"]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libolap_gpu.so)")


def _flags() -> list[str]:
    return [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    import glob
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "olap_gpu.h"), os.path.join(ROOT, "include", "olap_arrow.h"),
                                                               os.path.abspath(__file__)]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([cc, *_flags(), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    return LIB


def build_host(jobs: int = 8, verbose: bool = False) -> str:
    """make -C dpu_olap_amd/host: the C++ Arrow operator classes, the Native (Arrow CPU)
    operators, the benchmark harness and the operator tests (needs g++ and the Arrow C++
    libraries bundled with pyarrow)."""
    host = os.path.join(HERE, "host")
    r = subprocess.run(["make", "-C", host, f"-j{jobs}"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("C++ host build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return os.path.join(host, "build", "olap-query-host")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--host" in sys.argv:
        print(build_host(verbose=True))

"""Build libcpm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO = PKG_DIR.parent
SOURCES = ["cpm_core.hip", "cpm_rng_emission.hip", "cpm_trace.hip", "cpm_sort.hip", "cpm_lightvolume.hip", "cpm_fastvolume.hip",
           "cpm_correlated.hip", "cpm_temporal.hip", "cpm_stream.hip", "cpm_comm.hip", "cpm_gl.hip"]
# -ffp-contract=off: the arithmetic contract spells out every fma
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-ldl"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def build_library(force: bool = False, verbose: bool = True) -> Path:
    out = PKG_DIR / "libcpm_hip.so"
    srcs = [PKG_DIR / "csrc" / s for s in SOURCES]
    deps = srcs + list((PKG_DIR / "csrc").glob("*.h")) + list((REPO / "include" / "cpm").glob("*.h"))
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    cmd = [hipcc(), *FLAGS, "-I", str(REPO / "include"), "-I", str(PKG_DIR / "csrc"), "-o", str(out), *map(str, srcs)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


HOST_SOURCES = ["cpm_processors.cpp", "cpm_timevarying.cpp", "cpm_modules.cpp", "cpm_host_c.cpp", "cpm_hostmath.cpp"]


def build_hostmath_library(force: bool = False, verbose: bool = True) -> Path:
    """libcpm_hostmath.so: the path's host arithmetic alone (light rectangle, transfer-function difference; host/cpm_hostmath.cpp) --
    standard C++, no HIP: what pipeline.py calls, loadable before or without any GPU runtime.  libcpm_host.so carries the same
    translation unit."""
    out = PKG_DIR / "libcpm_hostmath.so"
    src = PKG_DIR / "host" / "cpm_hostmath.cpp"
    deps = [src, PKG_DIR / "host" / "cpm_hostmath.h"]
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    cxx = shutil.which("g++") or hipcc()
    cmd = [cxx, "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", str(PKG_DIR / "host"), "-o", str(out), str(src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_host_library(force: bool = False, verbose: bool = True, extras: bool = False) -> Path:
    """libcpm_host.so: the C++ Processor/Port layer over the C-ABI (host code only, links libcpm_hip.so).
    extras: also the processors outside the workspace's path (RadixSortCL node, UniformGrid3D export / selector / vector source:
    -DCPM_HOST_EXTRAS) -> libcpm_host_extras.so; the default library does not carry them."""
    out = PKG_DIR / ("libcpm_host_extras.so" if extras else "libcpm_host.so")
    srcs = [PKG_DIR / "host" / s for s in HOST_SOURCES]
    deps = srcs + list((PKG_DIR / "host").glob("*.h")) + list((REPO / "include" / "cpm").glob("*.h")) + [PKG_DIR / "libcpm_hip.so"]
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    cmd = [hipcc(), "-x", "c++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", *(["-DCPM_HOST_EXTRAS"] if extras else []),
           "-I", str(REPO / "include"),
           "-I", str(PKG_DIR / "host"), "-I", "/opt/rocm/include", "-o", str(out), *map(str, srcs),
           "-L", str(PKG_DIR), "-lcpm_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    build_hostmath_library(force="--force" in sys.argv)
    build_host_library(force="--force" in sys.argv)

"""Builds tests/fake_rccl/libfake_rccl.so (the RCCL test double: see fake_rccl.c) with hipcc; returns its path."""
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent


def build() -> Path:
    out = HERE / "libfake_rccl.so"
    src = HERE / "fake_rccl.c"
    if out.exists() and out.stat().st_mtime >= src.stat().st_mtime:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-x", "c", "-O2", "-fPIC", "-shared", "-Wall", "-I", "/opt/rocm/include", str(src), "-o", str(out),
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-lpthread", "-lrt", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


if __name__ == "__main__":
    print(build())

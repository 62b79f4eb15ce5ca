"""Build libcygym_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cygym_hip.hip")
INC = os.path.join(os.path.dirname(HERE), "include")
SO = os.path.join(HERE, "libcygym_hip.so")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [SRC, os.path.join(INC, "cygym_abi.h"), os.path.join(INC, "cygym_spec.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INC, "-o", SO, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))

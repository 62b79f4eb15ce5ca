"""Build libcygym_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cygym_hip.hip")
INC = os.path.join(os.path.dirname(HERE), "include")
SO = os.path.join(HERE, "libcygym_hip.so")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    import glob
    deps = [SRC, os.path.join(INC, "cygym_abi.h"), os.path.join(INC, "cygym_spec.h")]
    deps += glob.glob(os.path.join(os.path.dirname(SRC), "*.hpp"))   # the kernel sources included by SRC
    return any(os.path.getmtime(d) > t for d in deps)


RESOURCES = os.path.join(HERE, "build_resources.json")


def _parse_resources(text: str) -> dict:
    """kernel name -> {vgprs, sgpr_spill, vgpr_spill, scratch} from -Rpass-analysis=kernel-resource-usage."""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INC, "-o", SO, SRC,
           "-Rpass-analysis=kernel-resource-usage"]
    cmd += os.environ.get("CYGYM_BUILD_FLAGS", "").split()   # development: e.g. -DCG_DEV_MT=256 -DCG_DEV_WPB=8
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stdout[-4000:])
    with open(RESOURCES, "w") as f:
        json.dump(_parse_resources(p.stdout), f, indent=1, sort_keys=True)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))

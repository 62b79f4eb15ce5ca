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
    deps += [f for pat in ("*.hpp", "*.hip", "*.inc") for f in glob.glob(os.path.join(os.path.dirname(SRC), pat))]
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


INST = os.path.join(HERE, "csrc", "cg_inst.hip")
INST_ACTOR = os.path.join(HERE, "csrc", "cg_inst_actor.hip")
N_GROUPS = 8   # CG_INST_GROUPS of csrc/cg_device.hpp
GROUP_MT = {0: 256, 1: 256, 2: 64, 3: 64, 4: 0, 5: 0, 6: 0, 7: 0}   # device-count class each instantiation group holds


def build_to(so: str, resources: str | None = None, flags: list[str] | None = None, dev_mt: int | None = None,
             verbose: bool = False, jobs: int | None = None) -> str:
    """Compile the C-ABI unit and the instantiation units (csrc/cg_inst.hip, one per group of step_kernel variants)
    in parallel, then link `so`.  dev_mt (development): only the groups of that device-count class (0, 64 or 256),
    with -DCG_DEV_MT so that handles of the other classes fail at cygym_create."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = list(flags or []) + os.environ.get("CYGYM_BUILD_FLAGS", "").split()
    if dev_mt is not None:
        flags.append(f"-DCG_DEV_MT={dev_mt}")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-Rpass-analysis=kernel-resource-usage"] + flags
    groups = [g for g in range(N_GROUPS) if dev_mt is None or GROUP_MT[g] == dev_mt]
    jobs = jobs or int(os.environ.get("CYGYM_BUILD_JOBS", "0")) or min(len(groups) + 1, os.cpu_count() or 4)
    with tempfile.TemporaryDirectory(prefix="cygym_build_") as tmp:
        units = [("main", base + ["-c", SRC, "-o", os.path.join(tmp, "main.o")])]
        units += [(f"inst{g}", base + [f"-DCG_INST_GROUP={g}", "-c", INST, "-o", os.path.join(tmp, f"inst{g}.o")]) for g in groups]
        if dev_mt is None or dev_mt == 256:   # the tick + actor kernels (256 devices only)
            units.append(("inst_actor", base + ["-c", INST_ACTOR, "-o", os.path.join(tmp, "inst_actor.o")]))

        def run(unit):
            name, cmd = unit
            if verbose:
                print(" ".join(cmd), flush=True)
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode != 0:
                raise RuntimeError(f"hipcc failed on unit {name}:\n" + p.stdout[-4000:])
            return p.stdout

        with ThreadPoolExecutor(max_workers=jobs) as ex:
            logs = list(ex.map(run, units))
        objs = [u[1][-1] for u in units]
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", so] + objs   # -z defs: a kernel variant
        # declared in the C-ABI unit but missing from every instantiation group fails the build, not the first dlopen
        if verbose:
            print(" ".join(link), flush=True)
        p = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed:\n" + p.stdout[-4000:])
    if resources:
        res = {}
        for text in logs:
            res.update(_parse_resources(text))
        with open(resources, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    return so


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    return build_to(SO, RESOURCES, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))

#!/usr/bin/env python3
"""tools/rocprof_summary.py -- turn rocprofv3 output of `python3 bench.py ...` into the summaries kept under profiles/.

Recipe (on the GPU box; counters in their own passes, never together with a trace):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt  -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline
  python3 tools/rocprof_summary.py --steps 200 --kernel-stats $OUT/kt --pmc $OUT/pmc_fetch $OUT/pmc_write \
          --out-stats profiles/rNN_target_kernel_stats.csv --out-pmc profiles/rNN_target_pmc_summary.json

The PMC summary holds, per kernel class (`per_tick` = step_kernel<.., FUSED=0>, `fused` = step_kernel<.., FUSED=1>),
the mean counter value per launch (template arguments: <waves per workgroup, compile-time M, FUSED, XE, WIDE>); FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3's unit); bench.py applies the
gfx950 correction (FETCH_SIZE x2, MI355X_MICROARCH.md) when it turns them into `roofline.traffic`.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import re
import shutil
from collections import defaultdict


def kernel_class(name: str):
    if "step_kernel" not in name:
        return None
    m = re.search(r"step_kernelILi\d+ELi\d+ELb([01])", name)           # mangled: <WPB, MT, FUSED, ...>
    if m:
        return "fused" if m.group(1) == "1" else "per_tick"
    m = re.search(r"step_kernel<\s*\d+\s*,\s*\d+\s*,\s*(true|false)", name)   # demangled
    if m:
        return "fused" if m.group(1) == "true" else "per_tick"
    return "per_tick"


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def summarize_pmc(dirs, steps):
    acc = defaultdict(lambda: defaultdict(list))   # class -> counter -> per-dispatch values
    for d in dirs:
        for path in find(d, "*counter_collection.csv"):
            per_dispatch = defaultdict(float)
            meta = {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    cls = kernel_class(row.get("Kernel_Name", ""))
                    if cls is None:
                        continue
                    key = (row.get("Dispatch_Id"), row.get("Counter_Name"))
                    per_dispatch[key] += float(row.get("Counter_Value", 0.0))   # summed over XCDs / instances
                    meta[key] = cls
            for key, v in per_dispatch.items():
                acc[meta[key]][key[1]].append(v)
    out = {}
    for cls, counters in acc.items():
        out[cls] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(counters.items())}
        out[cls]["ticks_per_launch"] = steps if cls == "fused" else 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, required=True, help="--steps of the profiled bench command (ticks per fused launch)")
    ap.add_argument("--kernel-stats", help="directory of the --kernel-trace --stats pass")
    ap.add_argument("--pmc", nargs="*", default=[], help="directories of the --pmc passes")
    ap.add_argument("--out-stats")
    ap.add_argument("--out-pmc")
    a = ap.parse_args()
    if a.kernel_stats and a.out_stats:
        c = find(a.kernel_stats, "*kernel_stats.csv")
        if not c:
            raise SystemExit(f"no *kernel_stats.csv under {a.kernel_stats}")
        shutil.copyfile(c[0], a.out_stats)
        print("kernel stats ->", a.out_stats)
    if a.pmc and a.out_pmc:
        s = summarize_pmc(a.pmc, a.steps)
        if not s:
            raise SystemExit("no step_kernel rows in the counter CSVs")
        with open(a.out_pmc, "w") as f:
            json.dump(s, f, indent=1, sort_keys=True)
        print("pmc summary ->", a.out_pmc, {k: sorted(v) for k, v in s.items()})


if __name__ == "__main__":
    main()

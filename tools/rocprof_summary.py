#!/usr/bin/env python3
"""tools/rocprof_summary.py -- turn rocprofv3 output of `python3 bench.py ...` into the summaries kept under profiles/.

Recipe (on the GPU box; counters in their own passes, never together with a trace) -- tools/profile_all.sh runs it
for every single-GPU workload:

  cd /tmp && export TMPDIR=/tmp
  B="python3 $REPO/bench.py --workload W --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/W/kt        -- $B
  rocprofv3 --pmc FETCH_SIZE       --output-format csv -d $OUT/W/pmc_fetch -- $B
  rocprofv3 --pmc WRITE_SIZE       --output-format csv -d $OUT/W/pmc_write -- $B
  python3 tools/rocprof_summary.py --root $OUT --steps 20 --workloads target cfg2 cfg3 cfg5 \
          --out-pmc profiles/r02_pmc_summary.json --out-stats profiles/r02_kernel_stats.json

Kernel classes: `per_tick` = step_kernel<.., FUSED=0, ..>, `fused` = step_kernel<.., FUSED=1, ..> (template arguments:
<waves per workgroup, compile-time M, FUSED, full-feature, WIDE>).  bench.py issues the per-tick kernel in two
shapes: one launch over the whole batch, and `--sub-batches` S launches over N/S envs each; dispatches are therefore
keyed by (class, envs per launch = grid size / 64).

The PMC summary holds, per workload and class, the mean FETCH_SIZE / WRITE_SIZE per FULL-BATCH launch in KiB
(rocprofv3's unit); bench.py applies the gfx950 correction (FETCH_SIZE x 2, MI355X_MICROARCH.md) when it turns them
into `roofline.traffic`.  The kernel-stats summary holds launches and mean / min / max duration per (class, envs per
launch) from the kernel trace -- the figure bench.py's `launch_us` (HIP events) must agree with for the full-batch
shape."""
from __future__ import annotations

import argparse
import csv
import glob
import json
import sys
import os
import re
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


def envs_of(row):
    for key in ("Grid_Size", "Grid_Size_X", "grid_size"):
        if row.get(key):
            try:
                return int(float(row[key])) // 64
            except ValueError:
                pass
    return -1


def pmc_per_shape(dirs):
    """(class, envs per launch) -> counter -> list of per-dispatch values (summed over XCDs / instances)."""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in find(d, "*counter_collection.csv"):
            per, meta = defaultdict(float), {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    cls = kernel_class(row.get("Kernel_Name", ""))
                    if cls is None:
                        continue
                    key = (row.get("Dispatch_Id"), row.get("Counter_Name"))
                    per[key] += float(row.get("Counter_Value", 0.0))
                    meta[key] = (cls, envs_of(row))
            for key, v in per.items():
                acc[meta[key]][key[1]].append(v)
    return acc


def trace_per_shape(d):
    """(class, envs per launch) -> list of kernel durations in microseconds, from *kernel_trace.csv."""
    out = defaultdict(list)
    for path in find(d, "*kernel_trace.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                cls = kernel_class(row.get("Kernel_Name", ""))
                if cls is None:
                    continue
                try:
                    dur = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
                except (KeyError, ValueError):
                    continue
                out[(cls, envs_of(row))].append(dur)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True, help="directory holding <workload>/{kt,pmc_fetch,pmc_write}")
    ap.add_argument("--steps", type=int, required=True, help="--steps of the profiled bench command (ticks per fused launch)")
    ap.add_argument("--workloads", nargs="+", required=True)
    ap.add_argument("--out-pmc")
    ap.add_argument("--out-stats")
    a = ap.parse_args()
    pmc_out, stats_out = {}, {}
    for w in a.workloads:
        base = os.path.join(a.root, w)
        shapes = pmc_per_shape([os.path.join(base, "pmc_fetch"), os.path.join(base, "pmc_write")])
        if shapes:
            full = max(n for (_, n) in shapes)          # the full-batch launch shape
            rec = {"envs_per_launch": full}
            for (cls, n), counters in shapes.items():
                if n != full:
                    continue
                rec[cls] = {"ticks_per_launch": a.steps if cls == "fused" else 1}
                for c, v in counters.items():
                    rec[cls][f"{c}_KiB_per_launch"] = sum(v) / len(v)
                    rec[cls][f"{c}_launches"] = len(v)
            pmc_out[w] = rec
        tr = trace_per_shape(os.path.join(base, "kt"))
        if tr:
            stats_out[w] = [{"kernel": cls, "envs_per_launch": n, "launches": len(v), "mean_us": sum(v) / len(v),
                             "min_us": min(v), "max_us": max(v)} for (cls, n), v in sorted(tr.items())]
    if a.out_pmc and pmc_out:
        with open(a.out_pmc, "w") as f:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import kernel_source_hash      # the summary is tied to the kernels it was taken on: bench.py drops it on a mismatch
            pmc_out["kernel_source_hash"] = kernel_source_hash()
            json.dump(pmc_out, f, indent=1, sort_keys=True)
        print("pmc summary ->", a.out_pmc, sorted(pmc_out))
    if a.out_stats and stats_out:
        with open(a.out_stats, "w") as f:
            json.dump(stats_out, f, indent=1, sort_keys=True)
        print("kernel stats ->", a.out_stats, sorted(stats_out))


if __name__ == "__main__":
    main()

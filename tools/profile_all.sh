#!/bin/bash
# Collect the rocprofv3 evidence for every single-GPU workload of bench.py (run on the GPU box, from the repo root):
#   bash tools/profile_all.sh gpurun_out/r02/prof [workloads...]
# Counters go in their own passes (never together with a trace); the program after `--` is python3 itself.
set -e
OUT=${1:-gpurun_out/prof}; shift || true
WL=${@:-target cfg2 cfg3 cfg5}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  B="python3 $REPO/bench.py --workload $w --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-configs --no-closed-loop"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/kt" -- $B > "$OUT/$w.kt.json" 2> "$OUT/$w.kt.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$w/pmc_fetch" -- $B > /dev/null 2> "$OUT/$w.pf.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$w/pmc_write" -- $B > /dev/null 2> "$OUT/$w.pw.err"
  echo "profiled $w"
done
cd "$REPO"
python3 tools/rocprof_summary.py --root "$OUT" --steps 20 --workloads $WL --out-pmc "$OUT/pmc_summary.json" --out-stats "$OUT/kernel_stats.json"
# the raw CSVs are large: keep only the summaries and the stats tables
for w in $WL; do
  for f in $(find "$OUT/$w/kt" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/${w}_kernel_stats.csv"; done
  rm -rf "$OUT/$w"
done

#!/bin/bash
# rocprofv3 evidence for the closed-loop consumer (run on the GPU box, from the repo root):
#   bash tools/profile_closed_loop.sh gpurun_out/r03c
# Kernel trace (+ stats) of the eager loop per grid shape; FETCH_SIZE / WRITE_SIZE in their own passes (never with a trace).
set -e
OUT=${1:-gpurun_out/prof_cl}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for g in 1x1 2x2; do
  C="python3 $REPO/tools/exp_closed_loop.py --grid $g --no-randomize --eager --ticks 100 --loop-only --no-merge"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$g" -- $C > "$OUT/kt_$g.out" 2> "$OUT/kt_$g.err"
  for f in $(find "$OUT/kt_$g" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/closed_loop_${g}_eager_kernel_stats.csv"; done
  rm -rf "$OUT/kt_$g"
  echo "traced $g"
done
# the same loop with the tick and the next role's actor as one kernel (cygym_step_actor)
C="python3 $REPO/tools/exp_closed_loop.py --grid 1x1 --no-randomize --eager --ticks 100 --loop-only"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_merged" -- $C > "$OUT/kt_merged.out" 2> "$OUT/kt_merged.err"
for f in $(find "$OUT/kt_merged" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/closed_loop_1x1_one_launch_eager_kernel_stats.csv"; done
rm -rf "$OUT/kt_merged"
C="python3 $REPO/tools/exp_closed_loop.py --grid 1x1 --no-randomize --eager --ticks 100 --loop-only --no-merge"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- $C > /dev/null 2> "$OUT/pmc_$c.err"
  python3 - "$OUT/pmc_$c" $c > "$OUT/closed_loop_1x1_$c.txt" <<'PY'
import csv, glob, sys, collections
root, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != name:
            continue
        k = r["Kernel_Name"].split("(")[0][:90]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
print(f"# {name} per launch (raw counter units: KiB on gfx950; FETCH_SIZE counts 64-byte requests as 32: x2 per MI355X_MICROARCH.md), mean over launches")
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:90s} launches {n:6d}  mean {v / n:12.1f}")
PY
  rm -rf "$OUT/pmc_$c"
  echo "counted $c"
done
cd "$REPO"

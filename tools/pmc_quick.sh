# rocprofv3 counter passes of the `target` bench command (no traces in the same run): bash tools/pmc_quick.sh <outdir> "<C1 C2 ..>" ["<D1 D2 ..>" ...]
OUT=$1; shift
REPO=$(pwd); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 $REPO/bench.py --workload ${PMC_WORKLOAD:-target} --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-configs --no-closed-loop > /dev/null 2> "$OUT/p$i.err"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" not in k: continue
        k = k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: (round(v / n[(k, c)], 1), n[(k, c)]) for c, v in d.items()})
PY

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
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = collections.defaultdict(dict)
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
        for c, v in d.items():
            summary[k][c] = {"per_launch": v / n[(k, c)], "launches": n[(k, c)]}
for k, d in summary.items():   # per-wave figures (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles: x4 = shader cycles)
    w = d.get("SQ_WAVES", {}).get("per_launch")
    if w:
        d["per_wave"] = {c: v["per_launch"] / w * (4.0 if c.startswith(("SQ_WAVE_CYCLES", "SQ_WAIT", "SQ_ACTIVE_INST", "SQ_BUSY")) else 1.0)
                         for c, v in d.items() if c != "SQ_WAVES" and isinstance(v, dict) and "per_launch" in v}
        d["per_wave_note"] = "instructions per wave and launch; cycle counters in shader cycles (quad-cycle counts x 4)"
json.dump(summary, open(out + "/sq_counters.json", "w"), indent=1, sort_keys=True)
PY

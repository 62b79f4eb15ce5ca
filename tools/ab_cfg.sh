# A/B of prebuilt libraries on one workload: bash tools/ab_cfg.sh <workload> <lib>... (names under cygym_amd/, or `default`), two rounds each
W=$1; shift
for i in 1 2; do
for so in "$@"; do
  if [ "$so" = default ]; then unset CYGYM_SO; else export CYGYM_SO=$GRAFT_REPO_ROOT/cygym_amd/$so; fi
  python bench.py --workload $W --no-cpu-baseline --no-closed-loop --no-configs --steps 20 --warmup 5 --reps 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['per_tick_stepping'] if 'per_tick_stepping' in d else d
r=d['roofline']; sl=(d.get('per_tick_stepping') or {}).get('single_launch')
print('$W $so', 'value %.3e' % d['value'], 'launch %.2f us' % r['launch_us'], 'frac %.3f' % r['frac'], ('single %.3f' % sl['roofline']['frac']) if sl else '', 'rollout %.3e' % d['fused_rollout']['value'])"
done; done

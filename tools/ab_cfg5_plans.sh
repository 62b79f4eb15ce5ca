for i in 1 2; do
for v in "" "CYGYM_LISTS_LDS=1"; do
  env $v python bench.py --workload cfg5 --no-cpu-baseline --no-closed-loop --no-configs --steps ${STEPS:-10} --warmup 3 --reps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
r=d['roofline']; sl=(d.get('per_tick_stepping') or {}).get('single_launch')
print('cfg5 [$v]', 'value %.3e' % d['value'], 'launch %.2f us' % r['launch_us'], 'frac %.3f' % r['frac'], ('single %.3f' % sl['roofline']['frac']) if sl else '', 'rollout %.3e' % d['fused_rollout']['value'], 'frac %.3f' % d['fused_rollout']['roofline']['frac'])"
done; done

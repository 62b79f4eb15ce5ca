# A/B of prebuilt libraries on the target workload: bash tools/ab.sh <lib>... (names under cygym_amd/), two rounds each
for i in 1 2; do
for so in "$@"; do
  if [ "$so" = default ]; then unset CYGYM_SO; else export CYGYM_SO=$GRAFT_REPO_ROOT/cygym_amd/$so; fi
  python bench.py --no-cpu-baseline --no-closed-loop --no-configs --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so', '%.3e' % d['value'], '%.2f us' % d['roofline']['launch_us'], 'rollout %.3e' % d['fused_rollout']['value'])"
done; done

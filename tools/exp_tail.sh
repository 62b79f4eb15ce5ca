# On the GPU box: bash tools/exp_tail.sh <ticks> <variant>...   (variants built by tools/exp_build.sh; two rounds each)
T=$1; shift
for r in 1 2; do for v in "$@"; do
  TAIL_BRIEF=1 TAIL_SO=$GRAFT_REPO_ROOT/cygym_amd/libcygym_exp_$v.so TAIL_JSON=gpurun_out/exp_$v.json python tools/tail_hist.py 4096 256 $T 2>/dev/null
done; done

# Build a named stamps variant of the 256-device kernels: bash tools/exp_build.sh <name> "<extra flags>"
# -> cygym_amd/libcygym_exp_<name>.so (git-ignored; travels to the GPU box)
set -e
name=$1; shift
python - "$name" "$@" <<'PY'
import sys
from cygym_amd import build as B
name, flags = sys.argv[1], " ".join(sys.argv[2:]).split()
so = f"cygym_amd/libcygym_exp_{name}.so"
B.build_to(so, so + ".resources.json", flags=["-DCG_STAMPS"] + flags, dev_mt=256)
import json
r = json.load(open(so + ".resources.json"))
for k, v in r.items():
    if "ILi16ELi256ELb0ELb0ELb1E" in k and "step_kernel" in k:
        print(name, v)
PY

"""Development aid: compile only the instantiation groups of one device-count class into a scratch .so and print the
compiler's resource report (VGPRs / spills / scratch / occupancy).

    python tools/devbuild.py 256                 # the 256-device kernels only -> /tmp/cygym_dev.so  (about 40 s)
    python tools/devbuild.py 0 "" path/lib.so    # run-time sizes, named output   (second argument: unused, kept for old habits)
    CYGYM_SO=/tmp/cygym_dev.so python bench.py ...       # run against it (cygym_amd/_lib.py honours CYGYM_SO)
Extra compiler flags: CYGYM_BUILD_FLAGS="-DCG_LEAN_LB=5 ...".
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cygym_amd.build as b   # noqa: E402


def main():
    mt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/cygym_dev.so"
    res = out + ".resources.json"
    b.build_to(out, res, dev_mt=mt)
    r = json.load(open(res))
    for k, v in sorted(r.items()):
        m = re.search(r"ILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)", k)
        if "step_kernel" in k and m:
            print("WPB=%s MT=%s FUSED=%s XE=%s WIDE=%s" % m.groups(), v)


if __name__ == "__main__":
    main()

"""Development aid: compile a subset of the tick-kernel instantiations into a scratch .so and print the compiler's
resource report (VGPRs / spills / scratch / occupancy).

    python tools/devbuild.py 256 8          # MT=256, WPB=8 only -> /tmp/cygym_dev.so  (about 25 s instead of 3 min)
    CYGYM_SO=/tmp/cygym_dev.so python bench.py ...       # run against it (cygym_amd/_lib.py honours CYGYM_SO)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cygym_amd.build as b   # noqa: E402


def main():
    mt = sys.argv[1] if len(sys.argv) > 1 else "256"
    wpb = sys.argv[2] if len(sys.argv) > 2 else ""
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/cygym_dev.so"
    flags = f"-DCG_DEV_MT={mt}" + (f" -DCG_DEV_WPB={wpb}" if wpb else "")
    os.environ["CYGYM_BUILD_FLAGS"] = (os.environ.get("CYGYM_BUILD_FLAGS", "") + " " + flags).strip()
    b.SO, b.RESOURCES = out, out + ".resources.json"
    b.build(force=True)
    r = json.load(open(b.RESOURCES))
    for k, v in sorted(r.items()):
        m = re.search(r"ILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)", k)
        if "step_kernel" in k and m:
            print("WPB=%s MT=%s FUSED=%s XE=%s WIDE=%s" % m.groups(), v)


if __name__ == "__main__":
    main()

"""Code-object metadata of the built gfx950 objects: per kernel VGPRs / SGPRs used and spilled, scratch bytes, LDS.
    python scripts/kernel_meta.py [object ...] [--filter substring]      (default: every .o under hypad_amd/lib)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypad_amd import build  # noqa: E402

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    if flt in args:
        args.remove(flt)
    objs = args or [os.path.join(build.LIB_DIR, f) for f in sorted(os.listdir(build.LIB_DIR)) if f.endswith(".o")]
    for o in objs:
        for k in build.kernel_metadata(o):
            if flt in k["name"]:
                print("%-110s vgpr %3d (+%d agpr) spill %3d | sgpr %3d spill %3d | scratch %4d B | lds %6d" % (
                    k["name"][:110], k["vgpr_count"], k.get("agpr_count", 0), k["vgpr_spill_count"], k["sgpr_count"], k["sgpr_spill_count"],
                    k["private_segment_fixed_size"], k["group_segment_fixed_size"]))

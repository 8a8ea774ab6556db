"""Per-kernel medians of every counter found under the given rocprofv3 output directories (one --pmc pass each):
    python scripts/sq_summary.py DIR [DIR ...] [--match regex]"""
import csv, glob, json, re, sys
import numpy as np
args = sys.argv[1:]
match = "score_forward_packed|kde_mode|unroll_median|precompute|gen_kernel|dw_adam|critic_persistent"
if "--match" in args:
    i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
acc = {}
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(match, r["Kernel_Name"])
            if not m:
                continue
            a = acc.setdefault(m.group(0), {}).setdefault(r["Counter_Name"], {})
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
out = {k: {c: float(np.median(list(v.values()))) for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))

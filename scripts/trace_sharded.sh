#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_sharded; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/scripts/trace_sharded.py > $OUT/run.log 2> $OUT/run.err
cd $REPO; cat $OUT/run.log
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by the spin kernel
calls, cur = [], []
for r in rows:
    if "spin" in r["Kernel_Name"] or "sleep" in r["Kernel_Name"].lower():
        if cur: calls.append(cur)
        cur = []
    else:
        cur.append(r)
if cur: calls.append(cur)
def show(c, title):
    t0 = int(c[0]["Start_Timestamp"]); prev = t0; busy = 0
    print(title, "kernels", len(c))
    for r in c:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"]); nm = m.group(1) if m else r["Kernel_Name"][:40]
        print("  %8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, nm[:60]))
        prev = e; busy += e - s
    print("  span %.1f us, busy %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
# calls: [warm hyper (no marker before -> merged with setup)], 6 hyper, warm dtw + ..., pick by position
print(len(calls), "segments")
if len(calls) >= 6: show(calls[4], "hyperbolic call")
if len(calls) >= 12: show(calls[-2], "dtw call")
PY

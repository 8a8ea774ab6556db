"""Summarise a rocprofv3 kernel trace CSV: per-kernel mean duration and mean gap to the previous kernel's end."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev_end is not None:
        gap[n].append(s - prev_end)
    prev_end = e
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d = dur[n]; g = gap[n] or [0]
    print(f"{n:60s} n={len(d):5d} dur mean {sum(d)/len(d)/1e3:8.2f} us  total {sum(d)/1e6:8.2f} ms  gap-before mean {sum(g)/len(g)/1e3:7.2f} us")

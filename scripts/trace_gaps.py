"""Summarise a rocprofv3 kernel trace CSV: per-kernel count, mean duration, total, mean gap to the previous kernel's end."""
import collections, csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:50]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev_end is not None:
        gap[n].append(s - prev_end)
    prev_end = e
for n in sorted(dur, key=lambda k: -sum(dur[k]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    d = dur[n]; g = gap[n] or [0]
    print(f"{n:36s} n={len(d):5d} dur mean {sum(d)/len(d)/1e3:8.2f} med {statistics.median(d)/1e3:8.2f} us  total {sum(d)/1e6:8.2f} ms  gap-before mean {sum(g)/len(g)/1e3:7.2f} med {statistics.median(g)/1e3:6.2f} us")

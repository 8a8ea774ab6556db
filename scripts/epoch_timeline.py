"""One replayed epoch as rocprofv3 saw it: every kernel of the LAST epoch in a kernel-trace CSV, start relative to the epoch's first kernel,
duration, gap to the previous kernel's end.  Usage: rocprofv3 --kernel-trace --output-format csv -d D -o t -- python3 bench.py --steps 5 --warmup 2
--no-cpu-baseline --no-secondary --no-drop-in --no-sharded-scoring --no-extra-configs --no-scoring ; python scripts/epoch_timeline.py D/**/t_kernel_trace.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: (re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"]) or [None, r["Kernel_Name"][:40]])[1]
# epochs: from an epoch_shuffle_kernel to the next one
starts = [i for i, r in enumerate(rows) if name(r) == "epoch_shuffle_kernel"]
pick = None
for a, b in zip(starts, starts[1:]):
    seg = rows[a:b]
    if sum(name(r) == "gen_kernel" for r in seg) == 29 and sum(name(r) == "critic_persistent_kernel" for r in seg) == 1:
        pick = seg          # (the last complete graph-replayed epoch)
t0 = int(pick[0]["Start_Timestamp"]); prev = None; gaps = 0; busy = 0
for i, r in enumerate(pick):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = 0 if prev is None else s - prev
    gaps += max(g, 0); busy += e - s
    if i < 8 or i >= len(pick) - 6 or name(r) not in ("gen_kernel", "dw_adam_kernel"):
        print(f"{(s - t0) / 1e3:9.2f} us  {name(r):32s} dur {(e - s) / 1e3:8.2f}  gap-before {g / 1e3:6.2f}")
    prev = e
print(f"kernels {len(pick)}  span {(prev - t0) / 1e3:.2f} us  busy {busy / 1e3:.2f}  gaps {gaps / 1e3:.2f}")
import statistics
for k in ("gen_kernel", "dw_adam_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in pick if name(r) == k]
    print(f"{k}: first {d[0]:.2f} us, others mean {statistics.mean(d[1:]):.2f} median {statistics.median(d[1:]):.2f}")

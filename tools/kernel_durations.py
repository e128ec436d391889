"""Distribution of one kernel's durations over the timed steps of a rocprofv3 kernel trace: python tools/kernel_durations.py <trace.csv> <name substring> [...]"""
import csv, sys
rows = []
for r in csv.reader(open(sys.argv[1])):
    if len(r) >= 11 and r[9].isdigit():
        rows.append((int(r[9]), int(r[10]), r[7]))
rows.sort()
for pat in sys.argv[2:]:
    d = sorted((e - s) / 1e3 for s, e, n in rows[len(rows) // 2:] if pat in n)
    if d:
        q = lambda f: d[min(len(d) - 1, int(f * len(d)))]
        print(f"{pat:36s} n={len(d):5d}  min {d[0]:7.1f}  p25 {q(.25):7.1f}  median {q(.5):7.1f}  p75 {q(.75):7.1f}  p95 {q(.95):7.1f}  max {d[-1]:7.1f} us")

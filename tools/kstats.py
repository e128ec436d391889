#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms")

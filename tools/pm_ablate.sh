#!/bin/bash
# Phase ablation of the partition-major fill (wrong results, timing only): fill kernel time per measurement switch under rocprofv3.
set -u
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/prof_pm
  env N2M_BIN_PM=1 $cfg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pm -- python $R/tools/pair_bench.py --reps ${REPS:-30} > /tmp/prof_pm.log 2>&1
  python - "$cfg" <<PY
import csv, glob, sys
f = glob.glob('/tmp/prof_pm/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
out = []
for r in rows:
    n = r['Name']
    if 'fill_pair' in n or 'accumulate' in n:
        tag = ('fillTV' if 'ILi1E' in n else 'fill  ') if 'fill' in n else ('acc16' if 'DF16' in n else 'acc32')
        out.append(f"{tag} {float(r['AverageNs'])/1e3:7.1f}")
print(f"{sys.argv[1]:50s} | " + " | ".join(sorted(out)))
PY
done

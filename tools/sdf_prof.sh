#!/bin/bash
# Kernel-time split of the SDF recipe's step (end of the schedule): rocprofv3 --kernel-trace --stats over bench.py --recipe sdf, per-step averages of the
# timed window's kernels.    bash tools/sdf_prof.sh [tag]
set -u
R=$(pwd); TAG=${1:-sdf}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sdf
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sdf -- python $R/bench.py --recipe sdf --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --no-prof > $O/sdf.json 2> $O/sdf.err
python - <<'PY' | tee $O/sdf_kernels.txt
import csv, glob
from collections import defaultdict
f = glob.glob('/tmp/prof_sdf/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 40 steps: a step = kernels between two adam launches
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = idx[-41], idx[-1]
acc, cnt = defaultdict(float), defaultdict(int)
for r in rows[lo + 1:hi + 1]:
    n = r['Kernel_Name'][:100]
    acc[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    cnt[n] += 1
tot = sum(acc.values())
wall = (int(rows[hi]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / 1e3 / 40
print(f"per step over the last 40: wall {wall:.1f} us, kernel time {tot / 40:.1f} us (both streams)")
for n, t in sorted(acc.items(), key=lambda kv: -kv[1])[:28]:
    print(f"  {t / 40:8.1f} us/step  x{cnt[n] / 40:5.2f}  avg {t / cnt[n]:8.1f}  {n}")
PY

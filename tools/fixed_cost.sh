#!/bin/bash
# every kernel's time at a tiny batch (fixed cost) next to the full batch: tools/fixed_cost.sh
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for NP in 8192 262144; do
  rm -rf /tmp/fc_p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_p -- python $R/bench.py --no-cpu-baseline --no-prof --pretrain 100 --warmup 10 --steps 100 --num-points $NP > /tmp/fc.log 2>&1
  echo "num_points=$NP: $(python $R/tools/show_bench.py /tmp/fc.log | head -1 | cut -c1-110)"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/fc_p/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if int(r['Calls'])>=100]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print('   %-64s calls %5s avg %8.1f us' % (r['Name'].replace('(anonymous namespace)::','').replace('_ZN12_GLOBAL__N_1','')[:64], r['Calls'], float(r['AverageNs'])/1e3))
PY
done

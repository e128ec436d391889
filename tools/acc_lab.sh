#!/bin/bash
# per-kernel times of the shared binned backward under measurement switches (N2M_ACC_DEBUG bits: 1 no LDS atomics, 2 no flush, 4 no entry loads)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for dbg in 0 1 2 4 7; do
  rm -rf /tmp/acc_p
  N2M_ACC_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/acc_p -- python $R/tools/pair_bench.py --reps 20 > /tmp/acc.log 2>&1
  echo "dbg=$dbg"; python - <<PY
import csv,glob
f=glob.glob('/tmp/acc_p/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bin_' in r['Name']: print('   %-60s calls %5s avg %8.1f us' % (r['Name'][20:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done

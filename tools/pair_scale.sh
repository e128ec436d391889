#!/bin/bash
# per-kernel times of the shared binned backward at several batch sizes (fixed cost vs per-sample cost)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for B in 16384 65536 262144 524288; do
  rm -rf /tmp/acc_p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/acc_p -- python $R/tools/pair_bench.py --reps 20 --B $B > /tmp/acc.log 2>&1
  echo "B=$B"; python - <<PY
import csv,glob
f=glob.glob('/tmp/acc_p/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bin_' in r['Name']: print('   %-60s calls %5s avg %8.1f us' % (r['Name'][20:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done

R=$(pwd); cd /tmp; export TMPDIR=/tmp
for G in 4096 512 768 1024; do
  rm -rf /tmp/acc_p
  N2M_ACC_GRID=$G rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/acc_p -- python $R/tools/pair_bench.py --reps 20 > /tmp/acc.log 2>&1
  echo "grid cap=$G"; python - <<PY
import csv,glob
f=glob.glob('/tmp/acc_p/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bin_acc' in r['Name']: print('   %-60s calls %5s avg %8.1f us' % (r['Name'][20:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done

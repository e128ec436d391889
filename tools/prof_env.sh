#!/bin/bash
# Kernel stats of the bench command under environment settings given as arguments ("VAR=a VAR2=b" per run, one quoted string each).
# Traces stay in /tmp, the stats summaries come back.   tools/prof_env.sh tag "N2M_PM_SPLIT=1 N2M_PM_FINE_GROUPS=64" "N2M_PM_SPLIT=0" ...
set -u
R=$(pwd); TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1)); rm -rf /tmp/prof_e$i
  env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e$i -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-prof > $O/bench$i.json 2>$O/bench$i.err
  f=$(find /tmp/prof_e$i -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_$i.csv
  echo "== $envs"
  python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("pm_fill", "pm_accumulate", "adam_kernel", "forward3_packed", "field_backward", "field_forward")):
        print(f"  {n[:64]:64s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done

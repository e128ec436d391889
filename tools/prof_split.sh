#!/bin/bash
# Kernel stats of the bench command with the fill split on / off (traces stay in /tmp, the stats summaries come back).   tools/prof_split.sh [tag]
set -u
R=$(pwd); TAG=${1:-r6d}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_v$v
  N2M_PM_SPLIT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v$v -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-prof > $O/bench$v.json 2>$O/bench$v.err
  f=$(find /tmp/prof_v$v -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_split$v.csv
  echo "== SPLIT=$v"; head -12 $f | cut -c1-160
done

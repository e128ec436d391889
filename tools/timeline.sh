#!/bin/bash
# Kernel trace of a short bench run -> per-kernel stats + the timeline of one steady-state step (run on the GPU box from the repo root).
# usage: tools/timeline.sh <tag> [extra bench.py flags]
set -u
R=$(pwd); TAG=${1:-x}; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --no-cpu-baseline --no-prof --pretrain 300 --warmup 20 --steps 200 "$@" > $R/gpurun_out/$TAG/bench_prof.json 2>/tmp/ps.log
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$TAG/kernel_stats.csv
python $R/tools/step_timeline.py $(find /tmp/prof_s -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/$TAG/timeline.txt
cat $R/gpurun_out/$TAG/timeline.txt

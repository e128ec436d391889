#!/bin/bash
# Measurement set of a round (run on the GPU box from the repo root): PMC traffic passes, kernel trace of the bench command itself (stats,
# one step's timeline, events-vs-trace cross-check), default bench lines (stage 0 with the CPU leg, stage 1, sdf, garden).
# Outputs land in gpurun_out/<tag>/; copy what is to be kept into profiles/.      tools/collect_profiles.sh r02
set -u
R=$(pwd); TAG=${1:-r02}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --pretrain 200 --warmup 5 --steps 20 --no-prof --no-cpu-baseline"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/prof_s
# counters in their own passes, with --kernel-trace only (no other trace domains)
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $B > /tmp/pf.log 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/${TAG}_pmc_traffic.json > $O/pmc_traffic.txt 2>&1; head -12 $O/pmc_traffic.txt
# the bench command itself under the tracer (per-kernel hipEvents on, as the driver runs it minus the CPU leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --no-cpu-baseline > $O/${TAG}_bench_traced.json 2>/tmp/ps.log
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $O/${TAG}_step_kernel_stats.csv
TR=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR > $O/${TAG}_step_timeline.txt
python $R/tools/trace_vs_events.py $TR $O/${TAG}_bench_traced.json > $O/${TAG}_trace_vs_events.txt 2>&1; cat $O/${TAG}_trace_vs_events.txt
cd $R
python bench.py > $O/${TAG}_bench.json 2>$O/bench.err; python tools/show_bench.py $O/${TAG}_bench.json
python bench.py --stage 1 --no-cpu-baseline > $O/${TAG}_bench_stage1.json 2>/dev/null; tail -c 300 $O/${TAG}_bench_stage1.json
python bench.py --recipe sdf --no-cpu-baseline > $O/${TAG}_bench_sdf.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_sdf.json | head -1
python bench.py --recipe garden --no-cpu-baseline > $O/${TAG}_bench_garden.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_garden.json | head -1

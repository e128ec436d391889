#!/bin/bash
# The part of tools/collect_profiles.sh that depends on the final build but needs no counter passes: the bench command under the kernel tracer
# (stats, step / refresh timelines, events-vs-trace), the driver-flag and default bench lines, stage 1 (line + busy-time breakdown).
set -u
R=$(pwd); TAG=${1:-r04}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_traced.json 2>/tmp/ps.log
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $O/${TAG}_step_kernel_stats.csv
TR=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR > $O/${TAG}_step_timeline.txt
python $R/tools/refresh_timeline.py $TR > $O/${TAG}_refresh_timeline.txt
python $R/tools/trace_vs_events.py $TR $O/${TAG}_bench_traced.json > $O/${TAG}_trace_vs_events.txt 2>&1; cat $O/${TAG}_trace_vs_events.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -- python $R/bench.py --stage 1 --steps 80 --warmup 20 > /dev/null 2>&1
cp $(find /tmp/prof_1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_stage1_kernel_stats.csv
python $R/tools/stage1_busy.py $(find /tmp/prof_1 -name "*kernel_trace.csv" | head -1) 60 > $O/${TAG}_stage1_busy.txt; head -3 $O/${TAG}_stage1_busy.txt
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2>$O/bench.err; python tools/show_bench.py $O/${TAG}_bench_driver.json
python bench.py > $O/${TAG}_bench.json 2>>$O/bench.err; python tools/show_bench.py $O/${TAG}_bench.json | head -3
python bench.py --stage 1 > $O/${TAG}_bench_stage1.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_stage1.json | head -1

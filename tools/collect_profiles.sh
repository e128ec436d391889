#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): PMC traffic passes, kernel-trace stats, default bench lines.
# Outputs land in gpurun_out/; copy what is to be kept into profiles/.
set -u
R=$(pwd); TAG=${1:-v9}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --pretrain 200 --warmup 5 --steps 20 --no-prof --no-cpu-baseline"
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $B > /tmp/pf.log 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $R/gpurun_out/r01_pmc_traffic_$TAG.json > /tmp/pt.log 2>&1; tail -3 /tmp/pt.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --no-cpu-baseline --no-prof --pretrain 300 --warmup 20 --steps 200 > $R/gpurun_out/bench_prof_$TAG.json 2>/tmp/ps.log
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_step_kernel_stats_$TAG.csv
cd $R
python bench.py > gpurun_out/r01_bench_$TAG.json 2>gpurun_out/bench_$TAG.err; tail -c 400 gpurun_out/r01_bench_$TAG.json
python bench.py --stage 1 --no-cpu-baseline > gpurun_out/r01_bench_${TAG}_stage1.json 2>/dev/null; tail -c 300 gpurun_out/r01_bench_${TAG}_stage1.json

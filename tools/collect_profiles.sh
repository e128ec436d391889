#!/bin/bash
# Measurement set of a round (run on the GPU box from the repo root): PMC traffic + SQ passes, kernel trace of the bench command itself
# (stats, one step's timeline, events-vs-trace cross-check), default bench lines (stage 0 steady state with the CPU leg, the diffuse warm-up
# phase, stage 1, sdf, garden).  Outputs land in gpurun_out/<tag>/; copy what is to be kept into profiles/.      tools/collect_profiles.sh r04
set -u
R=$(pwd); TAG=${1:-r04}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# steady-state step (shading = full: global_step >= diffuse_step), short runs for the counter passes
B="python $R/bench.py --pretrain 1000 --warmup 5 --steps 20 --no-prof --no-cpu-baseline --no-other-configs"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s /tmp/prof_s
# counters in their own passes, with --kernel-trace only (no other trace domains)
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $B > /tmp/pf.log 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/${TAG}_pmc_traffic.json > $O/pmc_traffic.txt 2>&1; head -14 $O/pmc_traffic.txt
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pmc_s -- $B > /tmp/psq.log 2>&1
python $R/tools/pmc_sq.py /tmp/pmc_s $O/${TAG}_pmc_sq.json > $O/pmc_sq.txt 2>&1; head -14 $O/pmc_sq.txt
# the bench command itself under the tracer (per-kernel hipEvents on, as the driver runs it minus the CPU leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_traced.json 2>/tmp/ps.log
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $O/${TAG}_step_kernel_stats.csv
TR=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR > $O/${TAG}_step_timeline.txt
python $R/tools/refresh_timeline.py $TR > $O/${TAG}_refresh_timeline.txt
python $R/tools/trace_vs_events.py $TR $O/${TAG}_bench_traced.json > $O/${TAG}_trace_vs_events.txt 2>&1; cat $O/${TAG}_trace_vs_events.txt
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2>$O/bench.err; python tools/show_bench.py $O/${TAG}_bench_driver.json
python bench.py > $O/${TAG}_bench.json 2>>$O/bench.err; python tools/show_bench.py $O/${TAG}_bench.json
python bench.py --diffuse --no-cpu-baseline > $O/${TAG}_bench_diffuse.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_diffuse.json | head -1
python bench.py --stage 1 > $O/${TAG}_bench_stage1.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_stage1.json | head -1
python bench.py --recipe sdf --no-cpu-baseline > $O/${TAG}_bench_sdf.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_sdf.json | head -1
python bench.py --recipe sdf --diffuse --no-cpu-baseline > $O/${TAG}_bench_sdf_early.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_sdf_early.json | head -1
python bench.py --recipe sdf --no-cpu-baseline --autograd > $O/${TAG}_bench_sdf_autograd.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_sdf_autograd.json | head -1
python bench.py --recipe garden --no-cpu-baseline > $O/${TAG}_bench_garden.json 2>/dev/null; python tools/show_bench.py $O/${TAG}_bench_garden.json | head -1

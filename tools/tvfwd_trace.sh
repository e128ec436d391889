# the driver's command under the kernel tracer with the TV terms from the lookup (N2M_TV_FWD=1): per-kernel split of the step
R=$(pwd); O=$R/gpurun_out/r6ah; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_t
N2M_TV_FWD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/tvfwd_bench_traced.json 2>/tmp/pt.log
TR=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_vs_events.py $TR $O/tvfwd_bench_traced.json > $O/tvfwd_trace_vs_events.txt 2>&1; cat $O/tvfwd_trace_vs_events.txt
python $R/tools/step_timeline.py $TR > $O/tvfwd_step_timeline.txt; head -20 $O/tvfwd_step_timeline.txt

# stage-1 step under the kernel tracer: per-kernel busy time of 60 steps (tools/stage1_busy.py)     tools/stage1_trace.sh [tag]
TAG=${1:-r6x}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -- python $R/bench.py --stage 1 --steps 80 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/prof_1 -name "*kernel_stats.csv" | head -1) $O/stage1_kernel_stats.csv
python $R/tools/stage1_busy.py $(find /tmp/prof_1 -name "*kernel_trace.csv" | head -1) 60 > $O/stage1_busy.txt; head -50 $O/stage1_busy.txt

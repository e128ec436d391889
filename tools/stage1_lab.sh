#!/bin/bash
# stage-1 executor: tests, bench both drivers, busy-time breakdown.  gpurun -- bash tools/stage1_lab.sh
R=$(pwd); mkdir -p gpurun_out/s1; cd /tmp; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_stage1.py tests/test_raster_parity.py -x -q 2>&1 | tail -25 > gpurun_out/s1/tests.log
timeout 300 python bench.py --stage 1 --steps 200 --warmup 30 > gpurun_out/s1/bench_engine.json 2> gpurun_out/s1/bench_engine.err

cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/s1/trace -o t -- python $R/bench.py --stage 1 --steps 80 --warmup 20 > /dev/null 2>&1
cd $R
python tools/stage1_busy.py $(ls gpurun_out/s1/trace/*/*kernel_trace.csv gpurun_out/s1/trace/*kernel_trace.csv 2>/dev/null | head -1) 60 > gpurun_out/s1/busy.txt 2>&1
rm -rf gpurun_out/s1/trace
cat gpurun_out/s1/tests.log; python -c "import json;d=json.load(open(\"gpurun_out/s1/bench_engine.json\"));print(d[\"value\"],d[\"ms_per_step\"],{k:round(v[\"avg_us\"],1) for k,v in d[\"kernels\"].items()})"; tail -3 gpurun_out/s1/bench_engine.err; cat gpurun_out/s1/busy.txt

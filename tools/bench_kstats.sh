#!/bin/bash
# per-kernel averages of the default bench run (rocprofv3 --kernel-trace --stats)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bk -o t -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-prof > $R/gpurun_out/bk_bench.json 2>/dev/null
cd $R
python tools/kstats.py gpurun_out/bk 14
python -c "import json;d=json.load(open('gpurun_out/bk_bench.json'));print(d['ms_per_step'])"
rm -rf gpurun_out/bk

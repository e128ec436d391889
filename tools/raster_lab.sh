#!/bin/bash
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rl -o t -- python $R/tools/raster_lab.py 2>&1 | tail -1
cd $R
python tools/kstats.py gpurun_out/rl 6
rm -rf gpurun_out/rl

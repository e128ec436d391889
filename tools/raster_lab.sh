#!/bin/bash
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for d in 0; do
N2M_RASTER_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rl$d -o t -- python $R/tools/raster_lab.py 2>&1 | tail -1
cd $R; python tools/kstats.py gpurun_out/rl$d 6; rm -rf gpurun_out/rl$d; cd /tmp
done

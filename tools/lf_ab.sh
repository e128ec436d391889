#!/bin/bash
# A/B of the live-first sample order of the table backward (run on the GPU box from the repo root): bench.py's driver command per setting.
#   tools/lf_ab.sh [tag]      N2M_LIVE_FIRST: 0 off | 1 live-first | 2 identity order (the indirection alone);  N2M_FILL_DBG=128: TV-only path off
set -u
TAG=${1:-lf}; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_$name.json 2> $O/bench_$name.err
  echo "== $name ($*)"; python tools/show_bench.py $O/bench_$name.json 2>&1 | grep -E "samples/s|grid_encode_backward |composite|forward_packed " ; }
run off N2M_LIVE_FIRST=0
run ident N2M_LIVE_FIRST=2
run live N2M_LIVE_FIRST=1
run live_nodead N2M_LIVE_FIRST=1 N2M_FILL_DBG=128

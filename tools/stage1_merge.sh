#!/bin/bash
# Stage 1: how many levels of the colour-table backward merge same-cell runs of consecutive covered pixels (N2M_BIN_MERGE_LEVELS, default 9).
mkdir -p gpurun_out/${1:-s1m}
for m in 9 11 12 13 14 16; do
  N2M_BIN_MERGE_LEVELS=$m python bench.py --stage 1 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/${1:-s1m}/s1_m$m.json 2>/dev/null
  python - $m gpurun_out/${1:-s1m}/s1_m$m.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
print(f"merge levels {sys.argv[1]:>2s}: {d['ms_per_step']:.3f} ms/step   colour-table backward {k['grid_encode_backward']['avg_us']:.1f} us   lookup {k['grid_encode_forward']['avg_us']:.1f}")
PY
done

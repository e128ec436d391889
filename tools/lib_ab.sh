#!/bin/bash
# ABBA comparison of two builds of libn2m_hip.so on one box: bench.py's driver command, kernel times of the table backward.   tools/lib_ab.sh <variant.so> [tag]
set -u
V=$1; TAG=${2:-libab}; O=gpurun_out/$TAG; mkdir -p $O
i=0
for lib in default "$V" "$V" default default "$V"; do
  i=$((i+1))
  if [ "$lib" = default ]; then env -u N2M_HIP_LIB python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b$i.json 2>/dev/null
  else N2M_HIP_LIB=$(pwd)/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b$i.json 2>/dev/null; fi
  python - "$lib" $O/b$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
print(f"{sys.argv[1][-28:]:28s} {d['ms_per_step']:.4f} ms/step  backward {k['grid_encode_backward']['avg_us']:.1f} us  lookup {k['grid_encode_forward_packed']['avg_us']:.1f}  adam {k['adam_step']['avg_us']:.1f}")
PY
done

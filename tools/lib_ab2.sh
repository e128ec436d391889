#!/bin/bash
# ABBA over several "lib:ENV=VAL" specs on one box (bench.py driver command, table-backward kernel time).  tools/lib_ab2.sh tag spec1 spec2 ...
# spec = default | path.so[,ENV=VAL,...]
set -u
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
run() { spec=$1; i=$2
  lib=${spec%%,*}; envs=$(echo "$spec" | cut -s -d, -f2- | tr ',' ' ')
  if [ "$lib" = default ]; then e="env -u N2M_HIP_LIB $envs"; else e="env N2M_HIP_LIB=$(pwd)/$lib $envs"; fi
  $e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b$i.json 2>$O/b$i.err
  python - "$spec" $O/b$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); k = d["kernels"]
    print(f"{sys.argv[1][-44:]:44s} {d['ms_per_step']:.4f} ms/step  backward {k['grid_encode_backward']['avg_us']:.1f} us  lookup {k['grid_encode_forward_packed']['avg_us']:.1f}  adam {k['adam_step']['avg_us']:.1f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
i=0
for spec in "$@"; do i=$((i+1)); run "$spec" $i; done
for spec in $(printf '%s\n' "$@" | tac); do i=$((i+1)); run "$spec" $i; done

#!/bin/bash
# ABBA comparison of one environment switch on one box: bench.py's driver command, kernel times.   tools/env_ab.sh VAR A_VALUE B_VALUE [tag] [extra bench args]
set -u
VAR=$1; A=$2; B=$3; TAG=${4:-envab}; shift 4 || true; O=gpurun_out/$TAG; mkdir -p $O
i=0
for v in "$A" "$B" "$B" "$A" "$A" "$B"; do
  i=$((i+1))
  env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > $O/b$i.json 2>$O/b$i.err
  python - "$VAR=$v" $O/b$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
g = lambda n: k.get(n, {}).get("avg_us", float("nan"))
print(f"{sys.argv[1][-28:]:28s} {d['ms_per_step']:.4f} ms/step  backward {g('grid_encode_backward'):.1f} us  lookup {g('grid_encode_forward_packed'):.1f}  adam {g('adam_step'):.1f}  field bwd {g('mlp_backward'):.1f}", flush=True)
PY
done

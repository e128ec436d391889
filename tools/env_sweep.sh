#!/bin/bash
# One environment switch over several values on one box, forward then backward order (drift cancels):  tools/env_sweep.sh VAR "v1 v2 v3" tag [bench args]
set -u
VAR=$1; VALS=$2; TAG=${3:-sweep}; shift 3 || true; O=gpurun_out/$TAG; mkdir -p $O
REV=$(echo $VALS | tr ' ' '\n' | tac | tr '\n' ' ')
i=0
for v in $VALS $REV; do
  i=$((i+1))
  env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > $O/b$i.json 2>$O/b$i.err
  python - "$VAR=$v" $O/b$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
g = lambda n: k.get(n, {}).get("avg_us", float("nan"))
print(f"{sys.argv[1][-28:]:28s} {d['ms_per_step']:.4f} ms/step  backward {g('grid_encode_backward'):.1f} us  lookup {g('grid_encode_forward_packed'):.1f}  adam {g('adam_step'):.1f}  field bwd {g('mlp_backward'):.1f}", flush=True)
PY
done

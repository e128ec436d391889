#!/bin/bash
# Kernel-time split of the drop-in path (the reference's unchanged Python over backends/_*.py): rocprofv3 --kernel-trace --stats over
# `bench.py --dropin`, top kernels by total time + the sum per step.     bash tools/dropin_prof.sh [tag]
set -u
R=$(pwd); TAG=${1:-dropin}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dropin
STEPS=${STEPS:-48}; PRE=${PRE:-100}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dropin -- python $R/bench.py --dropin --steps $STEPS --warmup 16 --pretrain $PRE > $O/dropin.json 2> $O/dropin.err
python - $STEPS $PRE <<'PY' | tee $O/dropin_kernels.txt
import csv, glob, sys, json
steps, pre = int(sys.argv[1]), int(sys.argv[2])
f = glob.glob('/tmp/prof_dropin/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = steps + 16 + pre
print(f"all kernels: {tot/1e6:.1f} ms over {n} steps = {tot/1e3/n:.0f} us of GPU time per step")
for r in rows[:22]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>7s} avg {float(r['AverageNs'])/1e3:9.1f} us  per step {float(r['TotalDurationNs'])/1e3/n:8.1f} us")
PY
tail -c 400 $O/dropin.json

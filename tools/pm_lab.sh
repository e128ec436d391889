#!/bin/bash
# A/B of the partition-major update log against the tile-major one (run on the GPU box from the repo root): parity tests of the binned
# backward, then tools/pair_bench.py per configuration, then the per-kernel split under rocprofv3.      tools/pm_lab.sh [tag]
set -u
R=$(pwd); TAG=${1:-pm}; O=$R/gpurun_out/$TAG; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "binned or fold or pair or hot_cell or level_cap or total_variation" > $O/tests_pm1.log 2>&1; echo "tests PM=1 rc $?"; tail -3 $O/tests_pm1.log
fi
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python $R/tools/pair_bench.py --reps 40 2>&1 | grep "pair backward"; }
run N2M_BIN_PM=0
run N2M_BIN_PM=1 N2M_PM_TS=512
run N2M_BIN_PM=1 N2M_PM_TS=256
run N2M_BIN_PM=1 N2M_PM_TS=1024
run N2M_BIN_PM=1 N2M_PM_TS=512 FILL_MODE=2
run N2M_BIN_PM=1 N2M_PM_TS=512 N2M_PM_TILES=2
run N2M_BIN_PM=1 N2M_PM_TS=512 N2M_PM_TILES=8
run N2M_BIN_PM=1 N2M_PM_TS=256 N2M_PM_TILES=8
for cfg in "N2M_BIN_PM=0" "N2M_BIN_PM=1 N2M_PM_TS=512" "N2M_BIN_PM=1 N2M_PM_TS=256" ${EXTRA_PROF:+"$EXTRA_PROF"}; do
  rm -rf /tmp/prof_pm
  env $cfg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pm -- python $R/tools/pair_bench.py --reps 40 > /tmp/prof_pm.log 2>&1
  echo "== kernel stats: $cfg"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_pm/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows[:6]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done

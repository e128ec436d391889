#!/bin/bash
# Instruction-mix counters of the binned backward's kernels on tools/pair_bench.py (two PMC passes, --kernel-trace only).
set -u
R=$(pwd); cd /tmp && export TMPDIR=/tmp
pass() {
  rm -rf /tmp/pmc_x
  env N2M_BIN_PM=1 FILL_MODE=2 ${CFG:-} timeout 250 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pmc_x -- python $R/tools/pair_bench.py --reps 10 > /tmp/pmc_x.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/pmc_x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'fill_pair' in k or 'accumulate' in k:
            tag = (('fillTV' if 'ILi1E' in k else 'fill  ') if 'fill' in k else ('acc16' if 'DF16' in k else 'acc32'))
            acc[tag][r['Counter_Name']] += float(r['Counter_Value']); n[tag][r['Counter_Name']] += 1
for tag in sorted(acc):
    print(tag, '  '.join(f"{c}={acc[tag][c]/n[tag][c]:.4g}" for c in sorted(acc[tag])))
PY
}
pass "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"
pass "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pass "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"

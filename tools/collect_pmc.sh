#!/bin/bash
# The counter passes of tools/collect_profiles.sh alone (headline run only: no child configs, no long-run window).    tools/collect_pmc.sh r05
set -u
R=$(pwd); TAG=${1:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --pretrain 1000 --warmup 5 --steps 20 --no-prof --no-cpu-baseline --no-other-configs"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $B > /tmp/pf.log 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/${TAG}_pmc_traffic.json > $O/pmc_traffic.txt 2>&1; head -12 $O/pmc_traffic.txt
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pmc_s -- $B > /tmp/psq.log 2>&1
python $R/tools/pmc_sq.py /tmp/pmc_s $O/${TAG}_pmc_sq.json > $O/pmc_sq.txt 2>&1; head -12 $O/pmc_sq.txt

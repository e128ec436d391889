#!/bin/bash
# Per-level cost of the partition-major fill: each level alone (pinned to its XCD by the XCD-aware grid), fill kernel time under rocprofv3.
for l in $(seq 0 15); do
  m=$(( (l + 1) * 256 ))
  bash tools/pm_ablate.sh "FILL_MODE=$m" 2>/dev/null | sed "s/^FILL_MODE=$m */level $l /"
done

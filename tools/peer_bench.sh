#!/bin/bash
# two ranks time-sharing the one GPU of the box: collective path over gloo (host staging) against the peer-store exchange (device only)
for ps in 1; do
  N2M_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 N2M_PEER_STORE=$ps timeout 600 python bench.py --gpus 2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('peer_store=$ps', d['n_gpus'], 'ranks:', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e6,1), 'M samples/s', d['config'].get('parallelism'))"
done

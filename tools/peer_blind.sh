#!/bin/bash
# The peer-store exchange with a blind rank under several settings (debug aid): which one times out?
for cfg in "N2M_SHARD_REFRESH=0" "N2M_SHARD_REFRESH=1" "N2M_SHARD_REFRESH=1 N2M_PEER_TIMEOUT_MS=60000"; do
  echo "== $cfg"
  ( time env $cfg N2M_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 N2M_SHARD_ADAM=1 N2M_PEER_STORE=1 N2M_DIST_BLIND_RANK=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 tools/dist_check.py 24 engine 2>&1 | grep -E "DIST_CHECK|RuntimeError" | head -3 ) 2>&1 | grep -E "DIST_CHECK|RuntimeError|real"
done

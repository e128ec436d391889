#!/bin/bash
# FIRST RUN ON A MULTI-GPU MI355X NODE (no such node was available in rounds 1-6: every number of DESIGN.md section 6 is an estimate).
# One command, one table: the RCCL tests that skip on a one-GPU box, then the weak-scaling bench at 1 / 2 / 4 / 8 GPUs in both exchange
# modes -- the default (RCCL reduce-scatter / all-gather, sharded Adam, sharded occupancy refresh) and the opt-in peer-store exchange
# (N2M_PEER_STORE=1: no collective in the step) -- plus the all-reduce layout (N2M_SHARD_ADAM=0) and the replicated refresh
# (N2M_SHARD_REFRESH=0) as A/B points.  Every bench invocation ends in ONE JSON line (a watchdog turns a hang into an `error` line).
#     tools/first_multigpu.sh [output dir]
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/first_multigpu}; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $N" | tee "$O/summary.txt"
python -m pytest tests/test_parallel_gpu.py -q -m gpu -k rccl 2>&1 | tail -4 | tee -a "$O/summary.txt"
B="--steps 40 --warmup 10 --no-cpu-baseline --no-other-configs"
python bench.py --gpus 1 $B > "$O/bench_1_single.json" 2> "$O/bench_1_single.err"
for W in 2 4 8; do
  [ "$W" -le "$N" ] || continue
  timeout 1800 python bench.py --gpus $W $B > "$O/bench_${W}_default.json" 2> "$O/bench_${W}_default.err"
  N2M_PEER_STORE=1 timeout 1800 python bench.py --gpus $W $B > "$O/bench_${W}_peer.json" 2> "$O/bench_${W}_peer.err"
  N2M_SHARD_ADAM=0 timeout 1800 python bench.py --gpus $W $B > "$O/bench_${W}_allreduce.json" 2> "$O/bench_${W}_allreduce.err"
  N2M_SHARD_REFRESH=0 timeout 1800 python bench.py --gpus $W $B > "$O/bench_${W}_replrefresh.json" 2> "$O/bench_${W}_replrefresh.err"
done
python - "$O" <<'PY' | tee -a "$O/summary.txt"
import glob, json, os, sys
o = sys.argv[1]
rows, base = [], None
for f in sorted(glob.glob(os.path.join(o, "bench_*.json")), key=lambda p: (int(os.path.basename(p).split("_")[1]), p)):
    _, n, mode = os.path.basename(f)[:-5].split("_", 2)
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        rows.append((int(n), mode, None, None, f"no JSON line ({e!r}): see {f[:-5]}.err")); continue
    if d.get("value") is None:
        rows.append((int(n), mode, None, None, "ERROR: " + str(d.get("error"))[:200])); continue
    if int(n) == 1:
        base = d["value"]
    rows.append((int(n), mode, d["ms_per_step"], d["value"], (d.get("config") or {}).get("parallelism", "")[:110]))
print(f"{'GPUs':>4} {'mode':12s} {'ms/step':>8} {'samples/s':>12} {'value / (N x value(1))':>22}  exchange")
for n, mode, ms, v, note in rows:
    if v is None:
        print(f"{n:4d} {mode:12s} {'-':>8} {'-':>12} {'-':>22}  {note}")
    else:
        print(f"{n:4d} {mode:12s} {ms:8.4f} {v:12.4g} {(v / (n * base) if base else float('nan')):22.3f}  {note}")
PY

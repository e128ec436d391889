#!/bin/bash
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ip -o t -- python $R/tools/infer_bench.py --pretrain 600 --frames 10 2>&1 | tail -1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ip/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: kernels after the last batch of pixel-ray generation... take the last 12 ms of the trace before the eval_psnr tail
import collections
t_end = int(rows[-1]["End_Timestamp"])
# find frames by the near_far kernel
nf = [i for i, r in enumerate(rows) if "near_far" in r["Kernel_Name"]]
print("near_far launches:", len(nf))
a, b = nf[-3], nf[-2]
seg = rows[a:b]
busy = collections.Counter(); cnt = collections.Counter()
for r in seg:
    busy[r["Kernel_Name"][:70]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[r["Kernel_Name"][:70]] += 1
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"one frame: wall {wall:.0f} us, busy {sum(busy.values()):.0f} us, {len(seg)} launches")
for k, v in busy.most_common(14):
    print(f"  {v:8.1f} us x{cnt[k]:4d}  {k}")
PY
rm -rf gpurun_out/ip

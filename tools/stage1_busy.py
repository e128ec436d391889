"""GPU busy time per stage-1 step from a rocprofv3 --kernel-trace CSV: sum of kernel durations between consecutive stage1_head launches
over the last steps of the run, against the wall time between them.   python tools/stage1_busy.py <kernel_trace.csv> [steps]"""
import csv, sys, collections
HDR = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp",
       "End_Timestamp"]
rows = []
for r in csv.reader(open(sys.argv[1])):
    if len(r) < len(HDR) or not r[9].isdigit():
        continue
    d = dict(zip(HDR, r))
    rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"]))
rows.sort()
heads = [i for i, r in enumerate(rows) if "stage1_head_kernel" in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
heads = heads[-(n + 1):]
busy = collections.Counter()
cnt = collections.Counter()
for a, b in zip(heads[:-1], heads[1:]):
    for r in rows[a + 1:b + 1]:
        busy[r[2][:70]] += (r[1] - r[0]) / 1e3
        cnt[r[2][:70]] += 1
steps = len(heads) - 1
wall = (rows[heads[-1]][1] - rows[heads[0]][1]) / 1e3 / steps
tot = sum(busy.values()) / steps
print(f"{steps} steps: wall {wall:.1f} us/step, GPU busy {tot:.1f} us/step ({100 * tot / wall:.0f} %), {sum(cnt.values()) / steps:.0f} launches/step")
for k, v in busy.most_common(28):
    print(f"  {v / steps:8.1f} us  x{cnt[k] / steps:5.1f}  {k}")

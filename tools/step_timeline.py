"""Print the kernel timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV (tail without header is fine):
python tools/step_timeline.py <kernel_trace.csv> [offset from the step at 85 % of the run]"""
import csv, re, sys
HDR = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp",
       "End_Timestamp"]
rows = []
for r in csv.reader(open(sys.argv[1])):
    if len(r) < len(HDR) or not r[9].isdigit():
        continue
    d = dict(zip(HDR, r))
    rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Queue_Id"], d["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"_ZN12_GLOBAL__N_1\d+", "", re.sub(r"\(anonymous namespace\)::", "", n))[:52]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
k = int(len(adam) * 0.85) + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)      # inside bench.py's timed region (after pre-training and warm-up)
a, b = adam[k], adam[k + 1]
t0 = rows[a][1]
walls = [(rows[adam[i + 1]][1] - rows[adam[i]][1]) / 1e3 for i in range(len(adam) - 1)]
print(f"{len(rows)} kernels, {len(adam)} steps; step wall us: median {sorted(walls)[len(walls)//2]:.1f}  this one {(rows[b][1]-rows[a][1])/1e3:.1f}")
for r in rows[a + 1:b + 1]:
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[2]}  {short(r[3])}")

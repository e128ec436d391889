"""Where an occupancy-refresh step's time goes: from a rocprofv3 --kernel-trace CSV print the step-wall statistics (all steps / refresh
steps / the two steps behind a refresh) and the kernel timeline of one refresh step inside the timed region.
python tools/refresh_timeline.py <kernel_trace.csv>"""
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
short = lambda n: re.sub(r"_ZN12_GLOBAL__N_1\d+", "", re.sub(r"\(anonymous namespace\)::", "", n))[:60]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
lo = int(len(adam) * 0.84)                       # bench.py's timed region: the last ~16 % of the steps
steps = []
for i in range(lo, len(adam) - 1):
    a, b = adam[i], adam[i + 1]
    names = [rows[j][3] for j in range(a + 1, b + 1)]
    steps.append((i, (rows[b][1] - rows[a][1]) / 1e3, any("packbits" in n for n in names)))
mean = lambda xs: sum(xs) / max(len(xs), 1)
allw = [w for _, w, _ in steps]
ref = [w for _, w, r in steps if r]
after = [steps[k + 1][1] for k in range(len(steps) - 1) if steps[k][2]]
plain = [w for k, (_, w, r) in enumerate(steps) if not r and not (k > 0 and steps[k - 1][2])]
print(f"{len(steps)} steps in the window: mean wall {mean(allw):.1f} us, median {sorted(allw)[len(allw)//2]:.1f}")
print(f"  plain steps {len(plain)}: mean {mean(plain):.1f};  refresh steps {len(ref)}: mean {mean(ref):.1f};  the step behind a refresh: mean {mean(after):.1f}")
print(f"  refresh overhead per 16 steps: {mean(ref) + mean(after) - 2 * mean(plain):.1f} us = {(mean(ref) + mean(after) - 2 * mean(plain)) / 16:.1f} us per step")
k = [i for i, _, r in steps if r][len(ref) // 2]
a, b = adam[k], adam[k + 2]
t0 = rows[a][1]
for r in rows[a + 1:b + 1]:
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[2]}  {short(r[3])}")

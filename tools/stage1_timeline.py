"""Kernel timeline of one stage-1 step (between two stage1_head launches) from a rocprofv3 --kernel-trace CSV.
python tools/stage1_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.reader(open(sys.argv[1])):
    if len(r) >= 11 and r[9].isdigit():
        rows.append((int(r[9]), int(r[10]), r[7]))
rows.sort()
heads = [i for i, r in enumerate(rows) if "stage1_head_kernel" in r[2]]
a, b = heads[-2], heads[-1]
t0 = rows[a][1]
print(f"{b - a} launches, {(rows[b][1] - rows[a][1]) / 1e3:.1f} us wall (under the tracer), {sum(r[1] - r[0] for r in rows[a + 1:b + 1]) / 1e3:.1f} us busy")
for r in rows[a + 1:b + 1]:
    n = re.sub(r"at::native::|\(anonymous namespace\)::", "", r[2])
    print(f"{(r[0] - t0) / 1e3:8.1f} {(r[1] - r[0]) / 1e3:7.1f}  {n[:140]}")

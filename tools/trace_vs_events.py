"""Cross-check of bench.py's hipEvent timings against a rocprofv3 kernel trace of THE SAME run:
   rocprofv3 --kernel-trace --output-format csv -d D -- python bench.py --no-cpu-baseline --steps K ... > line.json
   python tools/trace_vs_events.py <kernel_trace.csv> line.json
For every library entry point of the bench line: mean duration by events vs the mean, over the last K steps of the trace, of the summed
kernel durations (and of first-start -> last-end, which is what an event pair around the launches can at best see)."""
import csv, json, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
HDR = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp",
       "End_Timestamp"]
rows = []
for r in csv.reader(open(sys.argv[1])):
    if len(r) < len(HDR) or not r[9].isdigit():
        continue
    rows.append((int(r[9]), int(r[10]), r[7]))
rows.sort()
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
K = int(line["steps"])
GROUPS = {"grid_encode_backward": ("pm_fill_pair_kernel", "pm_accumulate_both_kernel", "pm_accumulate_kernel", "bin_fill_pair_kernel", "bin_accumulate_kernel"), "grid_encode_forward_packed": ("grid_forward3_packed_kernel",),
          "mlp_backward": ("field_backward",), "adam_step": ("adam_kernel",), "mlp_forward": ("field_forward_kernel",),
          "composite_rays_train_forward": ("composite_loss_train_kernel",)}
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
assert len(adam) > K, "trace shorter than the timed region"
print(f"{len(adam)} steps in the trace, comparing the last {K} (a step = the kernels between two Adam launches)")
steps = [rows[adam[-K - 1 + i] + 1: adam[-K + i] + 1] for i in range(K)]
for name, pats in GROUPS.items():
    sums, spans, counts, parts = [], [], [], {}
    for st in steps:
        ks = [r for r in st if any(p in r[2] for p in pats)]
        if name != "grid_encode_backward" and len(ks) > 1:          # refresh steps launch the same kernels on other data: keep the step's own
            ks = [max(ks, key=lambda r: r[1] - r[0])]
        if not ks:
            continue
        sums.append(sum(r[1] - r[0] for r in ks) / 1e3)
        spans.append((ks[-1][1] - ks[0][0]) / 1e3)
        counts.append(len(ks))
        for r in ks:
            key = r[2].split("(")[0][-48:]
            parts.setdefault(key, []).append((r[1] - r[0]) / 1e3)
    ev = line["kernels"].get(name, {}).get("avg_us")
    m = lambda v: sum(v) / max(len(v), 1)
    print(f"{name:30s} events {ev if ev is None else round(ev, 1):>8} us | trace: {m(counts):.2f} kernels/step, sum {m(sums):7.1f} us, first start -> last end {m(spans):7.1f} us")
    if len(parts) > 1:
        for k, v in parts.items():
            print(f"{'':34s}{k:48s} {m(v):7.1f} us")

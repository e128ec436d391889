#!/usr/bin/env python3
"""Pretty-print the JSON line(s) bench.py wrote to a log."""
import json, sys
for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        c = d.get("config", {})
        print(f"{path}: {d['value']/1e6:.2f} M {d.get('unit', 'samples/s')}  {d.get('rays_per_sec', 0)/1e3:.1f} k rays/s  {d['ms_per_step']:.3f} ms/step  "
              f"rays/step={c.get('rays_per_step_per_gpu', 0):.0f} samples/step={c.get('samples_per_step_per_gpu', 0):.0f} psnr={d.get('psnr_view0_quarter_res')}")
        for k, v in (d.get("kernels") or {}).items():
            print(f"   {k:32s} n={v['launches']:5d} avg_us={v['avg_us']:9.1f} ms/step={v['ms_per_step']:.3f} GB/s={v['GBps'] or 0:8.0f}")
        if d.get("roofline"):
            r = d["roofline"]
            print(f"   roofline: {r['kernel']} {r['achieved']:.0f} GB/s = {100*r['frac']:.1f}% of {r['peak']:.0f}")
        if d.get("roofline_lookup"):
            r = d["roofline_lookup"]
            print(f"   lookup:   {r['kernel']} {r['achieved']:.0f} GB/s = {100*r['frac']:.1f}% of {r['peak']:.0f}  ({r['avg_us']:.1f} us)")
        if d.get("cpu_baseline"):
            print("   cpu:", d["cpu_baseline"])

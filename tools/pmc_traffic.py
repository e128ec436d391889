#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic per launch.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [min_calls]

FETCH_SIZE / WRITE_SIZE are reported in KiB of memory-side L2 requests; on gfx950 FETCH_SIZE counts a 128-byte read request
as 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is taken as is (uncalibrated there).  Infinity-cache
hits are included in both, i.e. the figures are an upper bound on true HBM traffic."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].strip()[:200]
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    min_calls = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    F, W = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        nf, sf = F.get(k, (0, 0.0))
        nw, sw = W.get(k, (0, 0.0))
        if max(nf, nw) < min_calls:
            continue
        res[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                  "fetch_bytes_per_launch": 2.0 * 1024.0 * sf / nf if nf else None,     # x2: gfx950 correction
                  "write_bytes_per_launch": 1024.0 * sw / nw if nw else None}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -((kv[1]["fetch_bytes_per_launch"] or 0) + (kv[1]["write_bytes_per_launch"] or 0)))[:25]:
        print(f"{(v['fetch_bytes_per_launch'] or 0)/1e6:10.1f} MB rd {(v['write_bytes_per_launch'] or 0)/1e6:10.1f} MB wr  x{v['launches_fetch_pass']:5d}  {k[:110]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from one rocprofv3 --pmc pass (counter_collection CSV):

    python tools/pmc_sq.py <dir> <out.json> [min_calls]

Reports, per kernel, the counters averaged over its launches and the fractions of SQ_WAVE_CYCLES that waves spent issuing VALU /
LDS / VMEM instructions, waiting (SQ_WAIT_ANY: parked on s_waitcnt / barrier; SQ_WAIT_INST_ANY: issue stall) and with the MFMA
pipe busy -- all SQ_* cycle counters are in quad-cycles per the guide, SQ_VALU_MFMA_BUSY_CYCLES in cycles."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    min_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].strip()[:160]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
    res = {}
    for k, cs in acc.items():
        n = max(calls[k].values())
        if n < min_calls:
            continue
        avg = {c: v / calls[k][c] for c, v in cs.items()}
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        row = {"launches": n, **{c: round(v, 1) for c, v in avg.items()}}
        if wc > 0:
            for c, name in (("SQ_ACTIVE_INST_VALU", "valu_frac"), ("SQ_ACTIVE_INST_LDS", "lds_frac"), ("SQ_ACTIVE_INST_VMEM", "vmem_frac"),
                            ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac")):
                if c in avg:
                    row[name] = round(avg[c] / wc, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                row["mfma_busy_frac_of_wave_cycles"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)
        res[k] = row
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    keys = ("valu_frac", "lds_frac", "vmem_frac", "wait_any_frac", "wait_inst_frac", "mfma_busy_frac_of_wave_cycles")
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
        print(" ".join(f"{v.get(x, float('nan')):6.3f}" for x in keys), f"x{v['launches']:4d}", k[:90])
    print("columns:", " ".join(keys))


if __name__ == "__main__":
    main()

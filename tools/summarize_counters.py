#!/usr/bin/env python
"""Per-launch counter table of rocprofv3 --pmc passes (one directory per pass under <root>/p*): rows = counters, columns = the launches
of the kernels whose short name contains the filter (default: render_forward_kernel<9, false, false>), in dispatch order.

    python tools/summarize_counters.py gpurun_out/cnt_frame ["kernel substring"]"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402

csv.field_size_limit(1 << 30)


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "render_forward_kernel<9, false, false>"
    rows = {}
    for p in sorted(glob.glob(os.path.join(root, "p*"))):
        if not os.path.isdir(p):
            continue
        per = defaultdict(dict)  # counter -> dispatch id -> value (summed over dimensions / instances)
        for path in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if filt not in short(r["Kernel_Name"]):
                    continue
                d = int(r["Dispatch_Id"])
                per[r["Counter_Name"]][d] = per[r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
        for c, v in per.items():
            rows[c] = [v[k] for k in sorted(v)]
    n = max((len(v) for v in rows.values()), default=0)
    print(f"kernel filter: `{filt}`; one column per launch in dispatch order\n")
    print("| counter | " + " | ".join(f"#{i}" for i in range(n)) + " |")
    print("|---|" + "---:|" * n)
    for c in sorted(rows):
        print(f"| {c} | " + " | ".join(f"{x:.4g}" for x in rows[c]) + " |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (average per launch).

    python tools/summarize_pmc.py gpurun_out/prof2_fetch gpurun_out/prof2_write [...]  > profiles/xyz.md

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B fabric
read request for wide coalesced streams, i.e. HALF the bytes (MI355X_MICROARCH.md, HBM section); the table shows
the raw counter and the corrected value (x2) side by side.  WRITE_SIZE is shown raw (uncalibrated)."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402

csv.field_size_limit(1 << 30)


def main():
    agg = defaultdict(lambda: defaultdict(list))
    for root in sys.argv[1:]:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in agg.values() for c in k})
    print("| kernel | launches | " + " | ".join(f"{c} avg/launch" for c in counters) + " |")
    print("|---|---:|" + "---:|" * len(counters))
    keep = [k for k in agg if k.startswith(("render_", "brick_", "scatter_", "expand_", "bin_")) or k in ("adam_kernel",)]
    for k in sorted(keep):
        n = max(len(v) for v in agg[k].values())
        cells = []
        for c in counters:
            v = agg[k].get(c)
            cells.append(f"{sum(v) / len(v):.1f}" if v else "-")
        print(f"| `{k}` | {n} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()

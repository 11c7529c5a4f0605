#!/usr/bin/env python
"""Table of tools/write_size_calibration.sh: counter value per launch against the bytes the launch is known to write."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
root = sys.argv[1]
known = {}
for path in glob.glob(os.path.join(root, "*.out")):
    for line in open(path):
        m = re.match(r"(\S.*),(\d+)$", line.strip())
        if m and m.group(1) != "kernel":
            known[m.group(1)] = int(m.group(2))
vals = defaultdict(dict)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)
        vals[name][r["Counter_Name"]] = vals[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
counters = sorted({c for v in vals.values() for c in v})
print("| launch | bytes written (known) | " + " | ".join(counters) + " | WRITE_SIZE KiB x 1024 / known |")
print("|---|---:|" + "---:|" * (len(counters) + 1))
for name in known:
    v = vals.get(name, {})
    ratio = v.get("WRITE_SIZE", float("nan")) * 1024.0 / known[name] if "WRITE_SIZE" in v else float("nan")
    print(f"| `{name}` | {known[name]} | " + " | ".join(f"{v.get(c, float('nan')):.6g}" for c in counters) + f" | {ratio:.3f} |")

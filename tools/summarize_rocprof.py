#!/usr/bin/env python
"""Condense a rocprofv3 result (--kernel-trace --stats, sqlite 'rocpd' database or *_kernel_stats.csv)
into a short markdown table with readable kernel names.

    python tools/summarize_rocprof.py gpurun_out/prof1 > profiles/r01_bench_kernel_stats.md
"""
import csv
import glob
import os
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?((?:render_(?:forward|backward|emit_direct|frame_tile)|brick_gather|brick_accumulate|scatter_records|expand_records)_kernel<[^>]*>)", name)
    if m:
        return m.group(1)
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", name)
    base = m.group(1) if m else name
    for key in ("onesweep_iteration", "onesweep_histograms", "randperm_handle_duplicate", "FillFunctor", "uniform_kernel",
                "random_from_to", "MeanOps", "CUDAFunctor_add", "MulFunctor", "compare_scalar", "BitwiseAnd", "sum_functor",
                "CatArrayBatchedCopy", "direct_copy", "arange", "sign_kernel", "index_elementwise", "gather", "abs_kernel"):
        if key in name:
            return f"torch:{key}"
    return base[:70]


def rows_from_db(path):
    con = sqlite3.connect(path)
    cur = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    return [(short(n), int(c), float(t), float(a), float(p)) for n, c, t, a, p in cur]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    root = sys.argv[1]
    dbs = glob.glob(os.path.join(root, "**", "*.db"), recursive=True)
    csvs = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
    rows = rows_from_db(dbs[0]) if dbs else rows_from_csv(csvs[0])
    merged = {}
    for n, c, t, a, p in rows:
        m = merged.setdefault(n, [0, 0.0, 0.0])
        m[0] += c
        m[1] += t
        m[2] += p
    print(f"source: rocprofv3 --kernel-trace --stats ({os.path.basename(dbs[0] if dbs else csvs[0])}); durations in microseconds\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for n, (c, t, p) in sorted(merged.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{n}` | {c} | {t:.1f} | {t / c:.1f} | {p:.2f} |")


if __name__ == "__main__":
    main()

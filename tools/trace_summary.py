#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid, block): calls, avg/min duration, share."""
import collections
import csv
import glob
import re
import sys


sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from symbols import short  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*/*_kernel_trace.csv"))[-1]
    rows = list(csv.DictReader(open(path)))
    groups = collections.defaultdict(list)
    for r in rows:
        key = (short(r["Kernel_Name"]), f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}', r["Workgroup_Size_X"])
        groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in groups.values())
    print(f"{'kernel':58s} {'grid':>16s} {'wg':>4s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'share':>6s} {'p10':>7s} {'p50':>7s} {'p90':>7s}")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        q = sorted(v)
        pct = [q[min(len(q) - 1, int(f * len(q)))] for f in (0.1, 0.5, 0.9)]
        print(f"{key[0]:58s} {key[1]:>16s} {key[2]:>4s} {len(v):6d} {sum(v) / len(v):8.2f} {min(v):8.2f} {100 * sum(v) / total:5.1f}%"
              f" {pct[0]:7.2f} {pct[1]:7.2f} {pct[2]:7.2f}")


if __name__ == "__main__":
    main()

"""Per-dispatch means of every counter in a set of rocprofv3 --pmc passes (counter_collection.csv files), one row per kernel symbol.

  python tools/attn_pmc_table.py <dir with pass sub-directories> [--match attention]
"""
import argparse
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from symbols import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--match", default="attention")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if a.match in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"  {c:32s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Both counters are in KiB per dispatch; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so it is doubled.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys


def symbol(name):
    m = re.match(r"_ZN5vitvs\d+(\w+?)I", name)
    if m:
        epi = re.search(r"(EpiStore|EpiPartial|EpiPatch|EpiResidual)", name)
        kind = "bf16" if "DF16b" in name else "f32"
        dims = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)E", name)
        return m.group(1) + f"<{kind}" + ("," + ",".join(dims.groups()) if dims else "") + ">" + (":" + epi.group(1) if epi else "")
    return re.sub(r"vitvs::", "", name).split("(")[0].replace("void ", "")


def load(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[symbol(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        out[k] = dict(dispatches=len(fetch.get(k, [])), fetch_size_kib_avg=round(f, 1), write_size_kib_avg=round(w, 1),
                      hbm_bytes_per_launch=int((2 * f + w) * 1024),
                      note="FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as reported")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

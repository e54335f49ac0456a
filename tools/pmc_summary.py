#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Both counters are in KiB per dispatch; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so it is doubled.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [<mfma counter_collection.csv>] \
        > profiles/rNN_pmc_traffic.json

The optional third file is a `--pmc SQ_VALU_MFMA_BUSY_CYCLES` pass: busy cycles per launch summed over the chip's 1024
SIMDs (16 per v_mfma_f32_16x16x32_bf16); utilisation = that / (1024 x launch duration x 2.4 GHz), the duration taken from
the timing pass (counter-collection runs serialise and slow the kernels, their own timestamps are not used).
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


def load(path, counter=None):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if counter is None or r["Counter_Name"] == counter:
            acc[symbol(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    mfma = load(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES") if len(sys.argv) > 3 else {}
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        out[k] = dict(dispatches=len(fetch.get(k, [])), fetch_size_kib_avg=round(f, 1), write_size_kib_avg=round(w, 1),
                      hbm_bytes_per_launch=int((2 * f + w) * 1024),
                      note="FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as reported")
        if k in mfma:
            out[k]["mfma_busy_cycles_per_launch"] = int(sum(mfma[k]) / len(mfma[k]))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Both counters are in KiB per dispatch; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so it is doubled.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [<mfma counter_collection.csv>] \
        [--commit HASH] [--command "..."] > profiles/rNN_pmc_traffic.json

The optional third file is a `--pmc SQ_VALU_MFMA_BUSY_CYCLES` pass: busy cycles per launch summed over the chip's 1024
SIMDs (16 per v_mfma_f32_16x16x32_bf16); utilisation = that / (1024 x launch duration x clock), the duration taken from
the timing pass (counter-collection runs serialise and slow the kernels, their own timestamps are not used).
The output records which commit of the kernels and which command the counters were collected on: bench.py quotes
`traffic` from this file and says so.
"""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from symbols import short  # noqa: E402


def load(path, counter=None):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if counter is None or r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("mfma", nargs="?")
    ap.add_argument("--commit", default=None)
    ap.add_argument("--command", default=None)
    a = ap.parse_args()
    fetch, write = load(a.fetch), load(a.write)
    mfma = load(a.mfma, "SQ_VALU_MFMA_BUSY_CYCLES") if a.mfma else {}
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        out[k] = dict(dispatches=len(fetch.get(k, [])), fetch_size_kib_avg=round(f, 1), write_size_kib_avg=round(w, 1),
                      hbm_bytes_per_launch=int((2 * f + w) * 1024))
        if k in mfma:
            out[k]["mfma_busy_cycles_per_launch"] = int(sum(mfma[k]) / len(mfma[k]))
    json.dump(dict(git_commit=a.commit, command=a.command,
                   note="FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as reported; separate --pmc passes",
                   kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How do the launches of several queues share the chip?  Reads a rocprofv3 kernel-trace CSV of a run with several updates
in flight and prints, for the busiest window of the trace:
  * per queue: launches, sum of kernel durations, sum of the gaps between consecutive launches of that queue;
  * how many kernels are running at once (share of the window with 0, 1, 2, ... kernels in flight);
  * per kernel symbol: duration when alone vs when overlapping another queue's kernel.
usage: tools/overlap_timeline.py <kernel_trace.csv> [window_ms]
"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from symbols import short  # noqa: E402


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    # the window with the most launches
    w = int(win_ms * 1e6)
    best, lo = (0, 0), 0
    for hi in range(len(ev)):
        while ev[hi][0] - ev[lo][0] > w:
            lo += 1
        if hi - lo + 1 > best[0]:
            best = (hi - lo + 1, lo)
    n, lo = best
    sel = ev[lo:lo + n]
    t0, t1 = sel[0][0], max(e[1] for e in sel)
    span = (t1 - t0) / 1e3
    print(f"window: {n} launches in {span:.1f} us ({span / n:.2f} us per launch overall)")
    byq = collections.defaultdict(list)
    for e in sel:
        byq[e[2]].append(e)
    for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        dur = sum(b - a for a, b, _, _ in v) / 1e3
        gaps = [max(0, v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        gs = sorted(gaps) or [0.0]
        print(f"queue {q:>4s}: {len(v):5d} launches, kernels {dur:9.1f} us, gaps {sum(gaps):9.1f} us "
              f"(median {gs[len(gs) // 2]:.2f}, p90 {gs[int(0.9 * (len(gs) - 1))]:.2f})")
    # concurrency histogram
    pts = []
    for a, b, _, _ in sel:
        pts.append((a, 1))
        pts.append((b, -1))
    pts.sort()
    level, last, hist = 0, t0, collections.Counter()
    for t, d in pts:
        hist[level] += t - last
        last = t
        level += d
    tot = sum(hist.values())
    print("kernels running at once: " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(hist.items())))
    # duration alone vs overlapped (by whether another queue's kernel intersects more than half of it)
    alone, shared = collections.defaultdict(list), collections.defaultdict(list)
    for i, (a, b, q, name) in enumerate(sel):
        ov = 0
        for j in range(max(0, i - 12), min(len(sel), i + 13)):
            if j == i or sel[j][2] == q:
                continue
            ov += max(0, min(b, sel[j][1]) - max(a, sel[j][0]))
        (shared if ov > 0.5 * (b - a) else alone)[name].append((b - a) / 1e3)
    print(f"{'kernel':58s} {'alone n':>8s} {'avg us':>8s} {'overlapped n':>13s} {'avg us':>8s}")
    for name in sorted(set(alone) | set(shared), key=lambda k: -(sum(alone[k]) + sum(shared[k]))):
        a, s = alone[name], shared[name]
        print(f"{name:58s} {len(a):8d} {sum(a) / max(len(a), 1):8.2f} {len(s):13d} {sum(s) / max(len(s), 1):8.2f}")


if __name__ == "__main__":
    main()

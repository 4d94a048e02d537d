#!/usr/bin/env python
"""Timeline statistics of a rocprofv3 --kernel-trace CSV (GPU box): how much of the wall time of the traced region has 0 / 1 / 2+
kernels executing, per-queue busy time, and the gaps between consecutive kernels of the busiest queue.
    python tools/trace_timeline.py <kernel_trace.csv> [--last-frac 0.5]"""
import csv
import sys
import collections


def main():
    path = sys.argv[1]
    last = float(sys.argv[sys.argv.index("--last-frac") + 1]) if "--last-frac" in sys.argv else 0.5
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * last                       # the steady-state tail (warmup, graph building in front)
    rows = [r for r in rows if r[0] >= cut]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, q, n in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, prev, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(depth, 3)] += t - prev
        prev = t
        depth += d
    tot = t1 - t0
    print("traced tail: %.3f ms, %d kernels" % (tot / 1e6, len(rows)))
    for k in sorted(hist):
        print("  %s kernels executing: %6.2f %%" % (("3+" if k == 3 else str(k)), 100.0 * hist[k] / tot))
    byq = collections.defaultdict(list)
    for s, e, q, n in rows:
        byq[q].append((s, e, n))
    for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in lst)
        gaps = [b[0] - a[1] for a, b in zip(lst, lst[1:]) if b[0] > a[1]]
        small = [g for g in gaps if g < 50000]
        print("queue %s: %5d kernels, busy %6.2f %% of the tail, %d gaps < 50 us: total %.3f ms, median %.1f us" %
              (q, len(lst), 100.0 * busy / tot, len(small), sum(small) / 1e6, (sorted(small)[len(small) // 2] / 1e3 if small else 0)))
    # time by kernel name when it is the ONLY kernel executing (critical-path proxy)
    alone = collections.Counter()
    active = {}
    ev2 = []
    for i, (s, e, q, n) in enumerate(rows):
        ev2.append((s, 1, i))
        ev2.append((e, -1, i))
    ev2.sort()
    prev = t0
    for t, d, i in ev2:
        if len(active) == 1:
            alone[rows[next(iter(active))][3].split("(")[0][:60]] += t - prev
        prev = t
        if d == 1:
            active[i] = 1
        else:
            active.pop(i, None)
    print("time as the only executing kernel (top 25):")
    for n, v in alone.most_common(25):
        print("  %8.3f ms  %s" % (v / 1e6, n))


if __name__ == "__main__":
    main()

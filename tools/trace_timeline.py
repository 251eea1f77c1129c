#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py (several streams): for the last complete step, per stream the
busy time, and over the step how long 0, 1, 2, ... kernels were running at once and which kernel classes ran alone.

    python tools/trace_timeline.py gpurun_out/<dir>/<w>/kt/*/*_kernel_trace.csv [steps]
"""
import collections
import csv
import re
import sys


def short(n):
    m = re.match(r"void (k_\w+)<([^>]*)>", n)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return n.split("(")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), short(r["Kernel_Name"])) for r in rows]
    ks.sort()
    # steps are delimited by k_init launches of the first stream that has them
    inits = [k for k in ks if k[3] == "k_init"]
    by_stream = collections.defaultdict(list)
    for k in inits:
        by_stream[k[2]].append(k[0])
    s0 = min(by_stream)
    starts = by_stream[s0]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    t0, t1 = starts[-nsteps - 1], starts[-1]
    win = [k for k in ks if k[0] >= t0 and k[1] <= t1 + 5_000_000 and k[0] < t1]
    span = (t1 - t0) / 1e6
    print("window: %d steps, %.3f ms per step, %d launches per step" % (nsteps, span / nsteps, len(win) / nsteps))
    busy = collections.defaultdict(float)
    for a, b, s, n in win:
        busy[s] += (b - a) / 1e6
    print("per stream busy ms per step:", {s: round(v / nsteps, 2) for s, v in sorted(busy.items())})
    ev = []
    for a, b, s, n in win:
        ev.append((a, 1, n)); ev.append((b, -1, n))
    ev.sort()
    conc = collections.Counter()
    alone = collections.Counter()
    cur = collections.Counter()
    last = t0
    for t, d, n in ev:
        k = sum(cur.values())
        conc[k] += (t - last)
        if k == 1:
            alone[next(iter(+cur))] += (t - last)
        if k == 2:
            alone[" + ".join(sorted(+cur))] += 0  # placeholder: keep the table small
        last = t
        cur[n] += d
    print("kernels running at once -> ms per step:", {k: round(v / 1e6 / nsteps, 3) for k, v in sorted(conc.items())})
    print("running ALONE, ms per step:")
    for n, v in alone.most_common(12):
        if v:
            print("   %-60s %.3f" % (n, v / 1e6 / nsteps))
    tot = collections.Counter()
    for a, b, s, n in win:
        tot[n] += b - a
    print("kernel ms per step summed over streams (top 14):")
    for n, v in tot.most_common(14):
        print("   %-60s %.3f" % (n, v / 1e6 / nsteps))


if __name__ == "__main__":
    main()

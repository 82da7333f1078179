#!/usr/bin/env python3
"""Where is the GPU idle, and what is the host doing meanwhile?

Lays the host-side scopes of one run (VMN_TRACE_EVENTS=file, csrc/hosttrace.h: label,t0_ns,t1_ns,thread on
CLOCK_MONOTONIC) over the kernel trace of the same run (rocprofv3 --kernel-trace --output-format csv: Start_Timestamp /
End_Timestamp, the same clock) and attributes every nanosecond in which no kernel runs to the innermost host scope of the
calling thread that is open at that moment.  Used for the small-N work of DESIGN.md §6 (at N = 10^4 a proof is bound by
what happens BETWEEN the kernels).

usage: idle_gaps.py EVENTS.csv KERNEL_TRACE.csv [--window LABEL[@k]]   (window = the k-th last scope with that label;
       default: from the first to the last host scope)
"""
import csv
import sys
from collections import defaultdict


def main():
    ev_path, kt_path = sys.argv[1], sys.argv[2]
    window = None
    if "--window" in sys.argv:
        window = sys.argv[sys.argv.index("--window") + 1]
    events = []
    for line in open(ev_path):
        parts = line.rstrip("\n").rsplit(",", 3)
        if len(parts) == 4:
            events.append((parts[0], int(parts[1]), int(parts[2]), parts[3]))
    kernels = []
    with open(kt_path) as f:
        for row in csv.DictReader(f):
            kernels.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    kernels.sort()
    if not events or not kernels:
        sys.exit("no events / kernels")
    lo, hi = min(e[1] for e in events), max(e[2] for e in events)
    if window:
        label, _, k = window.partition("@")
        sel = sorted((e for e in events if e[0] == label), key=lambda e: e[1])
        e = sel[-int(k or 1)]
        lo, hi = e[1], e[2]
    # the thread that owns the window: the one with the most scope time inside it
    by_thread = defaultdict(int)
    for lab, t0, t1, tid in events:
        if t1 > lo and t0 < hi:
            by_thread[tid] += min(t1, hi) - max(t0, lo)
    main_tid = max(by_thread, key=by_thread.get)
    scopes = sorted((e for e in events if e[3] == main_tid and e[2] > lo and e[1] < hi), key=lambda e: (e[1], -e[2]))
    # busy intervals of the GPU (union of the kernels), clipped to the window
    busy = []
    for s, e, _ in kernels:
        if e <= lo or s >= hi:
            continue
        s, e = max(s, lo), min(e, hi)
        if busy and s <= busy[-1][1]:
            busy[-1][1] = max(busy[-1][1], e)
        else:
            busy.append([s, e])
    idle = []
    cur = lo
    for s, e in busy:
        if s > cur:
            idle.append((cur, s))
        cur = max(cur, e)
    if cur < hi:
        idle.append((cur, hi))
    total_busy = sum(e - s for s, e in busy)
    total_idle = sum(e - s for s, e in idle)
    print(f"window {1e-6 * (hi - lo):.3f} ms: GPU busy {1e-6 * total_busy:.3f} ms, idle {1e-6 * total_idle:.3f} ms "
          f"({len(kernels)} kernels in the trace, clock overlap {'ok' if busy else 'NONE - different clocks?'})")
    # attribute idle time to the innermost open scope (scopes nest properly on one thread)
    acc = defaultdict(float)
    bounds = sorted({lo, hi} | {t for e in scopes for t in (e[1], e[2])})
    # sweep over idle intervals; innermost scope = the open one with the latest start
    for a, b in idle:
        pts = [a] + [t for t in bounds if a < t < b] + [b]
        for x, y in zip(pts, pts[1:]):
            mid = (x + y) // 2
            inner = None
            for lab, t0, t1, _ in scopes:
                if t0 <= mid < t1 and (inner is None or t0 >= inner[1]):
                    inner = (lab, t0)
            acc[inner[0] if inner else "(outside any scope)"] += y - x
    print(f"{'GPU idle while the host is in':40s} {'ms':>10s}")
    for lab, ns in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"{lab:40s} {1e-6 * ns:10.3f}")


if __name__ == "__main__":
    main()

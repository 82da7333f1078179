#!/usr/bin/env python3
"""Timeline of ONE pass of a proof from a host-scope log (VMN_TRACE_EVENTS, csrc/hosttrace.h) laid over the rocprofv3 kernel
trace of the same run: for every phase (top-level host scope of the calling thread) its wall time, the GPU-busy time inside
it, the number of launches, and -- with --detail -- the kernels and the idle gaps between them with the host scope that
was open.  usage: timeline.py EVENTS.csv KERNEL_TRACE.csv [--phases a,b,c] [--detail PHASE] [--nth K]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("vmn::", "")
    return name[:70]


def main():
    ev_path, kt_path = sys.argv[1], sys.argv[2]
    arg = lambda k, d=None: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    phases = (arg("--phases") or "vmn_shuffle_reencrypt,pos:precompute,pos:commit,pos:reply,pos:compute_af,pos:set_commitment,pos:verify").split(",")
    detail = arg("--detail")
    nth = int(arg("--nth", "1"))
    ev = []
    for line in open(ev_path):
        p = line.rstrip("\n").rsplit(",", 3)
        if len(p) == 4:
            ev.append((p[0], int(p[1]), int(p[2]), p[3]))
    ks = []
    for r in csv.DictReader(open(kt_path)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", "")))
    ks.sort()
    total_wall = total_busy = 0
    for ph in phases:
        sel = sorted((e for e in ev if e[0] == ph), key=lambda e: e[1])
        if len(sel) < nth:
            continue
        _, lo, hi, tid = sel[-nth]
        inside = [k for k in ks if k[0] >= lo and k[0] < hi]
        merged = []
        for s, e, n, st in inside:
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        busy = sum(e - s for s, e in merged)
        last_end = max((k[1] for k in inside), default=hi)
        print(f"{ph:28s} host {1e-6 * (hi - lo):7.3f} ms  kernels {len(inside):4d}  gpu-busy {1e-6 * busy:7.3f} ms  "
              f"gpu done {1e-6 * (last_end - lo):7.3f} ms after the scope opened")
        total_wall += hi - lo
        total_busy += busy
        if detail == ph:
            scopes = sorted((e for e in ev if e[3] == tid and e[2] > lo and e[1] < hi and e[0] != ph), key=lambda e: (e[1], -e[2]))
            prev_end = lo
            for s, e, n, st in inside:
                gap = s - prev_end
                if gap > 20_000:
                    open_ = [sc[0] for sc in scopes if sc[1] <= prev_end + gap // 2 < sc[2]]
                    print(f"      .. idle {1e-3 * gap:8.1f} us   host in {open_[-1] if open_ else '-'}")
                print(f"   {1e-6 * (s - lo):8.3f} ms  {1e-3 * (e - s):8.1f} us  s{st} {n}")
                prev_end = max(prev_end, e)
    print(f"sum of phases: host {1e-6 * total_wall:.3f} ms, gpu-busy {1e-6 * total_busy:.3f} ms")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense the rocprofv3 passes written by tools/profile_pmc.sh into one JSON under profiles/.

usage: summarize_pmc.py gpurun_out/pmc_<tag> <elements_per_launch> profiles/<name>.json

Run A (headline alone) gives the per-element figures of k_modpow<Cfg<74,1>> (HBM bytes from FETCH_SIZE / WRITE_SIZE, KB
units, separate passes; VALU instructions from the SQ_* pass).  Run B (the proof legs) gives, for every kernel that
takes more than 0.5 % of the GPU time, per-call averages of the same counters and the VALU issue rate they imply:
issued lane-instructions per second = SQ_INSTS_VALU x 64 / duration, against the 39.3 T/s integer-VALU roof.
The JSON carries the fingerprint of the kernel sources it was measured on; bench.py ignores it when the build differs.
"""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 256 * 4 * 16 * 2.4e9
HEADLINE = "k_modpow_phased<vmn::Cfg<74, 1>"          # (round 4: the headline launch runs the phased kernel)


FAMILIES = {"modp": ("mont28.h", "modp_kernels.h", "gen/mont_rows.inc"), "ec": ("ec_kernels.h",), "light": ("light_kernels.h",)}


def fingerprint(family=None):
    """sha256 (16 hex digits) over the kernel sources of one family (the headline kernel lives in "modp"), or over all."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "verificatum-vmn_amd", "csrc")
    names = FAMILIES[family] if family else ("mont28.h", "modp_kernels.h", "light_kernels.h", "ec_kernels.h", "gen/mont_rows.inc")
    for name in names:
        h.update(open(os.path.join(base, name), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    """void vmn::k_fixed_exp<vmn::Cfg<74, 1> >(unsigned int*, ...) -> k_fixed_exp<vmn::Cfg<74, 1>>"""
    n = name.split("(")[0].replace("void ", "").replace("vmn::k_", "k_").replace(" >", ">").strip()
    return n


def counters(d):
    """{kernel: {counter: (sum, calls)}} of one pass."""
    acc = {}
    paths = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
    for path in paths:
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            c = acc.setdefault(k, {})
            s, n = c.get(row["Counter_Name"], (0.0, 0))
            c[row["Counter_Name"]] = (s + float(row["Counter_Value"]), n + 1)
    return acc


def stats(d):
    out = {}
    for path in glob.glob(d + "/*/*_kernel_stats.csv") + glob.glob(d + "/*_kernel_stats.csv"):
        for row in csv.DictReader(open(path)):
            out[short(row["Name"])] = {"calls": int(row["Calls"]), "total_ns": float(row["TotalDurationNs"]),
                                       "avg_ns": float(row["AverageNs"]), "pct": float(row["Percentage"])}
    return out


def per_call(c, name):
    s, n = c.get(name, (0.0, 0))
    return s / n if n else None


def summarize(root, run, min_pct):
    st = stats(f"{root}/{run}/trace")
    f, w, sq = counters(f"{root}/{run}/fetch"), counters(f"{root}/{run}/write"), counters(f"{root}/{run}/sq")
    res = {}
    for k, srow in sorted(st.items(), key=lambda kv: -kv[1]["total_ns"]):
        if srow["pct"] < min_pct or k not in sq:
            continue
        insts = per_call(sq[k], "SQ_INSTS_VALU")
        gui = per_call(sq[k], "GRBM_GUI_ACTIVE")
        fetch, write = per_call(f.get(k, {}), "FETCH_SIZE"), per_call(w.get(k, {}), "WRITE_SIZE")
        avg_s = srow["avg_ns"] / 1e9
        e = {"calls": srow["calls"], "avg_kernel_ms": srow["avg_ns"] / 1e6, "share_of_gpu_time_pct": srow["pct"],
             "SQ_INSTS_VALU_per_call": insts, "FETCH_SIZE_KB_per_call": fetch, "WRITE_SIZE_KB_per_call": write}
        if insts:
            e["issued_Tlaneinstr_per_s"] = insts * 64 / avg_s / 1e12
            e["valu_issue_frac_of_39.3T"] = insts * 64 / avg_s / PEAK
        if gui and insts:
            cyc = gui / 8.0                                 # the counter is summed over the 8 XCDs
            e["effective_clock_GHz"] = cyc / avg_s / 1e9
            e["valu_busy_frac"] = 4.0 * insts / (cyc * 256 * 4)
        if fetch is not None and write is not None:
            e["hbm_bytes_per_call"] = (fetch + write) * 1024.0
            e["hbm_GBs"] = (fetch + write) * 1024.0 / avg_s / 1e9
        res[k] = e
    return res


def main():
    root, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    A = summarize(root, "A", 5.0)
    B = summarize(root, "B", 0.5) if os.path.isdir(root + "/B") else {}
    kernels = {}
    for k, e in A.items():
        if HEADLINE in k:
            e = dict(e)
            e["elements_per_launch"] = n
            e["valu_instr_per_unit"] = e["SQ_INSTS_VALU_per_call"] / (n / 64.0)      # wave instructions; one lane = one element
            e["hbm_bytes_per_unit"] = e["hbm_bytes_per_call"] / n
            e["note_fetch"] = ("FETCH_SIZE is not doubled: the guide's x2 correction is calibrated for wide coalesced streams; this kernel "
                               "reads 16 B per lane from per-lane table rows, so the read side may be under-counted by up to 2x; either way "
                               "traffic is ~2 % of the HBM roof (the per-lane window tables dominate the 768 B per element of pure I/O)")
            kernels[HEADLINE] = e
    res = {"command": "tools/profile_pmc.sh (rocprofv3 --kernel-trace --stats ; --pmc FETCH_SIZE ; --pmc WRITE_SIZE ; --pmc SQ_* : separate passes; "
                      "run A = headline alone, run B = the proof legs)",
           "source_fingerprint": fingerprint(), "family_fingerprints": {f: fingerprint(f) for f in FAMILIES}, "elements": n, "kernels": kernels, "proof_leg_kernels": B}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    main()

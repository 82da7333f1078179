#!/usr/bin/env python3
"""Regenerate the current round's measurement tables of DESIGN.md §6 from the committed profiles
(profiles/r04_bench_final.json = one `python bench.py` line), between the markers <!-- r04-tables-begin --> and
<!-- r04-tables-end -->.  (The round-2 and round-3 blocks of DESIGN.md were generated the same way from
profiles/r02_bench_v2.json / r03_bench_v10.json and are frozen text now.)

usage: tools/design_tables.py [profiles/r04_bench_final.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- r04-tables-begin -->", "<!-- r04-tables-end -->"
DEFAULT = os.path.join(ROOT, "profiles", "r04_bench_final.json")


def render(bench):
    """The generated block as text (tests/test_docs_consistency.py compares it with what DESIGN.md holds)."""
    r = json.load(open(bench))
    rel = os.path.relpath(bench, ROOT)
    rf = r["roofline"]
    mp, sm, cc, ec, dec = r["mix_prove"], r["mix_prove_n10000"], r["mix_ccpos_3072"], r["mix_ec_p256"], r["decrypt_2048"]
    e2e = mp["end_to_end"]

    def fr(leg):
        q = leg["roofline"]
        return (f"executed {q['frac']:.2f} wall / {q['frac_kernel_time']:.2f} kernel time; canonical (the headline's unit) "
                f"**{q['frac_canonical']:.2f}** wall / {q['frac_canonical_kernel_time']:.2f} kernel time")

    def shot(leg, key="total_ms"):
        return (f"set-up {leg['setup_ms']:.1f} ms + {leg[key]:.1f} ms = {leg[key + '_one_shot']:.1f} ms one-shot = "
                f"**{leg['ciphertexts_per_s_one_shot']:.3e} ciphertexts/s**")
    traffic = f"{rf['traffic'] / 1e9:.0f} GB per launch (PMC)" if rf.get("traffic") else "not re-measured for this build (see `roofline.traffic_source`)"
    rows = [
        f"| **headline**, configs[1]: 10^6 x 2048-bit modPow, 2047-bit exponents | **{r['value']:.3e} modexp/s**, {r['ms_per_step']:.1f} ms per step, "
        f"kernel {rf['avg_kernel_ms']:.1f} ms (HIP events) | canonical {rf['achieved']:.2f} TMAC/s = **{rf['frac']:.3f}** of the integer-VALU roof; HBM traffic {traffic} "
        f"against 0.77 GB algorithmic; CPU beside it: {r['cpu_baseline']['value']:.0f} modexp/s (GMP, {r['cpu_baseline']['cores']} cores, bit-exact) |",
        f"| **mix + prove**, 2048 bits, width 1 (re-encrypt + PoS prove + verify; mean of {len(mp['passes_total_ms'])} passes on fresh generators) | "
        f"{mp['total_ms']:.0f} ms = **{mp['ciphertexts_per_s']:.3e} ciphertexts/s**; {shot(mp)} | {fr(mp)}; CPU (GMP, "
        f"{mp['cpu_baseline']['cores']} cores): {mp['cpu_baseline']['value']:.0f} ciphertexts/s |",
        f"| the same END TO END (Fiat-Shamir hashing, byte trees published and parsed, verifier as another party) | {e2e['total_ms']:.0f} ms = "
        f"**{e2e['ciphertexts_per_s']:.2e} ciphertexts/s** ({e2e['ciphertexts_per_s_one_shot']:.2e} one-shot) | hash-bound: "
        f"{e2e['hashed_bytes_per_party'] / 1e9:.2f} GB of SHA-256 per party on one host core |",
        f"| the same at the reference's demo size, **configs[0]**: 10^4 ciphertexts (mean of {len(sm['passes_total_ms'])} passes) | {sm['total_ms']:.1f} ms = "
        f"**{sm['ciphertexts_per_s']:.3e} ciphertexts/s** (re-encrypt {sm['reencrypt_ms']:.1f}, prove {sm['prove_ms']:.1f}, verify {sm['verify_ms']:.1f} ms; "
        f"{sm['kernel_launches']} launches); {shot(sm)} | {fr(sm)} |",
        f"| **configs[2]**: 3072 bits, CCPoS path (mean of 2 cold passes) | offline {cc['offline_ms']:.0f} ms, online {cc['online_ms']:.0f} ms = "
        f"**{cc['ciphertexts_per_s_online']:.3e} ciphertexts/s**; {shot(cc, 'online_ms')}; factors precomputed as in `vmn -precomp`: "
        f"{cc['online_ms_factors_precomputed']:.0f} ms online | {fr(cc)} |",
        f"| **configs[4]** on one GPU: P-256, width 3, CCPoS (mean of 2 cold passes) | online {ec['online_ms']:.1f} ms = "
        f"**{ec['ciphertexts_per_s_online']:.3e} ciphertexts/s**; {shot(ec, 'online_ms')} | {fr(ec)} |",
        f"| **decryption half** (row A6 / N3): one of 3 parties, threshold 2, 2048 bits | {dec['total_ms']:.0f} ms = **{dec['ciphertexts_per_s']:.3e} "
        f"ciphertexts/s**; CPU (GMP, {dec['cpu_baseline']['cores']} cores): {dec['cpu_baseline']['value']:.0f} | {fr(dec)} |",
    ]
    text = (f"`python bench.py` on one MI355X (`{rel}`), every leg at its BASELINE configuration's size (10^6).  Round-4 rules: the set-up of "
            "the long-lived fixed-base tables is timed and counted (one-shot figures), every figure is the MEAN of its passes, every PoS pass "
            "runs on fresh generators, both roofline units are given (executed 28-bit multiply-adds; canonical SURVEY.md §8d MACs):\n\n"
            "| leg | result | roofline (bound: integer VALU, peak 39.3 T multiply-adds/s) |\n|---|---|---|\n" + "\n".join(rows))
    ol, op = r.get("operation_length"), r.get("operation_length_p256")
    if ol and "error" not in ol:
        e, v = ol["e(N)_ms"], ol["v(N)_ms"]
        text += (f"\n\nThe reference's metric shape (`demo/mixnet/benchmarks/operation_length_analyze:70-108`), 2048-bit group, N = {ol['ciphertexts']}: "
                 f"executing **e(N) = {e['per_ciphertext'] * 1e3:.4f} us x N + {e['constant']:.1f} ms**, verifying **v(N) = {v['per_ciphertext'] * 1e3:.4f} us x N + "
                 f"{v['constant']:.1f} ms**.")
    if op and "error" not in op:
        e, v = op["e(N)_ms"], op["v(N)_ms"]
        pts = ", ".join(f"{n}: {a:.1f} + {b:.1f} ms" for n, a, b in zip(op["ciphertexts"], op["executing_ms"], op["verifying_ms"]))
        text += (f"  On the reference's own benchmark group (`bench_config:33`: P-256, width 1, PoS), N: executing + verifying = {pts}; fits "
                 f"**e(N) = {e['per_ciphertext'] * 1e3:.4f} us x N + {e['constant']:.1f} ms**, **v(N) = {v['per_ciphertext'] * 1e3:.4f} us x N + {v['constant']:.1f} ms**; "
                 f"{op['ciphertexts_per_s'][-1]:.3e} ciphertexts/s at N = 10^6 (canonical frac {op['frac_canonical'][-1]:.2f}).")
    sh = r.get("modexp_by_shape")
    if sh and "error" not in sh:
        text += ("\n\nThe shapes a proof is made of (`modexp_by_shape`: 2048 bits, 10^6 elements per call, device-resident; GMP on "
                 f"{sh.get('gmp', {}).get('cores', '?')} cores beside it):\n\n| shape | ops/s | frac (canonical §8d budget per op) | frac (canonical, products executed) | GMP ops/s |\n|---|---|---|---|---|\n")
        names = {"K1a_256": "K1a: X.exp(E), 256-bit per-element exponents", "K1a_612": "K1a: 612-bit per-element exponents",
                 "K1b_256": "K1b: X.exp(v), one 256-bit exponent", "K2_full": "K2: g.exp(E), fixed base, full-length exponents (warm table)",
                 "K3_256": "K3: expProd per term, 256-bit", "K3_612": "K3: expProd per term, 612-bit"}
        for k, label in names.items():
            s = sh[k]
            text += (f"| {label} | **{s['ops_per_s']:.3e}** | {s['frac_canonical_budget']:.2f} | {s['frac_canonical_executed']:.2f} | "
                     f"{s.get('gmp_ops_per_s', float('nan')):.3e} |\n")
        text = text.rstrip("\n")
    return text


def main():
    bench = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    text = render(bench)
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    i, j = s.index(BEGIN) + len(BEGIN), s.index(END)
    open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])
    print("DESIGN.md §6 tables regenerated from", os.path.relpath(bench, ROOT))


if __name__ == "__main__":
    main()

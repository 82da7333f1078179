#!/usr/bin/env python3
"""Regenerate the current round's measurement tables of DESIGN.md §6 from the committed profiles
(profiles/r03_bench_v<k>.json = one `python bench.py` line, profiles/r03_pmc_kernels.json = tools/summarize_pmc.py),
between the markers <!-- r03-tables-begin --> and <!-- r03-tables-end -->.  (The round-2 block of DESIGN.md was generated the
same way from profiles/r02_bench_v2.json / r02_pmc_kernels.json and is frozen text now.)

usage: tools/design_tables.py [profiles/r03_bench_v10.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render(bench):
    """The generated block as text (tests/test_docs_consistency.py compares it with what DESIGN.md holds)."""
    r = json.load(open(bench))
    pm = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_kernels.json")))
    mp, cc, ec, dec = r["mix_prove"], r["mix_ccpos_3072"], r["mix_ec_p256"], r["decrypt_2048"]
    e2e = mp["end_to_end"]
    rel = os.path.relpath(bench, ROOT)

    def fam(leg, name):
        f = leg["roofline"]["by_family"].get(name)
        return f"{f['ms']:.0f} ms, {f['frac']:.2f}" if f else "-"
    pk = pm["proof_leg_kernels"]

    def pkrow(name):
        e = pk.get(name)
        if e is None:
            return f"| `{name}` | (not among the 20 longest of this run) | | | | |"
        return (f"| `{name}` | {e['calls']} | {e['avg_kernel_ms']:.2f} | {e['valu_issue_frac_of_39.3T']:.2f} | "
                f"{e['valu_busy_frac']:.2f} | {e['hbm_GBs']:.0f} |")
    hk = pm["kernels"]["k_modpow<vmn::Cfg<74, 1>"]
    rf = r["roofline"]
    issued = f"{rf['issued_Tlaneinstr_per_s']:.1f}" if rf.get("issued_Tlaneinstr_per_s") else "n/a"
    traffic = f"{rf['traffic'] / 1e9:.0f} GB" if rf.get("traffic") else "n/a"
    ol = r.get("operation_length")
    fit_text = ""
    if ol and "error" not in ol:
        e, v, pb = ol["e(N)_ms"], ol["v(N)_ms"], ol.get("p(N)_bytes")
        fit_text = (f"\n\nThe reference's own metric shape (`demo/mixnet/benchmarks/operation_length_analyze:70-108`: affine fits over several "
                    f"numbers of ciphertexts; `operation_length` of the line, N = {ol['ciphertexts']}): executing (shuffle + prove) "
                    f"**e(N) = {e['per_ciphertext'] * 1e3:.4f} us x N + {e['constant']:.1f} ms**, verifying **v(N) = {v['per_ciphertext'] * 1e3:.4f} us x N + "
                    f"{v['constant']:.1f} ms**" + (f", proof size p(N) = {pb['per_ciphertext']:.0f} x N + {pb['constant']:.0f} bytes" if pb else "") +
                    " (arithmetic only: no hashing, no network).")
    def pre_text(leg):                       # the reference's precomputed shuffle: factors offline (bench.py: precomputed_factors_fields)
        if "online_ms_factors_precomputed" not in leg:
            return ""
        return (f"; with the re-encryption factors precomputed as in `vmn -precomp` ({leg['reencrypt_factors_ms']:.0f} ms, offline) the online "
                f"part is {leg['online_ms_factors_precomputed']:.0f} ms = {leg['ciphertexts_per_s_online_factors_precomputed']:.3e} ciphertexts/s")
    sm = r.get("mix_prove_n10000")
    small_row = (f"| the same at the reference's demo size, **configs[0]**: 10^4 ciphertexts | {sm['total_ms']:.1f} ms = **{sm['ciphertexts_per_s']:.3e} ciphertexts/s** "
                 f"(prove {sm['prove_ms']:.1f} ms, verify {sm['verify_ms']:.1f} ms) | wide geometries + fixed-base chains cut into pieces (§5): kernels {sm['roofline']['kernel_ms']:.1f} ms of it, "
                 f"latency-bound chains (frac {sm['roofline']['frac']:.2f}) |\n") if sm and "error" not in sm else ""
    text = f"""`python bench.py` on one MI355X (`{rel}`), every leg at its BASELINE configuration's size (10^6):

| leg | result | roofline (bound: integer VALU, peak 39.3 T multiply-adds/s) |
|---|---|---|
| **headline**, configs[1]: 10^6 x 2048-bit modPow, 2047-bit exponents | **{r['value']:.3e} modexp/s**, {r['ms_per_step']:.1f} ms per step, kernel {rf['avg_kernel_ms']:.1f} ms (HIP events) | achieved {rf['achieved']:.2f} TMAC/s canonical (16 422 432 per modexp, SURVEY.md §8d) = **{rf['frac']:.3f}**; issued {issued} T lane-instr/s of the {rf['peak_measured']:.1f} T/s the hardware sustains for `v_mad_u64_u32` at two waves per SIMD; HBM traffic {traffic} per launch (PMC) against 0.77 GB algorithmic: the per-lane window tables, 2 % of the HBM roof |
| **mix + prove**, 2048 bits, width 1: re-encrypt + PoS prove + verify (GPU arithmetic; the N-sized random arrays expanded on the device, scalars and the permutation from a tape) | {mp['total_ms']:.0f} ms = **{mp['ciphertexts_per_s']:.3e} ciphertexts/s** | executed {mp['roofline']['executed_T_mads']:.1f} T multiply-adds: frac **{mp['roofline']['frac']:.2f}** of the wall clock, {mp['roofline']['frac_kernel_time']:.2f} of the kernel time (fixed {fam(mp, 'fixed')}; modpow {fam(mp, 'modpow')}; expprod {fam(mp, 'expprod')}) |
| the same END TO END (prover randomness on the device, Fiat-Shamir hashing, byte trees published and parsed, verifier as another party) | prove {e2e['prove_ms']:.0f} ms + verify {e2e['verify_ms']:.0f} ms = {e2e['total_ms']:.0f} ms = **{e2e['ciphertexts_per_s_mean_of_passes']:.2e} ciphertexts/s** (mean of the two passes {e2e['passes_total_ms']}; {e2e['ciphertexts_per_s_parties_in_parallel']:.2e} with prover and verifier on their own machines) | {e2e['hashed_bytes_per_party'] / 1e9:.2f} GB hashed per party (SHA-256, one host core, ~2.2 GB/s): the hash, not the GPU, is the critical path (see below) |
{small_row}| **configs[2]**: 3072 bits, CCPoS path | offline (commitment + PoSC) {cc['offline_ms']:.0f} ms, online (re-encrypt + CCPoS prove + verify) {cc['online_ms']:.0f} ms = **{cc['ciphertexts_per_s_online']:.3e} ciphertexts/s**{pre_text(cc)} | frac **{cc['roofline']['frac']:.2f}** wall / {cc['roofline']['frac_kernel_time']:.2f} kernel time (fixed {fam(cc, 'fixed')}; modpow {fam(cc, 'modpow')}; expprod {fam(cc, 'expprod')}) |
| **configs[4]** on one GPU: P-256, width 3, CCPoS | online {ec['online_ms']:.0f} ms = **{ec['ciphertexts_per_s_online']:.3e} ciphertexts/s**{pre_text(ec)} | executed-work frac {ec['roofline']['frac']:.2f} wall / {ec['roofline']['frac_kernel_time']:.2f} kernel time (expprod {fam(ec, 'expprod')}; scans {fam(ec, 'scan')}; normalisation {fam(ec, 'normalize')}); against SURVEY.md §8d's canonical field product M(8) = 136 (the kernels execute 160 per product): {ec['roofline']['frac_canonical']:.2f} wall / {ec['roofline']['frac_canonical_kernel_time']:.2f} kernel time (§5, curves, round 3) |
| **decryption half** (row A6 / N3): one of k = 3 parties, threshold 2, 2048 bits, 10^6 ciphertexts | own factors {dec['own_factors_ms']:.0f} ms + own proof {dec['own_proof_ms']:.0f} ms + check of the others {dec['verify_others_ms']:.0f} ms + combination and plaintexts {dec['combine_and_plaintexts_ms']:.0f} ms = {dec['total_ms']:.0f} ms = **{dec['ciphertexts_per_s']:.3e} ciphertexts/s**; CPU (GMP, {dec['cpu_baseline']['cores']} cores, {dec['cpu_baseline']['sample'].split(',')[0]}): {dec['cpu_baseline']['value']:.0f} ciphertexts/s | frac **{dec['roofline']['frac']:.2f}** wall / {dec['roofline']['frac_kernel_time']:.2f} kernel time (modpow {fam(dec, 'modpow')}: one full-length power per ciphertext with the party's secret exponent) |
| CPU beside it (GMP, {r['cpu_baseline']['cores']} cores of the box) | {r['cpu_baseline']['value']:.0f} modexp/s (`mpz_powm`, 96 000-element sample, bit-exact vs the GPU); mix + prove {mp['cpu_baseline']['value']:.0f} ciphertexts/s ({mp['cpu_baseline']['sample'].split(':')[0]}; fixed-base tables, Pippenger) | |

PMC passes (`tools/profile_pmc.sh` -> `tools/summarize_pmc.py` -> `profiles/r03_pmc_kernels.json`; separate `--pmc` passes
for FETCH_SIZE, WRITE_SIZE and the SQ counters; 262 144 elements / ciphertexts; per-kernel durations of the same runs:
`profiles/r03_pmc_runA_headline_kernel_stats.csv`, `r03_pmc_runB_proof_legs_kernel_stats.csv`).  "issue" = SQ_INSTS_VALU x 64 lanes
/ duration against 39.3 T/s; "busy" = VALU busy fraction from GRBM_GUI_ACTIVE:

| kernel | calls | avg ms | issue | busy | HBM GB/s |
|---|---|---|---|---|---|
| `k_modpow<Cfg<74,1>>` (headline alone, run A) | {hk['calls']} | {hk['avg_kernel_ms']:.1f} | {hk['valu_issue_frac_of_39.3T']:.2f} | {hk['valu_busy_frac']:.2f} | {hk['hbm_GBs']:.0f} |
{pkrow('k_fixed_exp<vmn::Cfg<110, 2>>')}
{pkrow('k_modpow2<vmn::Cfg<110, 2>>')}
{pkrow('k_modpow2<vmn::Cfg<74, 1>>')}
{pkrow('k_fixed_exp<vmn::Cfg<74, 1>>')}
{pkrow('k_bucket_level<vmn::Cfg<110, 2>, true>')}
{pkrow('k_bucket_level<vmn::Cfg<74, 1>, true>')}
{pkrow('k_ec_bucket_level<10, true>')}
{pkrow('k_ec_fixed_exp<10>')}
{pkrow('k_scan_apply<vmn::Cfg<112, 4>>')}

The headline kernel executes {hk['valu_instr_per_unit'] / 1e6:.2f} M VALU instructions per element (canonical 16.42 M MACs = {16.422432e6 / hk['valu_instr_per_unit']:.2f} of them) at
{hk['issued_Tlaneinstr_per_s']:.1f} T lane-instr/s.

End-to-end timeline at N = 10^6 (prover): seed-independent GPU work done at {e2e['prover_phases_ms']['seed_independent_gpu_work_done']:.0f} ms, seed known at
{e2e['prover_phases_ms']['seed_known']:.0f} ms (hash thread busy {e2e['instance_hash_thread_busy_ms'][0]:.0f} ms: fully overlapped, hash-bound), commitment published at {e2e['prover_phases_ms']['commitment_published']:.0f} ms, challenge
at {e2e['prover_phases_ms']['challenge_known']:.0f} ms, reply at {e2e['prover_phases_ms']['reply_published']:.0f} ms; verifier: seed at {e2e['verifier_phases_ms']['seed_known']:.0f} ms, computeAF beside the challenge hash, verdict at
{e2e['verifier_phases_ms']['verified']:.0f} ms.  GPU kernels are {e2e['gpu_kernel_ms']:.0f} ms of the {e2e['total_ms']:.0f} ms.""" + fit_text
    return text


BEGIN, END = "<!-- r03-tables-begin -->", "<!-- r03-tables-end -->"


def main():
    bench = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_bench_v10.json")
    text = render(bench)
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    i, j = s.index(BEGIN) + len(BEGIN), s.index(END)
    open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])
    print("DESIGN.md §6 tables regenerated from", os.path.relpath(bench, ROOT))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native VMN exponentiation core.

Workload (BASELINE.json configs[1]): one *step* = one batched variable-base modular
exponentiation  out[i] = x[i]^e[i] mod p  over N = 1,000,000 elements, p = RFC 3526 group 14
(2048-bit safe prime), random bases, random full-length (2047-bit) exponents, inputs and outputs
resident in HBM (device arrays of the C ABI), through ``vmn_garray_exp_array`` — the call the
reference makes as ``PGroupElementArray.exp(PRingElementArray)`` (PoSBasicTW.java:1032).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (torch.distributed / RCCL), every rank holds its own shard of N_per_gpu
elements (weak scaling, no data-path collective: element-wise op), barrier + synchronize on both
sides of the timed region, MAX over ranks.

The JSON line carries
  roofline     : the dominant kernel (k_modpow) against the integer-VALU roofline.  This path is
                 integer big-number arithmetic: it is bound by VALU issue, not by HBM and not by
                 MFMA, so bound = "valu-int".  achieved = algorithmic 32x32-bit
                 multiply-accumulates (SURVEY.md §8d canonical count, 16,422,432 per 2048/2047-bit
                 modexp) per launch / the kernel's average duration measured with HIP events on
                 the launch stream; peak = 39.3 TMAC/s = 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz,
                 the issue rate of v_mad_u64_u32 measured by tools/valu_rate.hip
                 (profiles/valu_rate_r01.txt) = the FP64-vector FMA rate of MI355X_MICROARCH.md.
  cpu_baseline : the GMP oracle (mpz_powm, OpenMP over all host cores) on a bounded sample of the
                 same inputs; the GPU output for that sample is compared bit for bit.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_2048_2047 = 16_422_432           # fixed-window modpow, n = 2048, t = 2047, w = 5
PEAK_TMACS = 256 * 4 * 16 * 2.4e9 / 1e12   # 39.3
HBM_PEAK_GBS = 8000.0


def leg_roofline(fam: dict, wall_ms: float) -> dict:
    """Roofline object of a proof leg, in BOTH units (peak = the integer-VALU issue rate, one multiply-add per lane per
    4 cycles: 39.3 T/s):
      executed   the v_mad_u64_u32 multiply-adds the kernels' lanes perform on 28-bit limbs (csrc/vmnhip.hip note_work:
                 Montgomery products x 2 S^2, block-symmetric squarings, point additions as field products);
      canonical  the SAME products priced in SURVEY.md §8d's unit, the headline's: 32 x 32-bit multiply-accumulates of a
                 product on s = bits / 32 limbs, M(s) = 2 s^2 + s (Q(s) for a squaring) -- (64/74)^2 ~ 0.75 of the
                 executed count at 2048 bits, 136/160 for a P-256 field product.  It counts the products the kernels
                 DO (a fixed-base power is its 108-128 table products, not the 256 of §8d's radix-2^8 budget).
    `frac*` are against the wall clock of the leg (host gaps, HBM-bound kernels and sorting included),
    `*_kernel_time` against the summed kernel durations."""
    mads = sum(v[2] for v in fam.values())
    canon = sum(v[3] for v in fam.values())
    kernel_ms = sum(v[1] for v in fam.values())
    by = {k: {"ms": round(v[1], 3), "T_mads": round(v[2] / 1e12, 4),
              "frac": round(v[2] / (v[1] / 1e3) / 1e12 / PEAK_TMACS, 4) if v[1] > 0 else None,
              "frac_canonical": round(v[3] / (v[1] / 1e3) / 1e12 / PEAK_TMACS, 4) if v[1] > 0 else None}
          for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]) if v[2] > 0}
    per_s = lambda x, ms: x / (ms / 1e3) / 1e12 if ms else None
    frac = lambda x, ms: x / (ms / 1e3) / 1e12 / PEAK_TMACS if ms else None
    return {"bound": "valu-int", "unit": "T multiply-adds/s", "peak": PEAK_TMACS,
            "executed_T_mads": mads / 1e12, "canonical_T_macs_survey_8d": canon / 1e12,
            "achieved": per_s(mads, wall_ms), "frac": frac(mads, wall_ms),
            "achieved_canonical": per_s(canon, wall_ms), "frac_canonical": frac(canon, wall_ms),
            "kernel_ms": kernel_ms, "achieved_kernel_time": per_s(mads, kernel_ms), "frac_kernel_time": frac(mads, kernel_ms),
            "frac_canonical_kernel_time": frac(canon, kernel_ms),
            "units": "frac: v_mad_u64_u32 executed (28-bit limbs); frac_canonical: the same products at SURVEY.md §8d's "
                     "M(s) = 2 s^2 + s on 32-bit limbs (the headline's unit)",
            "by_family": by, "profiles": "profiles/r04_pmc_kernels.json (rocprofv3 --pmc passes of the dominant kernels)"}


def source_fingerprint(names=("mont28.h", "modp_kernels.h", "gen/mont_rows.inc")) -> str:
    """sha256 (16 hex digits) over the sources of the headline kernel's family (tools/summarize_pmc.py FAMILIES["modp"]):
    PMC summaries under profiles/ carry the fingerprints of the build they were measured on, and the headline's counters
    are used only when its family's still matches (a change to the curve kernels does not stale them)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "verificatum-vmn_amd", "csrc")
    for name in names:
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_valu_peak():
    """T instr/s the hardware sustains for independent v_mad_u64_u32 at two waves per SIMD (tools/valu_rate.hip ->
    profiles/valu_rate_r01.csv), or None when the file is missing."""
    import csv
    try:
        best = None
        with open(os.path.join(ROOT, "profiles", "valu_rate_r01.csv")) as f:
            for row in csv.DictReader(f):
                if row.get("instr", "").startswith("v_mad_u64_u32 indep") and row.get("waves_per_simd") == "2":
                    best = float(row["Tlaneops_per_s"])
        return best
    except Exception:
        return None


def make_inputs(n: int, seed: int, nbytes: int):
    """n random bases (< 2^2047 < p) and exponents (2047-bit, < q) as big-endian bytes."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
    e = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
    x[:, 0] &= 0x7F
    e[:, 0] &= 0x7F
    x[:, -1] |= 1                     # no zero base
    return x.tobytes(), e.tobytes()


def load_sub(entry, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(f"verificatum_vmn_amd.{name}", os.path.join(entry.PKG_DIR, f"{name}.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m


class ReplaySource:
    """Random tape generated BEFORE the timed region and replayed in order.  Random generation (the
    reference's K9 family: PRG / randomElementArray, VCR code that is not in the reference tree) is an
    input of the measured path, not part of it."""

    def __init__(self, src, plan, pin: bool = True):
        from collections import deque
        self.q = deque((kind, self._host(getattr(src, kind)(*args), pin)) for kind, *args in plan)

    @staticmethod
    def _host(v, pin):
        """Large blocks live in page-locked host memory, as the direct buffers of an integration would: the
        upload is then asynchronous and runs at PCIe speed instead of through the runtime's staging copies."""
        if pin and isinstance(v, (bytes, bytearray)) and len(v) >= (1 << 20):
            import torch
            return torch.frombuffer(bytearray(v), dtype=torch.uint8).pin_memory()
        return v

    def _next(self, kind):
        k, v = self.q.popleft()
        assert k == kind, f"tape out of order: wanted {kind}, recorded {k}"
        return v

    def permutation(self, n):
        return self._next("permutation")

    def ring_array(self, n):
        return self._next("ring_array")

    def ring_element(self):
        return self._next("ring_element")

    def int_array(self, n, bits):
        return self._next("int_array")


class ReplayWithDeviceArrays(ReplaySource):
    """The same tape for the permutation and the O(1) scalars; the N-sized draws of the prover (r, s, b, beta, epsilon) are
    expanded ON THE DEVICE from 32-byte seeds (vmn_random_source.array_seed, csrc/vmnproofs.cpp: random_ring_array) -- the
    product's own path for PoSBasicTW.java:446, 473, 583, 612 and ShufflerElGamalSession.java:408-409: no N-sized
    array crosses PCIe inside the timed region."""

    def __init__(self, src, plan):
        super().__init__(src, plan)
        self._seeds = src

    def array_seed(self):
        return self._seeds.array_seed()


def fiat_shamir_seed(grp, arrays, prefix: bytes = b"bench"):
    """The hashing the reference does on the host around a proof (reported as its own line, never part
    of ciphertexts/s): seed = RO(prefix || bytetree(g, h, u, pk, w, w')) as in PoSTW.java:118-130 -- the byte trees of
    the N-sized arrays are framed on the GPU, downloaded into one page-locked buffer and fed to SHA-256 (hashlib).
    Returns (seed, milliseconds, bytes hashed)."""
    import hashlib
    import torch
    size = max(a.byteTreeSize() for a in arrays)
    buf = torch.empty(size, dtype=torch.uint8).pin_memory()
    view = memoryview(buf.numpy())
    t0 = time.perf_counter()
    hsh = hashlib.sha256((256).to_bytes(4, "big") + prefix)          # RandomOracle: H(uint32(nout) || data)
    total = 0
    for a in arrays:
        nbytes = a.toByteTreeInto(buf)
        hsh.update(view[:nbytes])
        total += nbytes
    seed = hsh.digest()
    return seed, (time.perf_counter() - t0) * 1e3, total


def modules(entry):
    """(native, randomsource): the ctypes bindings of the C++ proof drivers (include/vmnproofs.h) and the tapes."""
    return load_sub(entry, "native"), load_sub(entry, "randomsource")


def session_setup(ctx, grp, bases_uses, n: int, sync) -> dict:
    """The set-up a mix server does for its long-lived bases, TIMED: vmn_group_precompute_fixed builds the fixed-base table
    of each (base, expected uses) pair sized for arrays of n exponents.  The reference builds nothing ahead
    (ShufflerElGamalSession.java:400-409 calls widePublicKey.exp directly; VCR/GMPMEE build their fixed-base tables inside
    the call), so a ONE-SHOT figure must count this: every leg reports setup_ms and total_ms_one_shot = setup + the
    mean pass.  `uses` is what one shuffle + proof + verification really does with the base (g: 2N re-encryption + ~5N of
    the prover + 1N of the verifier; the key: the re-encryption only), not a long session's reuse."""
    sync()
    t0 = time.perf_counter()
    before = grp.tableBytes()
    for base, uses in bases_uses:
        grp.precomputeFixed(base, n, uses)
    sync()
    return {"setup_ms": (time.perf_counter() - t0) * 1e3, "fixed_tables": len(bases_uses),
            "table_bytes": grp.tableBytes() - before, "uses_hint": [u for _, u in bases_uses]}


def one_shot_fields(n: int, setup: dict, mean_ms: float, key: str = "total_ms") -> dict:
    return {"setup_ms": setup["setup_ms"], "setup": setup, f"{key}_one_shot": setup["setup_ms"] + mean_ms,
            "ciphertexts_per_s_one_shot": n / ((setup["setup_ms"] + mean_ms) / 1e3),
            "one_shot_note": f"ONE shuffle from a cold group: the fixed-base tables of g and the key (setup_ms, inside this figure) + the "
                             f"mean {key} of the passes; nothing is amortised over a session"}


def precomputed_factors_fields(nat, grp, pkey, W, S, pi, n, sync, prove_verify_s):
    """The re-encryption as the reference's precomputed shuffle splits it (ShufflerElGamalSession.java:645-661: the factors
    pk^s in `vmn -precomp`; :789-792: input.mul(factors).permute(inverse) when the ciphertexts arrive), timed on its own AFTER
    the leg's timed pass (which re-encrypts in one call, BASELINE's "full mix"): what a mix server with precomputed factors
    does online = applying them + CCPoS prove + verify."""
    sync()
    t0 = time.perf_counter()
    factors = nat.reencryption_factors_native(grp, pkey, S)
    sync()
    t1 = time.perf_counter()
    WP = nat.apply_factors_native(grp, W, factors, pi)
    sync()
    t2 = time.perf_counter()
    for a in list(factors) + list(WP):
        a.free()
    rest = (t2 - t1) + prove_verify_s
    return {"reencrypt_factors_ms": (t1 - t0) * 1e3, "reencrypt_apply_factors_ms": (t2 - t1) * 1e3,
            "online_ms_factors_precomputed": rest * 1e3, "ciphertexts_per_s_online_factors_precomputed": n / rest,
            "factors_precomputed_note": "the reference's precomputed shuffle computes the re-encryption factors in `vmn -precomp` "
                                        "(ShufflerElGamalSession.java:645-661) and only multiplies and permutes online (:789-792); "
                                        "online_ms_factors_precomputed = reencrypt_apply_factors_ms (timed after the pass) + "
                                        "ccpos_prove_ms + ccpos_verify_ms (of the pass)"}


def mean_of(runs, key):
    return sum(r[key] for r in runs) / len(runs)


def mix_prove(entry, vmn, ctx, grp, n: int, seed: int, sync, steps: int = 2, width: int = 1, fs_line: bool = True):
    """ciphertexts/s of [A0 re-encrypt + PoS prove + PoS verify] (SURVEY.md §8a rows A0 + A1), device-resident arrays,
    n_e = n_v = 256, n_r = 100, over any group of the library (ModPGroup or a curve).  The op sequence is the reference's
    (ShufflerElGamalSession.java:400-409, 273-278; PoSBasicTW.java precompute/commit/reply/computeAF/verify).
    Measurement rules (round 4): the set-up of the long-lived bases is timed and reported (session_setup); every pass runs on
    a FRESH list of independent generators h, so the table of the per-proof base h_0 is built inside the clock; the figure
    is the MEAN of the passes (all listed), not the best."""
    nat, rs = modules(entry)
    NV = NE = 256
    NR = 100
    q, g = grp.q, grp.g
    rnd = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    # synthetic instance (untimed): key y = g^x, honest ciphertexts (g^t, m*y^t) per column
    y = grp.k_exp(g, rnd.ring_element())
    pkey = [g] * width + [y] * width
    setup = session_setup(ctx, grp, [(g, 8), (y, 1)], n, sync)
    arr = lambda: grp.ringArrayFromPRG(rnd.array_seed(), n, q.bit_length() - 1)
    Hs = []
    for _ in range(steps + 1):                              # one list of generators per pass (the last one: the untimed first pass)
        A = arr()
        Hs.append(grp.exp(g, A))
        A.free()
    W = [None] * (2 * width)
    for c in range(width):
        T, Mx = arr(), arr()
        M, YT = grp.exp(g, Mx), grp.exp(y, T)
        W[c] = grp.exp(g, T)
        W[width + c] = M.mul(YT)
        for a in (T, Mx, M, YT):
            a.free()
    runs = []
    fam_instr = None
    first_pass = None
    for step in range(-1, steps + 1):
        # The LAST pass is the instrumented one: the library's per-launch accounting (two HIP events per launch, read back
        # by timing_report) is what the roofline breakdown comes from -- and costs ~10-20 us per launch, which at 350 launches
        # and 10^4 ciphertexts is a fifth of the proof.  It runs on the generators of the first pass, after the timed passes,
        # and its wall clock is reported as `instrumented_pass_ms`, never as the leg's figure.
        # step -1 is not timed: the first pass of a leg in this process (first launches of its kernels, the array pool at
        # its sizes) is reported as first_pass_total_ms beside the figure, like the headline's warm-up steps
        instrumented = step == steps
        H = Hs[0] if instrumented else Hs[step]           # (Hs[-1]: generators of their own, so that no timed pass finds its h_0 table)
        # pi, alpha, the seed of the batching vector, gamma / delta / phi (width of them), v; the N-sized draws (s, r,
        # epsilon, b, beta) are expanded on the device from 32-byte seeds (vmn_random_source.array_seed)
        tape = ReplayWithDeviceArrays(rnd, [("permutation", n), ("ring_element",), ("int_array", 1, 256),
                                            ("ring_element",), ("ring_element",)] + [("ring_element",)] * width + [("int_array", 1, NV)])
        ctx.timing_reset()
        ctx.timing_enable(instrumented)
        gc.collect()        # release the previous pass's arrays into the pool before the clock starts
        sync()
        t0 = time.perf_counter()
        # --- A0: re-encryption + permutation
        pi = tape.permutation(n)
        S = [nat.random_ring_array_native(grp, tape, n, NR) for _ in range(width)]
        prover = nat.PoSBasicTW(grp, NV, NE, NR, rand=tape)
        WP = nat.reencrypt_native(grp, pkey, W, S, pi)
        sync()
        t1 = time.perf_counter()
        # --- A1 prover
        prover.precompute(g, H, pi)
        prover.setInstance(pkey, W, WP, S)
        e_seed = bytes(tape.int_array(1, 256))[-32:]        # setBatchVector(byte[] prgSeed): e is derived on the GPU
        prover.setBatchVectorSeed(e_seed)
        com = prover.commit()
        v = int.from_bytes(tape.int_array(1, NV), "big")
        rep = prover.reply(v)
        sync()
        t2 = time.perf_counter()
        # --- A1 verifier
        ver = nat.PoSBasicTW(grp, NV, NE, NR)
        ver.precompute(g, H)
        ver.setPermutationCommitment(prover.u)
        ver.setInstance(pkey, W, WP)
        ver.setBatchVectorSeed(e_seed)
        ver.computeAF()
        ver.setCommitment(com)
        ver.setChallenge(v)
        ok = ver.verify(rep)
        sync()
        t3 = time.perf_counter()
        ctx.timing_enable(False)
        cur = {"reencrypt_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "verify_ms": (t3 - t2) * 1e3,
               "total_ms": (t3 - t0) * 1e3, "accepted": bool(ok),
               # size of the Fiat-Shamir proof: u, commitment, reply as byte trees (p(N) of the reference's analysis)
               "proof_bytes": prover.u.byteTreeSize() + com.native.byteTreeSize() + rep.native.byteTreeSize()}
        if instrumented:
            fam_instr = (ctx.timing_report(), cur)
        elif step < 0:
            first_pass = cur
        else:
            runs.append(cur)
        for a in WP + S:
            a.free()
        com = rep = None                                   # the native messages own B, B', k_B, k_E
        ver.free()
        prover.free()
    out = dict(min(runs, key=lambda r: abs(r["total_ms"] - mean_of(runs, "total_ms"))))     # the pass nearest the mean carries the detail
    for key in ("reencrypt_ms", "prove_ms", "verify_ms", "total_ms"):
        out[key] = mean_of(runs, key)
    fam, instr = fam_instr
    out["kernel_ms_by_family"] = {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}
    out["kernel_launches"] = sum(v[0] for v in fam.values())
    out["roofline"] = leg_roofline(fam, out["total_ms"])          # the work of a pass against the MEAN wall clock of the timed passes
    out["instrumented_pass_ms"] = instr["total_ms"]
    out["accepted"] = all(r["accepted"] for r in runs) and instr["accepted"]
    out["passes_total_ms"] = [round(r["total_ms"], 2) for r in runs]
    out["first_pass_total_ms"] = first_pass["total_ms"]
    out["accepted"] = out["accepted"] and first_pass["accepted"]
    out["statistic"] = (f"mean of {steps} passes, each on fresh generators h (the h_0 table is built inside the pass), per-launch event "
                        "accounting OFF, after one untimed pass (first_pass_total_ms); the roofline breakdown comes from one more, "
                        "instrumented pass (instrumented_pass_ms)")
    out["ciphertexts_per_s"] = n / (out["total_ms"] / 1e3)
    out["n"] = n
    out["width"] = width
    out.update(one_shot_fields(n, setup, out["total_ms"]))
    if fs_line:
        # the Fiat-Shamir hashing of the public arrays (host, SHA-256), its own line
        _, fs_ms, fs_bytes = fiat_shamir_seed(grp, Hs[:2] + W[:2] + W[:2])      # stands for h, u, w, w' (six N-sized arrays at width 1)
        out["fiat_shamir_host_ms"] = fs_ms
        out["fiat_shamir_bytes"] = fs_bytes
    for a in Hs + W:
        a.free()
    grp.releaseFixed(y)                                  # the key of this synthetic instance will not come back
    return out


def operation_length_fit(points):
    """The reference's own metric shape (demo/mixnet/benchmarks/operation_length_analyze:70-108): running time of executing
    e(N) and of verifying v(N) one shuffle, and the size p(N) of its Fiat-Shamir proof, as affine functions a N + b fitted
    (least squares) over several numbers of ciphertexts.  `points`: mix_prove results at different n."""
    import numpy as np
    ns = np.array([p["n"] for p in points], dtype=float)

    def fit(ys):
        a, b = np.polyfit(ns, np.array(ys, dtype=float), 1)
        return {"per_ciphertext": float(a), "constant": float(b)}
    out = {"ciphertexts": [int(x) for x in ns],
           "executing_ms": [p["reencrypt_ms"] + p["prove_ms"] for p in points], "verifying_ms": [p["verify_ms"] for p in points],
           "e(N)_ms": fit([p["reencrypt_ms"] + p["prove_ms"] for p in points]), "v(N)_ms": fit([p["verify_ms"] for p in points]),
           "shape": "e(N) = a N + b: shuffle (re-encrypt + permute) + prove; v(N): verify; arithmetic of the legs above (no hashing, no network)"}
    if all("proof_bytes" in p for p in points):
        out["proof_bytes"] = [p["proof_bytes"] for p in points]
        out["p(N)_bytes"] = fit(out["proof_bytes"])
    return out


def mix_prove_e2e(entry, vmn, ctx, grp, n: int, seed: int, sync):
    """ciphertexts/s of EVERYTHING PoSTW.prove + PoSTW.verify do around one shuffle except the network
    (ShufflerElGamalSession.java:400-409, 273-278; hvzk/PoSTW.java:95-165, 177-272; hvzk/ChallengerRO.java:96-116), width 1,
    n_e = n_v = 256, n_r = 100, C++ drivers:

      prover    draws pi, s, r, epsilon, b, beta (the N-sized arrays are expanded on the GPU from 32-byte seeds),
                re-encrypts, commits to the permutation, derives the batching seed = RO(prefix || node(g, h, u, pk, w, w'))
                -- 1.57 GB of byte trees framed on the GPU and hashed (SHA-256, one host core) by the helper thread
                while the protocol thread runs the seed-independent GPU work (re-encryption, precompute, commitPrepare) --
                commits, publishes the commitment as a byte tree, derives v = RO(prefix || node(seed, commitment)),
                replies, publishes the reply;
      verifier  (another party, run here after the prover) parses u, the commitment and the reply from their byte trees
                (range + subgroup membership on the GPU), derives the same seed and challenge by hashing the same bytes,
                computeAF beside the challenge hash, verify.

    Page-locked host buffers are session setup (allocated before the clock); the fixed-base tables of g and the key are built
    before the clock too, TIMED, and reported (setup_ms, total_ms_one_shot)."""
    import threading
    import torch
    (nat, rs), fs = modules(entry), load_sub(entry, "fiatshamir")
    NV = NE = 256
    NR = 100
    p, q, g = grp.p, grp.q, grp.g
    rnd = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    y = pow(g, rnd.ring_element(), p)
    pkey = [g, y]
    setup = session_setup(ctx, grp, [(g, 8), (y, 1)], n, sync)
    H = grp.exp(g, grp.ringArray(rnd.ring_array(n)))
    T = grp.ringArray(rnd.ring_array(n))
    M = grp.exp(g, grp.ringArray(rnd.ring_array(n)))
    YT = grp.exp(y, T)
    W = [grp.exp(g, T), M.mul(YT)]
    for a in (T, M, YT):
        a.free()
    chal = fs.Challenger(bytes(range(32)))                 # a 32-byte global prefix, as ProtocolElGamal.java:659-683 derives one
    tree = H.byteTreeSize()
    msg_bytes = 5 + 2 * tree + 8 * (5 + grp.nbytes) + 64
    com_buf = torch.empty(msg_bytes, dtype=torch.uint8).pin_memory()
    rep_buf = torch.empty(msg_bytes, dtype=torch.uint8).pin_memory()
    u_buf = torch.empty(tree, dtype=torch.uint8).pin_memory()
    com_view, rep_view = memoryview(com_buf.numpy()), memoryview(rep_buf.numpy())
    hasher_p = fs.InstanceHasher(ctx, chal.start(256), tree)       # threads + their page-locked buffers: session setup
    hasher_v = fs.InstanceHasher(ctx, chal.start(256), tree)
    gc.collect()
    sync()
    ctx.timing_reset()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    # ---------------------------------------------------------------- prover
    ctx.helper_mark()
    fs.hash_instance(hasher_p, grp, g, H, None, pkey, W, None)     # g, h: hashing starts now
    pi = rnd.permutation(n)
    S = [nat.random_ring_array_native(grp, rnd, n, NR)]             # ShufflerElGamalSession.java:408-409
    prover = nat.PoSBasicTW(grp, NV, NE, NR, rand=rnd)
    prover.precompute(g, H, pi)                                     # r, u, epsilon, A'
    hasher_p.mark()
    hasher_p.put_array(prover.u)
    hasher_p.put_bytes(fs.element_tree(grp, pkey))
    hasher_p.put_ciphertexts(W)
    WP = nat.reencrypt_native(grp, pkey, W, S, pi)
    hasher_p.mark()
    hasher_p.put_ciphertexts(WP)
    prover.setInstance(pkey, W, WP, S)
    prover.commitPrepare()                                          # b, beta, ..., F': no seed needed
    u_n = prover.u.toByteTreeInto(u_buf)                            # publish "PermutationCommitment"
    sync()
    t_gpu_indep = time.perf_counter()
    e_seed = chal.finish(hasher_p.finish(), 256)
    t_seed = time.perf_counter()
    prover.setBatchVectorSeed(e_seed)
    com = prover.commit()
    com_n = com.native.toByteTreeInto(com_buf)                      # publish "Commitment"
    t_com = time.perf_counter()
    d = chal.start(NV)
    d.update(b"\x00\x00\x00\x00\x02" + fs.leaf(e_seed))
    d.update(com_view[:com_n])
    v = int.from_bytes(chal.finish(d, NV), "big")
    t_chal = time.perf_counter()
    rep = prover.reply(v)
    rep_n = rep.native.toByteTreeInto(rep_buf)                      # publish "Reply"
    sync()
    t1 = time.perf_counter()
    # ---------------------------------------------------------------- verifier
    U = grp.toElementArrayFromByteTree(u_buf[:u_n])                 # setPermutationCommitment(reader): parse
    hasher_v.mark()
    fs.hash_instance(hasher_v, grp, g, H, U, pkey, W, WP)
    u_member = U.isMember()
    ver = nat.PoSBasicTW(grp, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(U)
    ver.setInstance(pkey, W, WP)
    com_in = ver.readCommitment((com_buf, com_n), n, 1)             # parse + membership while the seed is hashed
    rep_in = ver.readReply((rep_buf, rep_n), n, 1)
    seed_v = chal.finish(hasher_v.finish(), 256)
    t_vseed = time.perf_counter()
    ver.setBatchVectorSeed(seed_v)
    box = {}

    def challenge_thread():
        dv = chal.start(NV)
        dv.update(b"\x00\x00\x00\x00\x02" + fs.leaf(seed_v))
        dv.update(com_view[:com_n])
        box["v"] = int.from_bytes(chal.finish(dv, NV), "big")
    th = threading.Thread(target=challenge_thread)
    th.start()
    ver.computeAF()                                                 # beside the challenge hash (PoSTW.java:231-233)
    ver.setCommitment(com_in)
    ver.verifyPrepare(rep_in)                                       # the reply side of verify(): needs no challenge (vmn_pos_verify_prepare)
    th.join()
    ver.setChallenge(box["v"])
    ok = ver.verify(rep_in)
    sync()
    t2 = time.perf_counter()
    ctx.timing_enable(False)
    fam = ctx.timing_report()
    kernel_ms = sum(v[1] for v in fam.values())
    out = {"workload": "everything PoSTW.prove + PoSTW.verify do around one width-1 shuffle except the network: prover randomness "
                       "(N-sized draws expanded on the GPU from 32-byte seeds), re-encryption, proof, Fiat-Shamir hashing of the "
                       "instance and the commitment (SHA-256 on the host, overlapped with seed-independent GPU work on the helper "
                       "lane), byte trees published and parsed back (range + membership on the GPU), verification",
           "n": n, "accepted": bool(ok and u_member and seed_v == e_seed and box["v"] == v and com_in is not None and rep_in is not None),
           "prove_ms": (t1 - t0) * 1e3, "verify_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3,
           "ciphertexts_per_s": n / (t2 - t0), **one_shot_fields(n, setup, (t2 - t0) * 1e3),
           "ciphertexts_per_s_parties_in_parallel": n / max(t1 - t0, t2 - t1),
           "prover_phases_ms": {"seed_independent_gpu_work_done": (t_gpu_indep - t0) * 1e3, "seed_known": (t_seed - t0) * 1e3,
                                "commitment_published": (t_com - t0) * 1e3, "challenge_known": (t_chal - t0) * 1e3,
                                "reply_published": (t1 - t0) * 1e3},
           "verifier_phases_ms": {"seed_known": (t_vseed - t1) * 1e3, "verified": (t2 - t1) * 1e3},
           "hashed_bytes_per_party": hasher_p.bytes_hashed + com_n + 37,
           "instance_hash_thread_busy_ms": [hasher_p.busy_s * 1e3, hasher_v.busy_s * 1e3],
           "gpu_kernel_ms": kernel_ms, "roofline": leg_roofline(fam, (t2 - t0) * 1e3),
           "kernel_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
    for a in WP + S + [U, H] + W:
        a.free()
    com = rep = com_in = rep_in = None
    ver.free()
    prover.free()
    return out


def mix_ccpos(entry, vmn, ctx, n: int, seed: int, sync, bits: int = 3072, instrumented: bool = False):
    """BASELINE.json configs[2]: ElGamal ciphertexts over the 3072-bit ModPGroup (RFC 3526 group 15), width 1:
    offline  = permutation commitment (A4) + proof of a shuffle of commitments (A2, prove + verify)
    online   = re-encryption (A0) + commitment-consistent proof of a shuffle (A3, prove + verify, plain form).
    3072-bit elements run two lanes per element.  A pass creates its own group: the set-up of g and the
    key is rebuilt, timed and reported with every pass."""
    nat, rs = modules(entry)
    NV = NE = 256
    NR = 100
    p, q, g = load_sub(entry, "stdgroups").modp_group(bits)
    grp = vmn.ModPGroup(ctx, p, q, g, nbytes=bits // 8)
    bulk = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    y = pow(g, bulk.ring_element(), p)
    pkey = [g, y]
    setup = session_setup(ctx, grp, [(g, 6), (y, 1)], n, sync)
    H = grp.exp(g, grp.ringArray(bulk.ring_array(n)))
    T = grp.ringArray(bulk.ring_array(n))
    M = grp.exp(g, grp.ringArray(bulk.ring_array(n)))
    YT = grp.exp(y, T)
    W = [grp.exp(g, T), M.mul(YT)]
    for a in (T, M, YT):
        a.free()
    # the C++ drivers expand the provers' N-sized draws on the device (see mix_prove)
    tape = ReplayWithDeviceArrays(bulk, [("permutation", n), ("ring_array", n),          # pi, commitment exponents r (offline input)
                                         ("int_array", 1, 256),                          # seed of the PoSC batching vector
                                         ("ring_element",),                              # alpha (b, eps, beta: on the device)
                                         ("ring_element",), ("ring_element",), ("int_array", 1, NV),     # gamma, delta, v
                                         ("int_array", 1, 256), ("ring_element",), ("ring_element",),   # e seed, alpha, beta (s, eps: device)
                                         ("int_array", 1, NV)])
    ctx.timing_reset()
    ctx.timing_enable(instrumented)          # (per-launch event accounting only in the extra, instrumented pass: see mix_prove)
    gc.collect()            # release the previous pass's arrays into the pool before the clock starts
    sync()
    t0 = time.perf_counter()
    # ---- offline
    pi = tape.permutation(n)
    R = grp.ringArray(tape.ring_array(n))
    U = nat.permutation_commitment_native(grp, g, H, R, pi)
    e1 = bytes(tape.int_array(1, 256))[-32:]
    pr = nat.PoSCBasicTW(grp, NV, NE, NR, rand=tape)
    pr.setInstance(g, H, U, R, pi)
    pr.setBatchVectorSeed(e1)
    com = pr.commit()
    v1 = int.from_bytes(tape.int_array(1, NV), "big")
    rep = pr.reply(v1)
    ver = nat.PoSCBasicTW(grp, NV, NE, NR)
    ver.setInstance(g, H, U)
    ver.setBatchVectorSeed(e1)
    ver.setCommitment(com)
    ver.setChallenge(v1)
    ok_posc = ver.verify(rep)
    sync()
    t1 = time.perf_counter()
    # ---- online
    S = [nat.random_ring_array_native(grp, tape, n, NR)]
    WP = nat.reencrypt_native(grp, pkey, W, S, pi)
    sync()
    t2 = time.perf_counter()
    e2 = bytes(tape.int_array(1, 256))[-32:]
    cp = nat.CCPoSBasicW(grp, NV, NE, NR, rand=tape)
    cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVectorSeed(e2)
    com2 = cp.commit()
    v2 = int.from_bytes(tape.int_array(1, NV), "big")
    rep2 = cp.reply(v2)
    sync()
    t3 = time.perf_counter()
    cv = nat.CCPoSBasicW(grp, NV, NE, NR)
    cv.setInstance(g, H, U, pkey, W, WP)
    cv.setBatchVectorSeed(e2)
    cv.setCommitment(com2)
    cv.setChallenge(v2)
    cv.computeAB()
    ok = cv.verify(rep2)
    sync()
    t4 = time.perf_counter()
    ctx.timing_enable(False)
    fam = ctx.timing_report()
    online = t4 - t1
    pre = precomputed_factors_fields(nat, grp, pkey, W, S, pi, n, sync, t4 - t2)
    return {"workload": f"BASELINE.json configs[2]: ModPGroup {bits}-bit, width 1, CCPoS path; offline = permutation commitment + PoSC "
                        "prove+verify, online = re-encrypt + CCPoS prove+verify (n_e = n_v = 256, n_r = 100)",
            "n": n, "accepted": bool(ok and ok_posc),
            "offline_ms": (t1 - t0) * 1e3, "reencrypt_ms": (t2 - t1) * 1e3, "ccpos_prove_ms": (t3 - t2) * 1e3,
            "ccpos_verify_ms": (t4 - t3) * 1e3, "online_ms": online * 1e3, "total_ms": (t4 - t0) * 1e3,
            "ciphertexts_per_s_online": n / online, "ciphertexts_per_s_total": n / (t4 - t0),
            "setup": setup, **pre,
            # SURVEY.md §8d canonical BUDGET: 1090 M(96) per ciphertext online (M(96) = 18528 MAC; K2 at 384 products)
            "canonical_budget_TMACs_online_survey_8d": 1090 * 18528 * n / online / 1e12,      # NOT a roofline fraction: see `roofline`
            "roofline": leg_roofline(fam, (t4 - t0) * 1e3),          # offline + online: the counters cover the whole pass
            "_fam": fam, "kernel_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}


def mix_ec(entry, vmn, ctx, n: int, seed: int, sync, curve: str = "P-256", width: int = 3, instrumented: bool = False):
    """BASELINE.json configs[4] on one GPU: ElGamal ciphertexts over ECqPGroup P-256 (the reference's default group),
    width 3 (a ciphertext = 6 points): offline = permutation commitment; online = re-encryption (A0) +
    commitment-consistent proof of a shuffle (A3, prove + verify).  Point kernels: csrc/ec_kernels.h."""
    nat, rs = modules(entry)
    NV = NE = 256
    NR = 100
    grp = vmn.ECqPGroup(ctx, curve)
    g, q = grp.g, grp.q
    bulk = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    y = grp.k_exp(g, bulk.ring_element())
    pkey = [g] * width + [y] * width
    setup = session_setup(ctx, grp, [(g, 2 * width), (y, width)], n, sync)
    H = grp.exp(g, grp.ringArray(bulk.ring_array(n)))
    W = []
    Ts = [grp.ringArray(bulk.ring_array(n)) for _ in range(width)]
    for c in range(width):
        W.append(grp.exp(g, Ts[c]))
    for c in range(width):
        M = grp.exp(g, grp.ringArray(bulk.ring_array(n)))
        YT = grp.exp(y, Ts[c])
        W.append(M.mul(YT))
        M.free()
        YT.free()
    for t in Ts:
        t.free()
    plan = [("permutation", n), ("ring_array", n), ("int_array", 1, 256), ("ring_element",)] + [("ring_element",)] * width + \
           [("int_array", 1, NV)]
    tape = ReplayWithDeviceArrays(bulk, plan)
    ctx.timing_reset()
    ctx.timing_enable(instrumented)          # (see mix_prove)
    gc.collect()            # release the previous pass's arrays into the pool before the clock starts
    sync()
    t0 = time.perf_counter()
    pi = tape.permutation(n)
    R = grp.ringArray(tape.ring_array(n))
    U = nat.permutation_commitment_native(grp, g, H, R, pi)
    sync()
    t1 = time.perf_counter()
    S = [nat.random_ring_array_native(grp, tape, n, NR) for _ in range(width)]
    WP = nat.reencrypt_native(grp, pkey, W, S, pi)
    sync()
    t2 = time.perf_counter()
    e = bytes(tape.int_array(1, 256))[-32:]            # seed of the batching vector
    cp = nat.CCPoSBasicW(grp, NV, NE, NR, rand=tape)
    cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVectorSeed(e)
    com = cp.commit()
    v = int.from_bytes(tape.int_array(1, NV), "big")
    rep = cp.reply(v)
    sync()
    t3 = time.perf_counter()
    cv = nat.CCPoSBasicW(grp, NV, NE, NR)
    cv.setInstance(g, H, U, pkey, W, WP)
    cv.setBatchVectorSeed(e)
    cv.setCommitment(com)
    cv.setChallenge(v)
    cv.computeAB()
    ok = cv.verify(rep)
    sync()
    t4 = time.perf_counter()
    ctx.timing_enable(False)
    fam = ctx.timing_report()
    online = t4 - t1
    pre = precomputed_factors_fields(nat, grp, pkey, W, S, pi, n, sync, t4 - t2)
    return {"workload": f"BASELINE.json configs[4] on one GPU: ECqPGroup {curve}, width {width}; offline = permutation commitment, "
                        "online = re-encrypt + CCPoS prove+verify (n_e = n_v = 256, n_r = 100)",
            "n": n, "accepted": bool(ok), "offline_ms": (t1 - t0) * 1e3, "reencrypt_ms": (t2 - t1) * 1e3,
            "ccpos_prove_ms": (t3 - t2) * 1e3, "ccpos_verify_ms": (t4 - t3) * 1e3, "online_ms": online * 1e3, "total_ms": (t4 - t0) * 1e3,
            "ciphertexts_per_s_online": n / online,
            "setup": setup, **pre,
            "roofline": leg_roofline(fam, (t4 - t0) * 1e3),
            "_fam": fam, "kernel_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}


def mean_pass(runs, n: int, keys, rate_key: str, rate_name: str, instrumented: dict = None, fam=None, first: dict = None) -> dict:
    """Fold the passes of a leg whose every pass is a cold start (own group, own tables): the MEAN of the timed keys, all
    passes listed, and the one-shot figure (mean set-up + mean).  The per-family breakdown and the roofline's work counts come
    from `instrumented`, one more pass run with the library's per-launch event accounting on (its own wall clock is reported
    as instrumented_pass_ms, not used): the work of a pass against the mean wall clock of the timed passes."""
    out = dict(min(runs, key=lambda r: abs(r[rate_key] - mean_of(runs, rate_key))))
    for k in keys:
        out[k] = mean_of(runs, k)
    if instrumented is not None:
        out["kernel_ms_by_family"] = instrumented["kernel_ms_by_family"]
        out["roofline"] = leg_roofline(instrumented["_fam"], out["total_ms"])
        out["instrumented_pass_ms"] = instrumented[rate_key]
    out.pop("_fam", None)
    out[f"passes_{rate_key}"] = [round(r[rate_key], 2) for r in runs]
    out["passes_setup_ms"] = [round(r["setup"]["setup_ms"], 2) for r in runs]
    out["statistic"] = f"mean of {len(runs)} passes, each from a cold group (tables rebuilt)"
    if first is not None:
        # The first pass of a leg in this process is not timed and is reported beside the figure: it finds the array pool full of
        # the previous leg's blocks (other sizes), so its allocations go through hipFree + hipMalloc -- +25 ms on a 57 ms pass
        # over P-256, +100 ms at 3072 bits -- an artefact of running five legs of different shapes in one process, the
        # counterpart of the headline's warm-up steps.  Set-up (the fixed-base tables) is rebuilt and timed in every pass.
        out[f"first_pass_{rate_key}"] = first[rate_key]
        out["first_pass_accepted"] = bool(first["accepted"])
        out["statistic"] += (f"; after one untimed pass (first_pass_{rate_key}: the process's first pass at these array sizes refills "
                             "the array pool)")
    out["accepted"] = all(r["accepted"] for r in runs)
    out[rate_name] = n / (out[rate_key] / 1e3)
    setup = dict(runs[0]["setup"])
    setup["setup_ms"] = sum(r["setup"]["setup_ms"] for r in runs) / len(runs)
    out.update(one_shot_fields(n, setup, out[rate_key], rate_key))
    return out


def _synthetic_instance(grp, rs, seed: int, n: int, width: int = 1):
    """Public instance of a sharded leg, the same on every rank and made on the device (untimed setup): h = g^a,
    key y = g^x, honest ciphertexts w = (g^t, g^m y^t) per column -- exponent arrays expanded from 32-byte seeds."""
    q, g = grp.q, grp.g
    pub = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    y = grp.k_exp(g, pub.ring_element())
    pkey = [g] * width + [y] * width
    rnd_arr = lambda: grp.ringArrayFromPRG(pub.array_seed(), n, q.bit_length() - 1)
    A = rnd_arr()
    H = grp.exp(g, A)
    A.free()
    W = [None] * (2 * width)
    for c in range(width):
        T, Mx = rnd_arr(), rnd_arr()
        M, YT = grp.exp(g, Mx), grp.exp(y, T)
        W[c] = grp.exp(g, T)
        W[width + c] = M.mul(YT)
        for a in (T, Mx, M, YT):
            a.free()
    return pub, y, pkey, H, W


def _collective_record(dist, comm, ncomm):
    return {"backend": comm.backend_used or (dist.get_backend() if dist.is_initialized() else None), "world": comm.world,
            "fell_back": comm.fell_back, "all_gathers_per_proof_leg": ncomm.exchanges, "bytes_sent_per_rank": ncomm.bytes_sent}


def mix_prove_sharded(entry, vmn, ctx, grp, n: int, seed: int, sync, comm):
    """ONE proof of a shuffle of n ciphertexts sharded over the ranks through the C++ drivers (vmn_pos_set_comm,
    include/vmnproofs.h): every position-indexed array is split by position; the public inputs h and w are whole on
    every GPU (the permuted arrays are local gathers, no element crosses a link); the prover's N-sized random arrays, the
    re-encryption exponents and the batching vector are 32-byte seeds of which a rank expands only its own positions and
    the rows it reads through the permutation (O(n / world) per rank); the only collectives are all-gathers of a few
    hundred bytes (partial products, partial sums, scan carries, verdict bits; RCCL over xGMI) -- one per phase."""
    nat, rs = modules(entry)
    ncomm = nat.NativeComm(comm)
    NV = NE = 256
    NR = 100
    lo, hi = nat.shard_bounds_native(n, comm.world, comm.rank)
    pub, y, pkey, H, W = _synthetic_instance(grp, rs, seed, n)
    g = grp.g
    # every rank builds the tables of g and the key for ITS shard: timed, the slowest rank's figure is reported
    setup = session_setup(ctx, grp, [(g, 8), (y, 1)], max(1, hi - lo), sync)
    runs = []
    for pass_no in range(4):                   # one untimed pass (the array pool, see mean_pass), two timed, one instrumented (see mix_prove)
        instrumented = pass_no == 3
        ncomm.exchanges = ncomm.bytes_sent = 0
        ctx.timing_reset()
        ctx.timing_enable(instrumented)
        gc.collect()            # release the previous pass's arrays into the pool before the clock starts
        sync()
        t0 = time.perf_counter()
        pi = pub.permutation(n)                                                # the same on every rank
        prover = nat.PoSBasicTW(grp, NV, NE, NR, rand=pub)
        prover.setComm(ncomm)
        prover.precompute(g, H, pi)
        WP, S = nat.reencrypt_shard_seeded_native(grp, pkey, W, pub, NR, pi, lo, hi)   # this rank's shards of w' and s
        sync()
        t1 = time.perf_counter()
        prover.setInstance(pkey, W, WP, S)
        e_seed = pub.array_seed()
        prover.setBatchVectorSeed(e_seed)
        com = prover.commit()
        v = int.from_bytes(pub.int_array(1, NV), "big")
        rep = prover.reply(v)
        sync()
        t2 = time.perf_counter()
        ver = nat.PoSBasicTW(grp, NV, NE, NR)
        ver.setComm(ncomm)
        ver.precompute(g, H)
        ver.setPermutationCommitment(prover.u)
        ver.setInstance(pkey, W, WP)
        ver.setBatchVectorSeed(e_seed)
        ver.computeAF()
        ver.setCommitment(com)
        ver.setChallenge(v)
        ok = ver.verify(rep)
        sync()
        t3 = time.perf_counter()
        ctx.timing_enable(False)
        fam = ctx.timing_report()
        cur = {"kernel_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
               "precompute_and_reencrypt_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "verify_ms": (t3 - t2) * 1e3,
               "total_ms": (t3 - t0) * 1e3, "accepted": bool(ok), "n": n, "ciphertexts_per_rank": hi - lo,
               "collective": _collective_record(comm.dist, comm, ncomm),
               "roofline_rank0": leg_roofline(fam, (t3 - t0) * 1e3)}
        com = rep = None
        for a in WP + S:
            a.free()
        ver.free()
        prover.free()
        runs.append(cur)
    for a in [H] + W:
        a.free()
    best = dict(runs[-1])                      # the instrumented pass carries the per-family detail ...
    best["instrumented_pass_ms"] = runs[-1]["total_ms"]
    best["first_pass_total_ms_rank0"] = runs[0]["total_ms"]
    runs = runs[1:-1]                          # ... the figures are the timed passes'
    best["passes_total_ms_rank0"] = [round(r["total_ms"], 2) for r in runs]
    best["statistic"] = "mean of 2 passes (event accounting off) after one untimed pass (first_pass_*); per pass and for the set-up the slowest rank counts"
    best["total_ms"] = sum(comm.max_over_ranks(r["total_ms"]) for r in runs) / len(runs)
    best["accepted"] = comm.all_true(all(r["accepted"] for r in runs))
    best["ciphertexts_per_s"] = n / (best["total_ms"] / 1e3)
    setup["setup_ms"] = comm.max_over_ranks(setup["setup_ms"])
    best.update(one_shot_fields(n, setup, best["total_ms"]))
    return best


def mix_ccpos_sharded(entry, vmn, ctx, grp, label: str, n: int, seed: int, sync, comm, width: int, with_posc: bool):
    """The commitment-consistent path sharded over the ranks (BASELINE.json configs[4]: P-256, width 3, on 8 GPUs;
    configs[2]'s 3072-bit group the same way): offline = this rank's shard of the permutation commitment (+ a sharded
    PoSC prove + verify when with_posc), online = this rank's shard of the re-encryption + CCPoS prove + verify through
    vmn_ccpos_set_comm (hvzk/CCPoSBasicW.java:344-396, 462-506, 519-584).  Exponent arrays are seeds (see mix_prove_sharded)."""
    nat, rs = modules(entry)
    ncomm = nat.NativeComm(comm)
    NV = NE = 256
    NR = 100
    lo, hi = nat.shard_bounds_native(n, comm.world, comm.rank)
    pub, y, pkey, H, W = _synthetic_instance(grp, rs, seed, n, width)
    g = grp.g
    setup = session_setup(ctx, grp, [(g, 2 * width + (4 if with_posc else 0)), (y, width)], max(1, hi - lo), sync)
    runs = []
    for pass_no in range(4):                   # one untimed pass, two timed passes, then one instrumented (see mix_prove_sharded)
        instrumented = pass_no == 3
        ncomm.exchanges = ncomm.bytes_sent = 0
        ctx.timing_reset()
        ctx.timing_enable(instrumented)
        gc.collect()
        sync()
        t0 = time.perf_counter()
        pi = pub.permutation(n)
        U, R = nat.permutation_commitment_shard_seeded_native(grp, g, H, pub, NR, pi, lo, hi)
        ok_posc = True
        if with_posc:
            e1 = pub.array_seed()
            pr = nat.PoSCBasicTW(grp, NV, NE, NR, rand=pub)
            pr.setComm(ncomm)
            pr.setInstance(g, H, U, R, pi)
            pr.setBatchVectorSeed(e1)
            com = pr.commit()
            v1 = int.from_bytes(pub.int_array(1, NV), "big")
            rep = pr.reply(v1)
            pv = nat.PoSCBasicTW(grp, NV, NE, NR)
            pv.setComm(ncomm)
            pv.setInstance(g, H, U)
            pv.setBatchVectorSeed(e1)
            pv.setCommitment(com)
            pv.setChallenge(v1)
            ok_posc = pv.verify(rep)
            com = rep = None
            pv.free()
            pr.free()
        sync()
        t1 = time.perf_counter()
        WP, S = nat.reencrypt_shard_seeded_native(grp, pkey, W, pub, NR, pi, lo, hi)
        sync()
        t2 = time.perf_counter()
        e2 = pub.array_seed()
        cp = nat.CCPoSBasicW(grp, NV, NE, NR, rand=pub)
        cp.setComm(ncomm)
        cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
        cp.setBatchVectorSeed(e2)
        com2 = cp.commit()
        v2 = int.from_bytes(pub.int_array(1, NV), "big")
        rep2 = cp.reply(v2)
        sync()
        t3 = time.perf_counter()
        cv = nat.CCPoSBasicW(grp, NV, NE, NR)
        cv.setComm(ncomm)
        cv.setInstance(g, H, U, pkey, W, WP)
        cv.setBatchVectorSeed(e2)
        cv.setCommitment(com2)
        cv.setChallenge(v2)
        cv.computeAB()
        ok = cv.verify(rep2)
        sync()
        t4 = time.perf_counter()
        ctx.timing_enable(False)
        fam = ctx.timing_report()
        cur = {"workload": label, "n": n, "ciphertexts_per_rank": hi - lo, "accepted": bool(ok and ok_posc),
               "offline_ms": (t1 - t0) * 1e3, "reencrypt_ms": (t2 - t1) * 1e3, "ccpos_prove_ms": (t3 - t2) * 1e3,
               "ccpos_verify_ms": (t4 - t3) * 1e3, "online_ms": (t4 - t1) * 1e3, "total_ms": (t4 - t0) * 1e3,
               "collective": _collective_record(comm.dist, comm, ncomm),
               "roofline_rank0": leg_roofline(fam, (t4 - t0) * 1e3),
               "kernel_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
        com2 = rep2 = None
        for a in WP + S + [U, R]:
            a.free()
        cv.free()
        cp.free()
        runs.append(cur)
    for a in [H] + W:
        a.free()
    best = dict(runs[-1])
    best["instrumented_pass_ms"] = runs[-1]["online_ms"]
    best["first_pass_online_ms_rank0"] = runs[0]["online_ms"]
    runs = runs[1:-1]
    best["passes_online_ms_rank0"] = [round(r["online_ms"], 2) for r in runs]
    best["statistic"] = "mean of 2 passes (event accounting off) after one untimed pass (first_pass_*); per pass and for the set-up the slowest rank counts"
    best["online_ms"] = sum(comm.max_over_ranks(r["online_ms"]) for r in runs) / len(runs)
    best["total_ms"] = sum(comm.max_over_ranks(r["total_ms"]) for r in runs) / len(runs)
    best["accepted"] = comm.all_true(all(r["accepted"] for r in runs))
    best["ciphertexts_per_s_online"] = n / (best["online_ms"] / 1e3)
    best["ciphertexts_per_s_total"] = n / (best["total_ms"] / 1e3)
    setup["setup_ms"] = comm.max_over_ranks(setup["setup_ms"])
    best.update(one_shot_fields(n, setup, best["online_ms"], "online_ms"))
    return best


def decrypt_leg(entry, vmn, ctx, grp, n: int, seed: int, sync, k: int = 3, threshold: int = 2):
    """ciphertexts/s of the decryption half of `vmn -mix` as ONE party (j = 1 of k, threshold t) runs it, width 1, device-
    resident arrays (the other parties' factors and proofs are inputs, made before the clock):
      own decryption factors f_j = u^(-x_j / c)                       elgamal/DistrElGamalSession.java:365-385
      own batched proof: A = prod u^e, y' = g^r, B' = A^r, reply      DistrElGamalSessionBasic.java:513-540, 595-598
      check of every other party: B_l = prod f_l^e and the two equations   :642-727
      combination f = prod_l f_l^(lambda_l) with the modified Lagrange integers (possibly negative)   :406-452, 465-503
      plaintexts m = v f                                              DistrElGamalSession.java:536-538
    n_e = n_v = 256; the batching vector is expanded on the device from a 32-byte seed."""
    nat, rs = modules(entry)
    NE = NV = 256
    p, q, g = grp.p, grp.q, grp.g
    rnd = rs.InsecureBulkRandomSource(seed, q, grp.exp_bytes)
    coeffs = [rnd.ring_element() for _ in range(threshold)]              # Shamir sharing of the key over Z_q
    share = lambda j: sum(c * pow(j, d, q) for d, c in enumerate(coeffs)) % q
    xs = [None] + [share(j) for j in range(1, k + 1)]
    ys = [None] + [pow(g, xj, p) for xj in xs[1:]]
    y = pow(g, coeffs[0], p)
    arr = lambda: grp.ringArrayFromPRG(rnd.array_seed(), n, q.bit_length() - 1)
    T, Mx = arr(), arr()     # (the instance below is made with whatever tables the group has; the decrypting party itself uses
                             # no N-sized fixed-base power: its set-up is the group's constants, nothing to time)
    U = grp.exp(g, T)
    M, YT = grp.exp(g, Mx), grp.exp(y, T)
    V = M.mul(YT)
    for a in (T, Mx, YT):
        a.free()
    j = 1
    e_seed = rnd.array_seed()
    v = int.from_bytes(rnd.int_array(1, NV), "big")
    # the other parties' work (inputs of party j): their factors, commitments and replies
    F = [None] * (k + 1)
    others = {}
    for l in range(2, k + 1):
        F[l] = nat.decryptionFactors(U, xs[l], q, k)
    for l in range(2, k + 1):
        pr = nat.DistrElGamalSessionBasic(grp, l, k, threshold, NE, rand=rnd)
        fl = [None] * (k + 1)
        fl[l] = F[l]
        pr.setInstance(U, ys, fl)
        pr.setBatchVectorSeed(e_seed)
        pr.batchInput()
        yp, Bp = pr.commit(xs[l])
        others[l] = (yp, Bp, pr.reply(v))
        pr.free()
    runs = []
    for _ in range(2):
        ctx.timing_reset()
        ctx.timing_enable(True)
        gc.collect()
        sync()
        t0 = time.perf_counter()
        F[j] = nat.decryptionFactors(U, xs[j], q, k)
        sync()
        t1 = time.perf_counter()
        me = nat.DistrElGamalSessionBasic(grp, j, k, threshold, NE, rand=rnd)
        me.setInstance(U, ys, F)
        me.setBatchVectorSeed(e_seed)
        me.batchInput()
        yp, Bp = me.commit(xs[j])
        kx = me.reply(v)
        sync()
        t2 = time.perf_counter()
        verdicts = [False] * (k + 1)
        for l in range(2, k + 1):
            me.setCommitment(l, others[l][0], others[l][1])
            me.setReply(l, others[l][2])
            me.batch(l)
            verdicts[l] = me.verify(l, v)
        sync()
        t3 = time.perf_counter()
        correct = [False] + [True] * k
        comb = nat.combineDecryptionFactors(F, correct, k, threshold, q)
        plain = nat.plaintexts(V, comb)
        sync()
        t4 = time.perf_counter()
        ctx.timing_enable(False)
        fam = ctx.timing_report()
        ok = all(verdicts[2:]) and plain.equals(M)
        cur = {"n": n, "parties": k, "threshold": threshold, "accepted_and_plaintexts_recovered": bool(ok),
               "own_factors_ms": (t1 - t0) * 1e3, "own_proof_ms": (t2 - t1) * 1e3, "verify_others_ms": (t3 - t2) * 1e3,
               "combine_and_plaintexts_ms": (t4 - t3) * 1e3, "total_ms": (t4 - t0) * 1e3, "roofline": leg_roofline(fam, (t4 - t0) * 1e3),
               "kernel_ms_by_family": {kk: round(vv[1], 3) for kk, vv in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
        for a in (F[j], comb, plain):
            a.free()
        me.free()
        runs.append(cur)
    for a in [U, V, M] + [F[l] for l in range(2, k + 1)]:
        a.free()
    best = dict(min(runs, key=lambda r: abs(r["total_ms"] - mean_of(runs, "total_ms"))))
    for key in ("own_factors_ms", "own_proof_ms", "verify_others_ms", "combine_and_plaintexts_ms", "total_ms"):
        best[key] = mean_of(runs, key)
    best["passes_total_ms"] = [round(r["total_ms"], 2) for r in runs]
    best["statistic"] = "mean of 2 passes"
    best["accepted_and_plaintexts_recovered"] = all(r["accepted_and_plaintexts_recovered"] for r in runs)
    best["ciphertexts_per_s"] = n / (best["total_ms"] / 1e3)
    best.update(one_shot_fields(n, {"setup_ms": 0.0, "fixed_tables": 0, "table_bytes": 0, "uses_hint": [],
                                    "note": "a decrypting party raises the ciphertexts to ITS secret exponent (variable bases) and runs "
                                            "multi-exponentiations: no long-lived fixed base of array size, so no table set-up"}, best["total_ms"]))
    best["workload"] = (f"verifiable threshold decryption as one of k = {k} parties (threshold {threshold}), ModPGroup 2048-bit, width 1: own "
                        "factors u^(-x_j/c), own batched proof, check of the other parties' proofs, combination with the modified "
                        "Lagrange integers, plaintexts (n_e = n_v = 256)")
    return best


def cpu_decrypt(p, q, g, n: int, cores: int, k: int = 3, threshold: int = 2):
    """CPU baseline of decrypt_leg (test infrastructure, never the product): the same op sequence on the C + GMP oracle
    over `cores` OpenMP threads -- mpz_powm per element for the factors, Pippenger on GMP for the batches, the combination
    with the (small) Lagrange integers by mpz_powm; single elements are Python integers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyref_proofs as P
    from oracle.cbind import Oracle
    from tape import Tape
    orc = Oracle(p, q)
    orc.set_threads(cores)
    NE = NV = 256
    cpip = max(4, min(12, n.bit_length() - 3))
    t = Tape(b"cpu-dec", q)
    coeffs = t.ring_array(threshold)
    share = lambda j: sum(c * pow(j, d, q) for d, c in enumerate(coeffs)) % q
    xs = [None] + [share(j) for j in range(1, k + 1)]
    y = pow(g, coeffs[0], p)
    tt = t.ring_array(n)
    u = orc.exp_fixed(g, tt)
    msgs = orc.exp_fixed(g, t.ring_array(n))
    vv = orc.mul(msgs, orc.exp_fixed(y, tt))
    e, chal = t.int_array(n, NE), t.int_array(1, NV)[0]
    cinv = pow(P.prod_factor(q, k), -1, q)
    fexp = lambda l: (-xs[l]) * cinv % q
    f = [None, None] + [orc.exp_scalar(u, fexp(l)) for l in range(2, k + 1)]
    ys = [None] + [pow(g, x, p) for x in xs[1:]]
    A_o = orc.exp_prod(u, e, ebits=NE, pippenger_c=cpip)
    others = {}
    for l in range(2, k + 1):
        r = t.ring_element()
        others[l] = (pow(g, r, p), pow(A_o, r, p), (fexp(l) * chal + r) % q)
    t0 = time.perf_counter()
    f[1] = orc.exp_scalar(u, fexp(1))
    A = orc.exp_prod(u, e, ebits=NE, pippenger_c=cpip)
    r1 = t.ring_element()
    yp1, Bp1, k1 = pow(g, r1, p), pow(A, r1, p), (fexp(1) * chal + r1) % q
    ok = True
    for l in range(2, k + 1):
        Bl = orc.exp_prod(f[l], e, ebits=NE, pippenger_c=cpip)
        ypl, Bpl, kl = others[l]
        ok = ok and pow(pow(ys[l], -1, p), cinv * chal % q, p) * ypl % p == pow(g, kl, p) and pow(Bl, chal, p) * Bpl % p == pow(A, kl, p)
    ints = P.lagrange_integers(q, [False] + [True] * k, k, threshold)
    comb = None
    for l, c in zip(range(1, threshold + 1), ints):
        term = orc.exp_scalar(f[l], abs(c))
        if c < 0:
            term = [pow(x, -1, p) for x in term]
        comb = term if comb is None else orc.mul(comb, term)
    plain = orc.mul(vv, comb)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "ciphertexts/s", "cores": cores, "kind": "port", "accepted_and_plaintexts_recovered": bool(ok and plain == msgs),
            "sample": f"{n} ciphertexts, 2048-bit group, k = {k}, threshold {threshold}: own factors (mpz_powm per element), batches "
                      "(Pippenger on GMP), checks, combination, plaintexts; OpenMP static chunks"}


def cpu_mix_prove(p, q, g, n: int, cores: int):
    """CPU baseline of the mix + prove leg (test infrastructure, never the product): the reference's op sequence
    (oracle/pyref_proofs.py: re-encrypt, PoS prove, PoS verify) with every array operation in the C + GMP oracle over
    `cores` OpenMP threads.  Per-element exponentiations are mpz_powm, fixed-base ones go through a precomputed table whose
    window is sized for the array (orc_exp_fixed_table: what VCR + GMPMEE's fpowm do), the multi-exponentiations are a
    Pippenger on GMP; single elements are Python integers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyref_proofs as P
    from oracle.cbind import GmpAdapter, Oracle
    from tape import Tape
    orc = Oracle(p, q)
    orc.set_threads(cores)
    K = GmpAdapter(orc, pippenger_c=max(4, min(12, n.bit_length() - 3)), fixed_tables=True)
    NV = NE = 256
    NR = 100
    t = Tape(b"cpu-mix", q)
    h = orc.exp_fixed(g, t.ring_array(n))
    y = pow(g, t.ring_element(), p)
    pkey = [g, y]
    w = [orc.exp_fixed(g, t.ring_array(n)), orc.exp_fixed(y, t.ring_array(n))]
    pi, s, e, v = t.permutation(n), [t.ring_array(n)], t.int_array(n, NE), t.int_array(1, NV)[0]
    t0 = time.perf_counter()
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    pr = P.GPoS(K, NV, NE, NR, rand=Tape(b"cpu-prover", q))
    pr.precompute(g, h, pi)
    pr.setInstance(pkey, w, wp, s)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    ver = P.GPoS(K, NV, NE, NR)
    ver.precompute(g, h)
    ver.u = pr.u
    ver.setInstance(pkey, w, wp)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ok = ver.verify(rep, v)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "ciphertexts/s", "cores": cores, "kind": "port", "accepted": bool(ok),
            "sample": f"{n} ciphertexts, 2048-bit group, width 1: re-encrypt + PoS prove + verify; GMP: mpz_powm per element, "
                      "fixed-base tables (window sized for the array, rebuilt per call), Pippenger, OpenMP static chunks"}


def canonical_modpow_macs(nbits: int, t: int) -> int:
    """fixed-window modpow of a t-bit exponent, s = nbits / 32 limbs, canonical window (5 for t >= 1024, 4
    otherwise): t Q(s) + (ceil(t / w) + 2^w) M(s)."""
    s32 = nbits // 32
    M, Q = 2 * s32 * s32 + s32, s32 * (s32 + 1) // 2 + s32 * s32 + s32
    w = 5 if t >= 1024 else 4
    return t * Q + (-(-t // w) + (1 << w)) * M


def modexp_by_shape(entry, vmn, ctx, p, q, g, n: int, sync, cores: int, with_cpu: bool) -> dict:
    """modexps/s of the SHAPES a proof of a shuffle is made of (SURVEY.md §8a totals; north_star's >= 1e7 modexps/s is about
    these, not about 2047-bit per-element exponents, whose ceiling at 100 % of the VALU roof is 2.4e6/s): 2048-bit group,
    n elements per call, device-resident, mean of two calls after one warm-up.
      K1a_256 / K1a_612   X.exp(E): per-element exponents of 256 / 612 bits   (PoSBasicTW.java:1032: B^k_E)
      K1b_256             X.exp(v): one 256-bit exponent for the array         (:1028: B^v)
      K2_full             g.exp(E): fixed base, full-length exponents          (:447, 606, 608, 644, 646; the table is warm,
                          its build is reported beside it)
      K3_256 / K3_612     X.expProd(E) per TERM, 256- / 612-bit exponents      (:408-409, 481, 690, 1021, 1063)
    frac_canonical_budget prices an op at SURVEY.md §8d's canonical count (what the headline's frac does for K1a at 2047 bits);
    frac_canonical_executed prices the products the kernel actually does (fewer for K2 / K3: larger tables, wider windows).
    gmp: the same op on the C + GMP oracle over `cores` threads on a sample."""
    nat, rs = modules(entry)
    nbytes = 256
    grp = vmn.ModPGroup(ctx, p, q, g, nbytes=nbytes)
    rnd = rs.InsecureBulkRandomSource(31337, q, grp.exp_bytes)
    xb, _ = make_inputs(n, 424242, nbytes)
    X = grp.toElementArray(xb)
    E = {bits: grp.ringArray(rnd.int_array(n, bits)) for bits in (256, 612)}
    Efull = grp.ringArray(rnd.ring_array(n))
    v256 = int.from_bytes(rnd.int_array(1, 256), "big") | (1 << 255)
    M64 = 2 * 64 * 64 + 64
    setup = session_setup(ctx, grp, [(g, 8)], n, sync)
    shapes = {
        "K1a_256": (lambda: X.exp(E[256], 256), canonical_modpow_macs(2048, 256), "modpow"),
        "K1a_612": (lambda: X.exp(E[612], 612), canonical_modpow_macs(2048, 612), "modpow"),
        "K1b_256": (lambda: X.exp(v256), canonical_modpow_macs(2048, 256), "modpow"),
        "K2_full": (lambda: grp.exp(g, Efull), 256 * M64, "fixed"),
        "K3_256": (lambda: X.expProd(E[256], 256), 18 * M64, "expprod"),
        "K3_612": (lambda: X.expProd(E[612], 612), 44 * M64, "expprod"),
    }
    out = {"n": n, "group": "RFC 3526 group 14 (2048 bits)", "fixed_base_table": setup,
           "unit": "ops/s (K3: exponentiated-and-multiplied terms/s)"}
    for name, (fn, canon_budget, _fam) in shapes.items():
        for _ in range(2):                                      # warm-up (the first shape also settles the pool after the set-up)
            r = fn()
            if hasattr(r, "free"):
                r.free()
        ctx.timing_reset()
        ctx.timing_enable(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            r = fn()
            if hasattr(r, "free"):
                r.free()
        sync()
        dt = (time.perf_counter() - t0) / 2
        ctx.timing_enable(False)
        fam = ctx.timing_report()
        canon_exec = sum(v[3] for v in fam.values()) / 2
        out[name] = {"ms": dt * 1e3, "ops_per_s": n / dt, "canonical_macs_per_op_survey_8d": canon_budget,
                     "frac_canonical_budget": canon_budget * n / dt / 1e12 / PEAK_TMACS,
                     "frac_canonical_executed": canon_exec / dt / 1e12 / PEAK_TMACS,
                     "frac_executed": sum(v[2] for v in fam.values()) / 2 / dt / 1e12 / PEAK_TMACS,
                     "kernel_ms": sum(v[1] for v in fam.values()) / 2}
    if with_cpu:
        from oracle.cbind import Oracle
        orc = Oracle(p, q, nbytes)
        orc.set_threads(cores)
        m = 1500 * cores                                        # ~1-3 s of GMP per shape
        xs_b = xb[: m * nbytes]
        e_b = {bits: bytes(E[bits].copyOfRange(0, m).toBytes()) for bits in (256, 612)}
        ef_b = bytes(Efull.copyOfRange(0, m).toBytes())
        xb_w = grp.exp_bytes

        def timed(fn):
            t0 = time.perf_counter()
            fn()
            return m / (time.perf_counter() - t0)
        out["K1a_256"]["gmp_ops_per_s"] = timed(lambda: orc.exp_array_bytes(xs_b, e_b[256], m, xb_w))
        out["K1a_612"]["gmp_ops_per_s"] = timed(lambda: orc.exp_array_bytes(xs_b, e_b[612], m, xb_w))
        v_rep = v256.to_bytes(xb_w, "big") * m
        out["K1b_256"]["gmp_ops_per_s"] = timed(lambda: orc.exp_array_bytes(xs_b, v_rep, m, xb_w))
        gb = g.to_bytes(nbytes, "big")
        out["K2_full"]["gmp_ops_per_s"] = timed(lambda: orc.exp_fixed_table_bytes(gb, ef_b, m, xb_w, q.bit_length(), 0))
        cpip = max(4, min(12, m.bit_length() - 3))
        out["K3_256"]["gmp_ops_per_s"] = timed(lambda: orc.expprod_pippenger_bytes(xs_b, e_b[256], m, xb_w, 256, cpip))
        out["K3_612"]["gmp_ops_per_s"] = timed(lambda: orc.expprod_pippenger_bytes(xs_b, e_b[612], m, xb_w, 612, cpip))
        out["gmp"] = {"cores": cores, "sample": f"first {m} elements per shape (mpz_powm per element; fixed-base table; Pippenger on GMP)"}
    for a in [X, Efull] + list(E.values()):
        a.free()
    grp.close()
    return out


def gpu_runtime_loaded() -> bool:
    """True when this process has already mapped the HIP runtime or a profiler's preloaded library (rocprofv3 initialises
    the GPU before the program starts): such a process must not spawn compilers or launchers on this pool."""
    try:
        with open("/proc/self/maps") as f:
            maps = f.read()
    except OSError:
        return False
    return "libamdhip64" in maps or "librocprofiler" in maps


def ensure_built(entry) -> None:
    """The three libraries, built here only when missing AND nothing in this process has touched the GPU (the round-end
    driver and tools/*.sh build beforehand; under rocprofv3 a missing library is an error, not a reason to run hipcc)."""
    need = [entry.LIB, entry.PROOFS_LIB, os.path.join(ROOT, "oracle", "libvmnoracle.so")]
    missing = [p for p in need if not os.path.exists(p)]
    if not missing:
        return
    if gpu_runtime_loaded():
        print("bench.py: " + ", ".join(os.path.relpath(m, ROOT) for m in missing) + " missing and the GPU runtime is already loaded in this "
              "process: run `python3 __graft_entry__.py` first", file=sys.stderr)
        sys.exit(2)
    entry.build()


def visible_gpu_count() -> int:
    """GPUs this process would see, counted WITHOUT loading any GPU runtime: the KFD topology (nodes with SIMDs are GPUs)
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set."""
    total = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in sorted(os.listdir(base)):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    total += 1
            except (OSError, ValueError):
                continue
    except OSError:
        total = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = [x for x in v.split(",") if x.strip() != ""]
            total = min(total, len(listed)) if total else len(listed)
    return total


def self_launch(args) -> int:
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (one process per GPU,
    `python -m torch.distributed.run`, rendezvous on 127.0.0.1) before anything in this process touches the GPU, relay
    its output (rank 0 prints the JSON line) and return its exit code.  With fewer visible GPUs than ranks the run is a
    rehearsal: the ranks share the GPUs and gloo carries the collectives (RCCL needs one GPU per rank) -- the line says so."""
    import socket
    import subprocess
    if gpu_runtime_loaded():
        # under `rocprofv3 -- python3 bench.py --gpus N` the profiler's library has initialised the GPU before main():
        # such a process must not start launchers on this pool (profile one rank: --gpus 1, or the ranks of a launcher)
        print("bench.py: --gpus N without a launcher, but the GPU runtime is already loaded in this process (a profiler?): "
              "start the ranks with `python -m torch.distributed.run ... bench.py --gpus N`", file=sys.stderr)
        return 2
    import __graft_entry__ as entry
    ensure_built(entry)
    have = visible_gpu_count()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if have < args.gpus and "VMN_BENCH_BACKEND" not in env:
        print(f"bench.py: {have} GPU(s) visible for {args.gpus} ranks -- rehearsal: ranks share GPUs, collectives over gloo", file=sys.stderr)
        env["VMN_BENCH_BACKEND"] = "gloo"
    if have and have < args.gpus:                   # ranks that share a GPU share its memory: each caches a share of the usual bounds
        share = -(-args.gpus // have)
        env.setdefault("VMN_POOL_LIMIT_BYTES", str((64 << 30) // share))
        env.setdefault("VMN_FIXED_CACHE_BYTES", str((64 << 30) // share))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--elements", dest="n", type=int, default=1_000_000, help="elements per GPU")
    ap.add_argument("--cpu-sample-elements", dest="cpu_sample", type=int, default=0, help="elements of the CPU baseline sample (0 = auto)")
    ap.add_argument("--skip-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--mix-elements", dest="mix_n", type=int, default=1_000_000, help="ciphertexts of the mix+prove leg (0 = skip)")
    ap.add_argument("--ec-elements", dest="ec_n", type=int, default=1_000_000,
                    help="ciphertexts of the P-256 width-3 leg (BASELINE configs[4]; 0 = skip; single GPU only)")
    ap.add_argument("--ccpos-elements", dest="ccpos_n", type=int, default=1_000_000,
                    help="ciphertexts of the 3072-bit CCPoS leg (BASELINE configs[2]; 0 = skip; single GPU only)")
    ap.add_argument("--decrypt-elements", dest="dec_n", type=int, default=1_000_000,
                    help="ciphertexts of the verifiable-decryption leg (k = 3, threshold 2; 0 = skip; single GPU only)")
    ap.add_argument("--no-e2e", dest="no_e2e", action="store_true", help="skip the end-to-end pass of the mix + prove leg (profiling runs)")
    ap.add_argument("--no-shapes", dest="no_shapes", action="store_true",
                    help="skip modexp_by_shape (counter runs: the headline kernel must be the only k_modpow launch of the run)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --elements / --mix-elements PER GPU (BASELINE configs[1] on every GPU), strong = in TOTAL "
                         "(north_star's 1M-ciphertext shuffle split over the GPUs); the mix legs always report both")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` typed without a launcher starts its own ranks (as a child, before any GPU call)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # Build (only if a prebuilt library is missing) BEFORE anything touches the GPU: a process that has
    # initialised the GPU must not exec children on this pool, and under rocprofv3 it already has.
    import __graft_entry__ as entry
    ensure_built(entry)
    vmn = entry.load_package()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1
    if os.environ.get("VMN_BENCH_LAUNCH_ONLY"):       # launcher self-test (tests/test_bench_contract.py): no GPU is touched
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": world, "scaling": args.scaling, "steps": args.steps}))
        sys.exit(0)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box: VMN_BENCH_BACKEND=gloo lets several ranks share GPU 0 (RCCL needs one GPU per rank).
    backend = os.environ.get("VMN_BENCH_BACKEND", "nccl")
    n_devices = torch.cuda.device_count()
    if distributed and backend == "nccl" and n_devices < world:
        backend = "gloo"                          # RCCL needs one GPU per rank: this is a rehearsal on shared GPUs
    dev_index = local_rank % n_devices
    torch.cuda.set_device(dev_index)
    gloo_group = None

    class _stdout_to_stderr:
        """gloo announces its connections on the C-level stdout; the line of this program is the only thing allowed there."""
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

        def __exit__(self, *exc):
            os.dup2(self.saved, 1)
            os.close(self.saved)
            return False

    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        limit = datetime.timedelta(seconds=150)   # a rank that dies inside a leg must not hold the others for RCCL's default 10 min
        with _stdout_to_stderr():
            if backend == "nccl":
                try:
                    # a first all-reduce on the device proves the RCCL path before anything depends on it
                    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=limit)
                    probe = torch.ones(1, device=torch.device("cuda", dev_index))
                    dist.all_reduce(probe)
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        raise RuntimeError(f"RCCL all-reduce probe returned {probe.item()} for {world} ranks")
                    gloo_group = dist.new_group(backend="gloo", timeout=limit)      # safety net of parallel.Comm (never used unless RCCL raises)
                    dist.barrier(group=gloo_group)                   # connect now, while stdout is diverted
                except Exception as exc:      # pragma: no cover - needs a broken RCCL
                    nccl_failure = f"{type(exc).__name__}: {exc}"[:300]
                    print(f"bench.py: RCCL could not be brought up ({nccl_failure}); the run continues over gloo and says so", file=sys.stderr)
                    try:
                        if dist.is_initialized():
                            dist.destroy_process_group()
                    except Exception:
                        pass
                    backend = "gloo (RCCL failed: " + nccl_failure + ")"
                    dist.init_process_group(backend="gloo", timeout=limit)
                    dist.barrier()
            else:
                dist.init_process_group(backend=backend, timeout=limit)
                dist.barrier()
    red_device = "cuda" if backend == "nccl" else "cpu"
    comm = load_sub(entry, "parallel").Comm(dist if distributed else None, torch.device("cuda", dev_index) if (distributed and backend == "nccl") else None,
                                            fallback=gloo_group)
    sharded_leg_failed = []                       # (leg, error) of the first sharded leg that raised on THIS rank

    p, q, g = load_sub(entry, "stdgroups").modp_group(2048)      # RFC 3526 group 14
    nbytes = 256
    # weak: --elements per GPU; strong: --elements in total, this rank's contiguous share of them
    if distributed and args.scaling == "strong":
        base, rem = divmod(args.n, world)
        n = base + (1 if rank < rem else 0)
    else:
        n = args.n
    n_all = args.n if (distributed and args.scaling == "strong") else args.n * world
    ctx = vmn.Context(dev_index)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    grp = vmn.ModPGroup(ctx, p, q, g, nbytes=nbytes)

    xb, eb = make_inputs(n, 20260000 + rank, nbytes)
    X = grp.toElementArray(xb)
    E = grp.ringArray(eb)

    def step():
        return X.exp(E)

    for _ in range(args.warmup):
        step().free()
    ctx.timing_reset()
    ctx.timing_enable(True)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        if out is not None:
            out.free()
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.timing_enable(False)
    launches, kernel_ms = ctx.timing_get("modpow")

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total = n_all * args.steps
    value = total / elapsed
    avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
    achieved = MAC_2048_2047 * n / avg_kernel_s / 1e12
    # algorithmic HBM bytes of one launch: x, e (packed words), out, plus the per-lane window tables
    # written once and read once per window (they stay in L2/MALL mostly); SURVEY.md §8d: 768 B/element
    alg_bytes = 768 * n
    # HBM traffic from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, summarised in
    # profiles/r01_pmc_modpow_v11.json by tools/profile_pmc.sh); per element, scaled to this launch.
    traffic = None
    valu_busy = None
    pmc_instr = None
    n_launch = n
    pmc_file = "profiles/r04_pmc_kernels.json"
    pmc_note = None
    try:
        with open(os.path.join(ROOT, pmc_file)) as f:
            pmc_all = json.load(f)
        measured_on = pmc_all.get("family_fingerprints", {}).get("modp")
        if measured_on != source_fingerprint():
            pmc_note = (f"{pmc_file} was measured on another build of the headline kernel's sources (fingerprint {measured_on} != "
                        f"{source_fingerprint()}): its counters are not used; rerun tools/profile_pmc.sh")
            print("bench.py: " + pmc_note, file=sys.stderr)
        else:
            pmc = pmc_all["kernels"].get("k_modpow_phased<vmn::Cfg<74, 1>") or pmc_all["kernels"]["k_modpow<vmn::Cfg<74, 1>"]
            traffic = pmc["hbm_bytes_per_unit"] * n
            valu_busy = pmc["valu_busy_frac"]
            pmc_instr = pmc["valu_instr_per_unit"]
    except Exception as exc:
        pmc_note = f"{pmc_file} not usable ({type(exc).__name__}: {exc})"
        print("bench.py: " + pmc_note, file=sys.stderr)

    result = {
        "metric": "modexps/sec (batched variable-base modPow, 2048-bit ModPGroup, full-length exponents)",
        "value": value,
        "unit": "modexp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling if distributed else "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (28-bit radix), u64 accumulate",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: batched modPow, RFC 3526 group 14 (2048-bit safe prime), "
                               "random bases, random 2047-bit exponents, device-resident in/out",
                   "elements_per_gpu": n, "elements_total": n_all, "parallelism": f"shard{world}" if world > 1 else "single",
                   "world_size": world, "ranks_seen": (dist.get_world_size() if distributed else 1),
                   "collective_backend": (backend if distributed else None),
                   "gpus_visible": n_devices, "rehearsal_ranks_share_gpus": bool(distributed and n_devices < world),
                   "scaling": args.scaling if distributed else "weak",
                   "note": "element-wise op: contiguous shards, no data-path collective; the mix legs below exchange scalars only"},
        "roofline": {"bound": "valu-int", "kernel": "k_modpow_phased<74>", "achieved": achieved, "peak": PEAK_TMACS,
                     "unit": "TMAC/s (32x32->64-bit multiply-accumulate)", "frac": achieved / PEAK_TMACS,
                     "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
                     "traffic": traffic, "traffic_source": pmc_note or f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this build, bytes per element x n)",
                     "valu_busy_pmc": valu_busy,
                     # what the hardware sustains for v_mad_u64_u32 at this kernel's two waves per SIMD
                     # (profiles/valu_rate_r01.txt), and the kernel's issue rate from the PMC instruction count
                     "peak_measured": measured_valu_peak(), "peak_measured_source": "profiles/valu_rate_r01.csv: independent v_mad_u64_u32, two waves per SIMD",
                     "issued_Tlaneinstr_per_s": (pmc_instr * n_launch / avg_kernel_s / 1e12) if pmc_instr else None,
                     "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                             "achieved_GBs": alg_bytes / avg_kernel_s / 1e9, "peak_GBs": HBM_PEAK_GBS}},
    }

    # The legs below are extra objects of the line; a failure in one of them must not cost the headline value.
    def guarded(name, fn):
        # Sharded legs: every rank makes the same calls, so an error raised on one rank leaves the others waiting in the next
        # exchange until the group's timeout -- after which THEY raise in the same leg.  Every rank therefore sees a failure
        # in that leg (its own error or the timeout), and all of them skip the remaining sharded legs instead of blocking
        # once more (seen in a three-ranks-on-one-GPU rehearsal: one rank ran out of memory for a table).
        if distributed and sharded_leg_failed:
            result[name] = {"error": f"skipped: the sharded leg {sharded_leg_failed[0][0]} failed on a rank ({sharded_leg_failed[0][1]})"}
            return
        try:
            fn()
        except Exception as exc:                       # pragma: no cover - reported in the line
            import traceback
            traceback.print_exc(file=sys.stderr)
            result[name] = {"error": f"{type(exc).__name__}: {exc}"}
            if distributed:
                sharded_leg_failed.append((name, f"{type(exc).__name__}: {exc}"[:200]))

    def leg_mix_prove():
        X.free()
        E.free()
        ctx.timing_reset()
        if distributed:
            # One proof sharded over the ranks, in both scaling forms: weak = mix_n ciphertexts PER GPU, strong = mix_n in
            # TOTAL (north_star: "near-linear 1/2/4/8-GPU scaling on a 1M-ciphertext shuffle"); --scaling picks the one the
            # object's top level reports.  BASELINE configs[3] (4 194 304 ciphertexts on the node) is added when it is
            # neither of the two.
            sizes = {"weak": args.mix_n * world, "strong": args.mix_n}
            first = args.scaling
            mp = mix_prove_sharded(entry, vmn, ctx, grp, sizes[first], 777, barrier, comm)
            mp["scaling"] = first
            other = "strong" if first == "weak" else "weak"
            extra = {other: sizes[other]}
            if 4_194_304 not in sizes.values() and args.mix_n >= 100_000:
                extra["configs3_4Mi"] = 4_194_304
            for name, size in extra.items():
                # no try / except here: an exception on ONE rank must reach guarded(), which records it in sharded_leg_failed so
                # that every rank skips the remaining sharded legs the same way (a rank that swallowed the error would pair
                # its next all-gathers with the peers' pending exchanges of the failed leg)
                sub = mix_prove_sharded(entry, vmn, ctx, grp, size, 778, barrier, comm)
                sub.pop("roofline_rank0", None)
                mp[name] = sub
        else:
            mp = mix_prove(entry, vmn, ctx, fresh_group(), args.mix_n, 777 + rank, barrier, steps=2)
        if not distributed and not args.no_e2e:
            try:
                runs = [mix_prove_e2e(entry, vmn, ctx, fresh_group(), args.mix_n, 4321 + k, barrier) for k in range(2)]
                e2e = dict(min(runs, key=lambda r: abs(r["total_ms"] - mean_of(runs, "total_ms"))))
                for key in ("prove_ms", "verify_ms", "total_ms", "setup_ms", "total_ms_one_shot"):
                    e2e[key] = mean_of(runs, key)
                e2e["passes_total_ms"] = [round(r["total_ms"], 1) for r in runs]
                e2e["statistic"] = "mean of 2 passes, each from a cold group and with its own generators"
                e2e["ciphertexts_per_s"] = args.mix_n / (e2e["total_ms"] / 1e3)
                e2e["ciphertexts_per_s_one_shot"] = args.mix_n / (e2e["total_ms_one_shot"] / 1e3)
                mp["end_to_end"] = e2e
            except Exception as exc:                   # pragma: no cover - reported in the line
                import traceback
                traceback.print_exc(file=sys.stderr)
                mp["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        mp["workload"] = ("re-encrypt + PoS (Terelius-Wikstrom) prove + verify, ModPGroup 2048-bit, width 1, "
                          "n_e = n_v = 256, n_r = 100" + ("; ONE proof sharded by position over the ranks (all-gather of partial "
                                                           "products / scan carries only)" if distributed else ""))
        result["mix_prove"] = mp

    def leg_ccpos_sharded():
        ctx.timing_reset()
        p3, q3, g3 = load_sub(entry, "stdgroups").modp_group(3072)
        grp3 = vmn.ModPGroup(ctx, p3, q3, g3, nbytes=384)
        size = args.ccpos_n * world if args.scaling == "weak" else args.ccpos_n
        r = mix_ccpos_sharded(entry, vmn, ctx, grp3, "BASELINE.json configs[2] sharded over the ranks: ModPGroup 3072-bit, width 1; offline = "
                              "permutation commitment + PoSC prove+verify, online = re-encrypt + CCPoS prove+verify", size, 4242, barrier, comm, 1, True)
        r["scaling"] = args.scaling
        result["mix_ccpos_3072"] = r

    def leg_ec_sharded():
        ctx.timing_reset()
        grpc = vmn.ECqPGroup(ctx, "P-256")
        # BASELINE configs[4] is 1M ciphertexts of width 3 on the 8 GPUs of the node (= the strong form); weak = 1M per GPU
        sizes = {"weak": args.ec_n * world, "strong": args.ec_n}
        label = "BASELINE.json configs[4]: ECqPGroup P-256, width 3, sharded over the ranks; offline = permutation commitment, online = re-encrypt + CCPoS prove+verify"
        r = mix_ccpos_sharded(entry, vmn, ctx, grpc, label, sizes[args.scaling], 555, barrier, comm, 3, False)
        r["scaling"] = args.scaling
        other = "strong" if args.scaling == "weak" else "weak"
        sub = mix_ccpos_sharded(entry, vmn, ctx, grpc, label, sizes[other], 556, barrier, comm, 3, False)     # (errors reach guarded(), see leg_mix_prove)
        sub.pop("roofline_rank0", None)
        r[other] = sub
        result["mix_ec_p256"] = r

    def leg_ccpos():
        ctx.timing_reset()
        first = mix_ccpos(entry, vmn, ctx, args.ccpos_n, 4241, barrier)       # untimed: see mean_pass (first_pass_*)
        runs = [mix_ccpos(entry, vmn, ctx, args.ccpos_n, 4242 + k, barrier) for k in range(2)]
        instr = mix_ccpos(entry, vmn, ctx, args.ccpos_n, 4242, barrier, instrumented=True)
        result["mix_ccpos_3072"] = mean_pass(runs, args.ccpos_n, ("offline_ms", "reencrypt_ms", "ccpos_prove_ms", "ccpos_verify_ms", "online_ms", "total_ms"),
                                             "online_ms", "ciphertexts_per_s_online", instr, first=first)

    def leg_ec():
        ctx.timing_reset()
        first = mix_ec(entry, vmn, ctx, args.ec_n, 554, barrier)              # untimed: see mean_pass (first_pass_*)
        runs = [mix_ec(entry, vmn, ctx, args.ec_n, 555 + k, barrier) for k in range(2)]
        instr = mix_ec(entry, vmn, ctx, args.ec_n, 555, barrier, instrumented=True)
        result["mix_ec_p256"] = mean_pass(runs, args.ec_n, ("offline_ms", "reencrypt_ms", "ccpos_prove_ms", "ccpos_verify_ms", "online_ms", "total_ms"),
                                          "online_ms", "ciphertexts_per_s_online", instr, first=first)

    def leg_small():
        # BASELINE.json configs[0]'s size (the reference's demo: 10^4 ciphertexts, 2048 bits, width 1): below ~4 x 10^4
        # elements every launch runs in a wide geometry; the proof is bound by per-lane chain latency
        ctx.timing_reset()
        sm = mix_prove(entry, vmn, ctx, fresh_group(), 10000, 777, barrier, steps=3, fs_line=False)
        sm["workload"] = "re-encrypt + PoS prove + verify at the reference's demo size (BASELINE.json configs[0]: 10^4 ciphertexts, 2048 bits, width 1)"
        result["mix_prove_n10000"] = sm

    def leg_fit():
        # e(N), v(N), p(N) as the reference's benchmark reports them: the two sizes already measured + two in between
        pts = [result["mix_prove_n10000"]]
        for n_mid in (100_000, 300_000):
            if n_mid < args.mix_n:
                ctx.timing_reset()
                pts.append(mix_prove(entry, vmn, ctx, fresh_group(), n_mid, 777, barrier, steps=2, fs_line=False))
        pts.append(result["mix_prove"])
        result["operation_length"] = operation_length_fit(pts)

    def leg_fit_p256():
        # the same analysis on the reference's OWN benchmark group (demo/mixnet/benchmarks/bench_config:33-34: P-256;
        # operation_length:30-35: 200 ... 1000 x size ciphertexts, width 1), re-encrypt + PoS prove + verify
        pts = []
        for n_pt in (1000, 10_000, 100_000, 1_000_000):
            if n_pt <= max(args.ec_n, 10_000):
                ctx.timing_reset()
                grpc = vmn.ECqPGroup(ctx, "P-256")
                pts.append(mix_prove(entry, vmn, ctx, grpc, n_pt, 888, barrier, steps=2, fs_line=False))
                grpc.close()
        fit = operation_length_fit(pts)
        fit["group"] = "ECqPGroup P-256, width 1 (the reference's benchmark group, demo/mixnet/benchmarks/bench_config:33)"
        fit["ciphertexts_per_s"] = [p_["ciphertexts_per_s"] for p_ in pts]
        fit["setup_ms"] = [p_["setup_ms"] for p_ in pts]
        fit["frac_canonical"] = [p_["roofline"]["frac_canonical"] for p_ in pts]
        fit["kernel_launches"] = [p_["kernel_launches"] for p_ in pts]
        fit["warmup"] = "every point: one untimed pass first (mix_prove: first_pass_total_ms), then the mean of two"
        fit["first_pass_total_ms"] = [p_["first_pass_total_ms"] for p_ in pts]
        result["operation_length_p256"] = fit

    def leg_shapes():
        ctx.timing_reset()
        cores_ = min(len(os.sched_getaffinity(0)), 16)
        result["modexp_by_shape"] = modexp_by_shape(entry, vmn, ctx, p, q, g, args.n, barrier, cores_, with_cpu=(rank == 0 and not args.no_cpu))

    def leg_decrypt():
        ctx.timing_reset()
        result["decrypt_2048"] = decrypt_leg(entry, vmn, ctx, grp, args.dec_n, 999, barrier)

    def fresh_group():
        """A leg that reports a one-shot figure starts from a group without cached tables."""
        return vmn.ModPGroup(ctx, p, q, g, nbytes=nbytes)

    if args.n >= 10000 and not distributed and not args.no_shapes:
        X.free()
        E.free()
        guarded("modexp_by_shape", leg_shapes)
    if args.mix_n > 0:
        guarded("mix_prove", leg_mix_prove)
    if args.dec_n > 0 and not distributed:
        guarded("decrypt_2048", leg_decrypt)
    if args.mix_n >= 10000 and not distributed:
        guarded("mix_prove_n10000", leg_small)
        if "error" not in result.get("mix_prove", {"error": 1}) and "error" not in result.get("mix_prove_n10000", {"error": 1}) and args.mix_n > 10000:
            guarded("operation_length", leg_fit)
    if args.ccpos_n > 0:
        guarded("mix_ccpos_3072", leg_ccpos_sharded if distributed else leg_ccpos)
    if args.ec_n > 0:
        guarded("mix_ec_p256", leg_ec_sharded if distributed else leg_ec)
        if not distributed:
            guarded("operation_length_p256", leg_fit_p256)

    if rank == 0 and not args.no_cpu:
        from oracle.cbind import Oracle
        orc = Oracle(p, q, nbytes)
        # host cores this job may use: the affinity mask, capped at the 16-core share of a one-GPU box
        # (a launcher pins OMP_NUM_THREADS to 1 for its ranks: the oracle's thread count is set explicitly, not inherited)
        cores = min(len(os.sched_getaffinity(0)), 16 * max(1, world)) if distributed else min(orc.threads, len(os.sched_getaffinity(0)), 16)
        orc.set_threads(cores)
        sample = args.cpu_sample or min(n, 6000 * cores)          # ~15 s of mpz_powm
        t1 = time.perf_counter()
        want = orc.exp_array_bytes(xb[: sample * nbytes], eb[: sample * nbytes], sample, nbytes)
        cpu_s = time.perf_counter() - t1
        got = out.copyOfRange(0, sample).toBytes() if sample < n else out.toBytes()
        result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "modexp/s", "cores": cores, "kind": "port",
                                  "sample": f"first {sample} of the {n} elements of rank 0 (GMP mpz_powm, OpenMP static chunks)",
                                  "bit_exact_vs_gpu": got == want}
        if got != want:
            result["parity_error"] = "GPU output differs from the GMP oracle on the sample"
            result["value"] = None                         # a wrong result has no throughput
        if "mix_prove" in result and "error" not in result["mix_prove"] and not distributed:
            try:
                result["mix_prove"]["cpu_baseline"] = cpu_mix_prove(p, q, g, 30000, cores)
            except Exception as exc:                   # pragma: no cover
                result["mix_prove"]["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        if "decrypt_2048" in result and "error" not in result["decrypt_2048"]:
            try:
                result["decrypt_2048"]["cpu_baseline"] = cpu_decrypt(p, q, g, 20000, cores)
            except Exception as exc:                   # pragma: no cover
                result["decrypt_2048"]["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()
    if "parity_error" in result:
        print("bench.py: " + result["parity_error"], file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()

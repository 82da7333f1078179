#!/usr/bin/env python3
"""Generate the hand-scheduled gfx950 instruction stream of one Montgomery "row" (see
csrc/mont28.h) as an inline-asm device function, for a given limb count S.

hipcc's register allocator does not find the in-place, destination-shifted form of the row
(P[j-1] = m*N[j] + P[j]); it doubles the live ranges and spills.  The row is therefore emitted as
one asm statement whose operands the compiler only has to *place*: S column pairs ("+v"),
S multiplicand limbs ("v"), S modulus limbs ("s"), the multiplier limb and -1/N mod 2^28.

Instruction stream per row (exactly 2*S v_mad_u64_u32 + 5 other VALU, no memory, no carries):
    P[0]   = a[0]*b + P[0]
    m      = (lo(P[0]) * n0inv) & (2^28-1)
    c      = m*N[0] + P[0]                  ; low 28 bits zero
    P[1]   = a[1]*b + P[1]
    c      = c >> 28
    P[0]   = m*N[1] + P[1] ; P[0] += c
    j>=2:  P[j] = a[j]*b + P[j]  (top column: + 0) ; P[j-1] = m*N[j] + P[j]
The a*b and m*N multiply-adds of neighbouring columns are interleaved so that no instruction
depends on its immediate predecessor.

usage: gen_mont_asm.py S [S ...] > csrc/gen/mont_rows.inc
"""
import sys


P0REG = 2   # column 0 is pinned to v[2:3]: inline asm cannot name the low half of a 64-bit operand
SQR_BLK = 1  # rows per block of the squaring schedule
FILL_FROM_BLOCK = int(__import__("os").environ.get("VMN_ROW_FILL", "0"))
LEAD = int(__import__("os").environ.get("VMN_ROW_LEAD", "3"))   # products issued ahead of the reduction (see _row)


def _row(S: int, first: bool, j0: int = 0, blk: int = 0):
    """Instruction list and operand lists of one row.

    blk == 0: general multiplication row, multiplier b for every column.
    blk  > 0: squaring row of the block of rows [j0, j0+blk): the row's own limb a_i multiplies the
              columns of its block with b (= a_i) and the columns of later blocks with b2 (= 2 a_i);
              columns before the block get no product (their cross terms were added, doubled, by the
              earlier rows).  Every cross product a_i a_j is thus formed once (doubled) when i and j are in
              different blocks and twice (plain) inside a block; only the multiplications differ from the
              general row, the reduction half (m * N) is identical.
    """
    sqr = blk > 0
    P = lambda j: f"%{j}"                 # 0..S-1      u64 "+v"
    M = f"%{S}"                           # m           u32 "=&v"
    C = f"%{S + 1}"                       # c           u64 "=&v"
    A = lambda j: f"%{S + 2 + j}"         # a[j]        u32 "v"
    B = f"%{2 * S + 2}"                   # b limb      u32 "v"
    N = lambda j: f"%{2 * S + 3 + j}"     # N[j]        u32 "s"
    NI = f"%{3 * S + 3}"                  # n0inv       u32 "s"
    B2 = f"%{3 * S + 4}"                  # 2*b         u32 "v"  (squaring rows only)
    MASK = "0xfffffff"                    # 2^28-1 as a 32-bit literal (saves an SGPR)

    L = []
    emitted = set()

    def ab(j):   # pass 1 for column j
        if j in emitted or j >= S:
            return
        emitted.add(j)
        if sqr and j < j0:
            return                                    # no product for this column in this block
        mult = B2 if (sqr and j >= j0 + blk) else B
        if first or j == S - 1:
            L.append(f"v_mad_u64_u32 {P(j)}, vcc, {A(j)}, {mult}, 0")
        else:
            L.append(f"v_mad_u64_u32 {P(j)}, vcc, {A(j)}, {mult}, {P(j)}")

    # head: the chain  P[0] -> m -> c -> P[0]  is five dependent instructions; LEAD independent products of the
    # same row are spread between its links so that a wave has something to issue while a link is in flight
    # (LEAD = 3 was the first schedule; see DESIGN.md for the measurement)
    ab(0)
    k = 1
    def lead(cnt):
        nonlocal k
        for _ in range(cnt):
            if sqr and FILL_FROM_BLOCK and k < j0:
                k = j0                               # squaring rows: columns before the block get no product; fill
            ab(k)                                    # the chain's gaps with the block's own products instead
            k += 1
    per = max(1, (LEAD - 1) // 4)
    lead(per)
    L.append(f"v_mul_lo_u32 {M}, v{P0REG}, {NI}")   # low dword of column 0 (operand 0 is pinned to v[P0REG:P0REG+1])
    lead(per)
    L.append(f"v_and_b32 {M}, {MASK}, {M}")
    lead(per)
    L.append(f"v_mad_u64_u32 {C}, vcc, {M}, {N(0)}, {P(0)}")
    ab(1)                                            # column 1 must have its product before it moves to column 0
    L.append(f"v_mad_u64_u32 {P(0)}, vcc, {M}, {N(1)}, {P(1)}")
    L.append(f"v_lshrrev_b64 {C}, 28, {C}")
    lead(per)
    L.append(f"v_lshl_add_u64 {P(0)}, {P(0)}, 0, {C}")
    # steady state: a*b for column j+LEAD is issued LEAD slots before m*N consumes column j
    for j in range(2, S):
        ab(j + LEAD)
        ab(j)                                        # (no-op when already emitted)
        L.append(f"v_mad_u64_u32 {P(j - 1)}, vcc, {M}, {N(j)}, {P(j)}")
    return L


def _emit(name: str, S: int, lines, first: bool, sqr: bool, tparams: str) -> str:
    out = []
    extra = ", u32 b2" if sqr else ""
    out.append(f"template <> __device__ __forceinline__ void {name}<{tparams}>(u64 (&P)[{S}], const u32 (&a)[{S}], u32 b{extra},\n"
               f"        const u32 (&n)[{S}], u32 n0inv) {{")
    out.append("    u32 m; u64 c;")
    out.append("    asm volatile(")
    for l in lines:
        out.append(f'        "{l}\\n\\t"')
    # in a "first" row of a squaring block only the columns that get a product are written before being read
    cons = lambda j: ("=&" if first else "+") + ("{v[%d:%d]}" % (P0REG, P0REG + 1) if j == 0 else "v")
    outs = ", ".join([f'"{cons(j)}"(P[{j}])' for j in range(S)] + ['"=&v"(m)', '"=&v"(c)'])
    ins = ", ".join([f'"v"(a[{j}])' for j in range(S)] + ['"v"(b)'] + [f'"s"(n[{j}])' for j in range(S)]
                    + ['"s"(n0inv)'] + (['"v"(b2)'] if sqr else []))
    out.append(f"        : {outs}")
    out.append(f"        : {ins}")
    out.append('        : "vcc");')
    out.append("}")
    out.append("")
    return "\n".join(out)


def gen(S: int) -> str:
    parts = []
    for first in (True, False):
        name = f"mont_row_asm_{'first' if first else 'next'}"
        parts.append(_emit(name, S, _row(S, first), first, False, f"{S}"))
    # squaring rows: the very first row (all columns fresh), then one variant per block
    parts.append(_emit("mont_sqr_row_asm_first", S, _row(S, True, 0, SQR_BLK), True, True, f"{S}"))
    for j0 in range(0, S, SQR_BLK):
        parts.append(_emit("mont_sqr_row_asm", S, _row(S, False, j0, SQR_BLK), False, True, f"{S}, {j0}"))
    return "\n".join(parts)


# ------------------------------------------------------------------------------------------------
# Two lanes per element (moduli > 2072 bits): lane h of a pair (h = lane & 1) holds limbs / columns
# [h*L, (h+1)*L) of the S = 2L limbs.  One row, executed by both lanes in lockstep:
#     pass 1:  P[j] = a[j]*b + P[j]                        (local columns; b is the same for both lanes)
#     m        = (P[0] * n0inv) & mask  taken from the EVEN lane (DPP quad_perm [0,0,2,2])
#     c        = m*N[0] + P[0]                              even lane: low 28 bits zero, c>>28 carries into new column 0
#                                                           odd lane : this is the value of global column L, which
#                                                                      becomes the even lane's new column L-1
#     pass 2:  P[j-1] = m*N[j] + P[j]   (j = 1..L-1)
#     P[0]    += (c >> 28) & evenmask                       (only the even lane has a carry)
#     P[L-1]   = dpp_from_odd(c) & evenmask                 even lane receives the odd lane's c, odd lane gets the
#                                                           fresh zero of the top column
# Column 0, column L-1 and c are pinned to fixed register pairs because DPP moves 32-bit halves.
# A VALU write followed by a DPP read of the same VGPR needs 2 wait states: every DPP below is preceded by
# "s_nop 1" (hardware does not interlock this hazard).
# ------------------------------------------------------------------------------------------------
PAIR_P0, PAIR_PT, PAIR_C = 2, 4, 6      # v[2:3] column 0, v[4:5] column L-1, v[6:7] c


def _row_pair(L: int, first: bool, lanes: int = 2, j0: int = 0, blk: int = 0):
    """blk > 0: SQUARING row of the block of local rows [j0, j0 + blk) -- see the note below the function.
    lanes = 2: the pair row described above.  lanes = 4 (moduli > 3080 bits): lane h = lane & 3 holds columns
    [h*L, (h+1)*L); m is taken from lane 0 of the quad (quad_perm [0,0,0,0]); every lane hands its c to the lane
    below (quad_perm [1,2,3,3]; the top lane receives the fresh zero); only lane 0 carries c >> 28 into its new
    column 0.  Two masks: LOW = all ones on lane 0 of the element, NOTTOP = all ones on every lane but the last."""
    P = lambda j: f"%{j}"                 # 0..L-1      u64
    M = f"%{L}"                           # m
    A = lambda j: f"%{L + 2 + j}"         # a[j]
    B = f"%{2 * L + 2}"                   # b
    N = lambda j: f"%{2 * L + 3 + j}"     # N[j] (VGPR: differs between the lanes of a pair)
    NI = f"%{3 * L + 3}"                  # n0inv (SGPR)
    EM = f"%{3 * L + 4}"                  # LOW mask: 0xffffffff on lane 0 of the element (pair: the even lane)
    NT = f"%{3 * L + 5}"                  # NOTTOP mask: 0xffffffff on all lanes but the last (pair: the even lane)
    B2 = f"%{3 * L + 6}"                  # 2*b (squaring rows only)
    sqr = blk > 0
    # lanes = 8 (the widest geometry, small arrays of 2048-bit elements): the element is half a DPP row of 16 lanes.  m comes
    # from lane 0 in two steps (quad broadcast, then the upper quad copies the lower one: row_shr:4 into banks 1 and 3), c from
    # the lane above by row_shl:1 (the last lane of an element reads its neighbour's lane 0: masked by NOTTOP as before).
    # lanes = 16 (moduli above 8192 bits): a whole DPP row; one more broadcast step (row_shr:8 into banks 2 and 3).
    bcast = "[0,0,2,2]" if lanes == 2 else "[0,0,0,0]"
    from_above = "quad_perm:[1,1,3,3]" if lanes == 2 else "quad_perm:[1,2,3,3]" if lanes == 4 else "row_shl:1"
    C = f"%{L + 1}"
    MASK = "0xfffffff"
    out = []
    emitted = set()

    def ab(j):
        if j in emitted or j >= L:
            return
        emitted.add(j)
        if sqr and j < j0:
            return                                    # no product for this local column in this block of rows
        mult = B2 if (sqr and j >= j0 + blk) else B
        out.append(f"v_mad_u64_u32 {P(j)}, vcc, {A(j)}, {mult}, " + ("0" if first else P(j)))

    ab(0)
    ab(1)
    out.append(f"v_mul_lo_u32 {M}, v{PAIR_P0}, {NI}")
    ab(2)
    out.append(f"v_and_b32 {M}, {MASK}, {M}")
    ab(3)
    ab(4)
    out.append("s_nop 1")
    out.append(f"v_mov_b32_dpp {M}, {M} quad_perm:{bcast} row_mask:0xf bank_mask:0xf")
    if lanes == 8:
        out.append("s_nop 1")
        out.append(f"v_mov_b32_dpp {M}, {M} row_shr:4 row_mask:0xf bank_mask:0xa")
    if lanes == 16:
        out.append("s_nop 1")
        out.append(f"v_mov_b32_dpp {M}, {M} row_shr:4 row_mask:0xf bank_mask:0x2")
        out.append("s_nop 1")
        out.append(f"v_mov_b32_dpp {M}, {M} row_shr:8 row_mask:0xf bank_mask:0xc")
    ab(5)
    ab(6)
    out.append(f"v_mad_u64_u32 {C}, vcc, {M}, {N(0)}, {P(0)}")
    for j in range(1, L):
        ab(j + 6)
        out.append(f"v_mad_u64_u32 {P(j - 1)}, vcc, {M}, {N(j)}, {P(j)}")
        if j == 2:
            # hand the odd lane's c to the even lane's column L-1 ... after pass 2 has consumed the old column L-1
            pass
    # column L-1 was consumed by the last pass-2 multiply-add; now refill it from c
    out.append("s_nop 1")
    out.append(f"v_mov_b32_dpp v{PAIR_PT}, v{PAIR_C} {from_above} row_mask:0xf bank_mask:0xf")
    out.append(f"v_mov_b32_dpp v{PAIR_PT + 1}, v{PAIR_C + 1} {from_above} row_mask:0xf bank_mask:0xf")
    out.append(f"v_lshrrev_b64 {C}, 28, {C}")
    out.append(f"v_and_b32 v{PAIR_PT}, {NT}, v{PAIR_PT}")
    out.append(f"v_and_b32 v{PAIR_PT + 1}, {NT}, v{PAIR_PT + 1}")
    out.append(f"v_and_b32 v{PAIR_C}, {EM}, v{PAIR_C}")
    out.append(f"v_and_b32 v{PAIR_C + 1}, {EM}, v{PAIR_C + 1}")
    out.append(f"v_lshl_add_u64 {P(0)}, {P(0)}, 0, {C}")
    return out


# Squaring with several lanes per element.  Lane h holds the limbs a_h = a[hL .. (h+1)L) (contiguous shares), so
#     a^2 = sum_h a_h^2 B^(2hL)  +  2 sum_{u<h} a_u a_h B^((u+h)L),          B = 2^28.
# The rows are the general ones (row i: every lane multiplies its own limbs by b = a_i from LDS, local columns, the
# cross-lane shift of the reduction), but the SAME local columns are skipped on every lane, which is what saves time
# in a wave: with i' = i mod L the row's local index and J0 the first row of its block of SQR_BLK rows, every lane
#     skips its local columns j < J0,  multiplies the columns J0 <= j < J0 + SQR_BLK by b,  the later ones by 2b.
# Why this is a^2: take row i = uL + i' (the multiplier limb belongs to share u) on lane h.
#   h == u  the symmetric rule of the one-lane squaring: a_j a_i' is formed once, doubled, when j is in a later block;
#           twice, plain, inside the block (rows i' and j); the square a_i'^2 once.
#   h != u  the lane forms the cross products a_h[j] * a_u[i'], column (h + u)L + j + i'.  The same pair is also within
#           reach of lane u in row hL + j (its limb a_u[i'] times b = a_h[j]).  Splitting the pairs by blocks of the local
#           indices -- block(j) > block(i'): here, doubled; block(j) < block(i'): there, doubled; same block: both, plain
#           -- forms every cross product exactly twice, and is the very same rule.
# Per lane a squaring is LPE * (L^2 / 2 + L * SQR_BLK / 2) products instead of LPE * L^2.  Column bound: a column lives
# S rows and receives one product per row -- none, plain (< 2^56) or doubled (< 2^57), about half of each -- plus one
# m * N < 2^56: S * 2^57 + LPE * SQR_BLK * 2^56 < 2^64 for S = 110; S = 148 is relieved half way (mont28.h).
def gen_pair_sqr(S: int, lanes: int) -> str:
    L = S // lanes
    parts = []
    variants = [(True, 0)] + [(False, j0) for j0 in range(0, L, SQR_BLK)]
    for first, j0 in variants:
        lines = _row_pair(L, first, lanes, j0, SQR_BLK)
        o = []
        if first:
            o.append(f"template <> __device__ __forceinline__ void mont_lanes_sqr_row_asm_first<{L}, {lanes}>(u64 (&P)[{L}], const u32 (&a)[{L}], u32 b, u32 b2,\n"
                     f"        const u32 (&n)[{L}], u32 n0inv, u32 lowmask, u32 nottopmask) {{")
        else:
            o.append(f"template <> __device__ __forceinline__ void mont_lanes_sqr_row_asm<{L}, {lanes}, {j0}>(u64 (&P)[{L}], const u32 (&a)[{L}], u32 b, u32 b2,\n"
                     f"        const u32 (&n)[{L}], u32 n0inv, u32 lowmask, u32 nottopmask) {{")
        o.append("    u32 m; u64 c;")
        o.append("    asm volatile(")
        for l in lines:
            o.append(f'        "{l}\\n\\t"')

        def cons(j):
            pre = "=&" if first else "+"
            if j == 0:
                return pre + "{v[%d:%d]}" % (PAIR_P0, PAIR_P0 + 1)
            if j == L - 1:
                return pre + "{v[%d:%d]}" % (PAIR_PT, PAIR_PT + 1)
            return pre + "v"
        outs = ", ".join([f'"{cons(j)}"(P[{j}])' for j in range(L)] + ['"=&v"(m)', '"=&{v[%d:%d]}"(c)' % (PAIR_C, PAIR_C + 1)])
        ins = ", ".join([f'"v"(a[{j}])' for j in range(L)] + ['"v"(b)'] + [f'"v"(n[{j}])' for j in range(L)]
                        + ['"s"(n0inv)', '"v"(lowmask)', '"v"(nottopmask)', '"v"(b2)'])
        o.append(f"        : {outs}")
        o.append(f"        : {ins}")
        o.append('        : "vcc");')
        o.append("}")
        o.append("")
        parts.append("\n".join(o))
    return "\n".join(parts)


def gen_pair(S: int, lanes: int = 2) -> str:
    L = S // lanes
    assert lanes * L == S and L >= 8
    parts = []
    for first in (True, False):
        name = f"mont_lanes_row_asm_{'first' if first else 'next'}"
        lines = _row_pair(L, first, lanes)
        o = []
        o.append(f"template <> __device__ __forceinline__ void {name}<{L}, {lanes}>(u64 (&P)[{L}], const u32 (&a)[{L}], u32 b,\n"
                 f"        const u32 (&n)[{L}], u32 n0inv, u32 lowmask, u32 nottopmask) {{")
        o.append("    u32 m; u64 c;")
        o.append("    asm volatile(")
        for l in lines:
            o.append(f'        "{l}\\n\\t"')

        def cons(j):
            pre = "=&" if first else "+"
            if j == 0:
                return pre + "{v[%d:%d]}" % (PAIR_P0, PAIR_P0 + 1)
            if j == L - 1:
                return pre + "{v[%d:%d]}" % (PAIR_PT, PAIR_PT + 1)
            return pre + "v"
        outs = ", ".join([f'"{cons(j)}"(P[{j}])' for j in range(L)] + ['"=&v"(m)', '"=&{v[%d:%d]}"(c)' % (PAIR_C, PAIR_C + 1)])
        ins = ", ".join([f'"v"(a[{j}])' for j in range(L)] + ['"v"(b)'] + [f'"v"(n[{j}])' for j in range(L)]
                        + ['"s"(n0inv)', '"v"(lowmask)', '"v"(nottopmask)'])
        o.append(f"        : {outs}")
        o.append(f"        : {ins}")
        o.append('        : "vcc");')
        o.append("}")
        o.append("")
        parts.append("\n".join(o))
    return "\n".join(parts)


def render(sizes, pair_sizes=(), quad_sizes=(), octo_sizes=(), hexa_sizes=()) -> str:
    parts = ["// GENERATED by tools/gen_mont_asm.py " + " ".join(map(str, sizes)) + " -- do not edit.",
             "// One Montgomery row as a single asm statement: 2*S v_mad_u64_u32 + 5 VALU, see the generator."]
    for S in sizes:
        parts.append(gen(S))
    for S in pair_sizes:
        parts.append(f"// two lanes per element, S = {S} limbs ({S // 2} per lane)")
        parts.append(gen_pair(S))
        parts.append(gen_pair_sqr(S, 2))
    for S in quad_sizes:
        parts.append(f"// four lanes per element, S = {S} limbs ({S // 4} per lane)")
        parts.append(gen_pair(S, 4))
        parts.append(gen_pair_sqr(S, 4))
    for S in octo_sizes:
        parts.append(f"// eight lanes per element, S = {S} columns ({S // 8} per lane)")
        parts.append(gen_pair(S, 8))
        parts.append(gen_pair_sqr(S, 8))
    for S in hexa_sizes:
        parts.append(f"// sixteen lanes per element, S = {S} columns ({S // 16} per lane)")
        parts.append(gen_pair(S, 16))
        parts.append(gen_pair_sqr(S, 16))
    return "\n".join(parts) + "\n"


def main():
    args = sys.argv[1:]
    pair = [int(x[1:]) for x in args if x.startswith("p")]
    quad = [int(x[1:]) for x in args if x.startswith("q")]
    octo = [int(x[1:]) for x in args if x.startswith("o")]
    hexa = [int(x[1:]) for x in args if x.startswith("h")]
    sizes = [int(x) for x in args if x[0].isdigit()] or [74]
    sys.stdout.write(render(sizes, pair, quad, octo, hexa))


if __name__ == "__main__":
    main()

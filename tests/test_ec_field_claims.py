"""Compile-time knowledge `csrc/ec_kernels.h` holds about the field primes, checked against the primes themselves (CPU only).

The curve kernels skip the zero limbs of P-256 / P-384 in every reduction row (`FieldPrime<S>::limb`), take the Montgomery
quotient digit as the low limb itself (-p^-1 = 1 mod 2^28), and reject "difference = 0 mod p ?" with a four-instruction
filter on the limbs 1 and 2 of the difference (`f_maybe_zero`).  Each of these is a statement about an integer; a wrong
table would still pass the GPU parity tests only by luck of the inputs, so the statements are tested here directly."""
import pathlib
import random
import re

HEADER = pathlib.Path(__file__).resolve().parents[1] / "verificatum-vmn_amd" / "csrc" / "ec_kernels.h"
MASK = (1 << 28) - 1
P256 = 2**256 - 2**224 + 2**192 + 2**96 - 1
P384 = 2**384 - 2**128 - 2**96 + 2**32 - 1


def limbs_of(s: int) -> list[int]:
    m = re.search(r"struct FieldPrime<%d> \{.*?limb\[%d\] = \{(.*?)\};" % (s, s), HEADER.read_text(), re.S)
    assert m, s
    return [int(t.strip().rstrip("u"), 16) for t in m.group(1).split(",")]


def radix28(v: int, s: int) -> list[int]:
    return [(v >> (28 * j)) & MASK for j in range(s)]


def test_the_limb_tables_are_the_primes():
    assert limbs_of(10) == radix28(P256, 10)
    assert limbs_of(15) == radix28(P384, 15)
    for p in (P256, P384):
        assert (-pow(p, -1, 1 << 28)) % (1 << 28) == 1          # the quotient digit of a reduction row is the low limb itself
        assert p & MASK == MASK                                  # the carry fold of mont_row


def test_zero_filter_never_rejects_a_multiple_of_p256():
    """f_maybe_zero: a value k p (k < 2^28) with normalised limbs has limbs 1 and 2 both all ones (k > 0) or both zero (k = 0);
    the kernels only apply it to differences below 2^9 p."""
    rnd = random.Random(5)
    ks = list(range(0, 600)) + [MASK, MASK - 1, 1 << 27] + [rnd.randrange(1 << 28) for _ in range(2000)]
    for k in ks:
        l = radix28(k * P256, 11)
        passes = (l[1] & l[2]) == MASK or (l[1] | l[2]) == 0
        assert passes, k
    # and it does reject: a random residue passes with probability 2^-55
    assert sum(1 for _ in range(20000)
               if (lambda l: (l[1] & l[2]) == MASK or (l[1] | l[2]) == 0)(radix28(rnd.randrange(258 * P256), 11))) == 0


def test_carry_less_sums_leave_room_in_a_product_column():
    """f_addl feeds sums of two normalised values (limbs < 2^29) into f_sqr, and one such sum beside a normalised value into
    f_mul: a column of the CIOS then holds at most S products of limbs plus S reduction products, below 2^64 for every field
    size the library instantiates (9, 10, 15, 21 limbs)."""
    for s in (9, 10, 15, 21):
        lazy, norm = (1 << 29) - 1, (1 << 28) - 1
        top = 1 << 30                                            # the top limb of a lazy value (excess of a 264 p value) is tiny; be generous
        assert s * lazy * lazy + s * norm * norm + (1 << 40) < 1 << 64
        assert s * lazy * norm + top * top + s * norm * norm + (1 << 40) < 1 << 64

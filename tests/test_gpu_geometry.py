"""GPU suite: the two execution geometries of 2048-bit moduli -- one element per lane (Cfg<74, 1>) and the WIDE one for
small arrays (Cfg<76, 4>: the same rows, four lanes per element, vmn_ctx_set_small_array_threshold) -- must give the same
bits.  The parity cases of the other modules run here once with every launch forced into each geometry (by default the
choice depends on the size of the array, so a suite of small cases would only ever see the wide one)."""
import os

import pytest

import test_gpu_parity as tp
import test_gpu_proofs as tpr
from test_gpu_parity import groups          # noqa: F401  (fixtures)
from test_gpu_proofs import mods            # noqa: F401
from oracle import pyref

pytestmark = pytest.mark.gpu

DEFAULT = int(os.environ.get("VMN_WIDE_MAX", 40960))


@pytest.fixture(params=["one-lane", "wide"])
def forced_geometry(request, gpu_ctx):
    gpu_ctx.set_small_array_threshold(0 if request.param == "one-lane" else 2 ** 63)
    yield request.param
    gpu_ctx.set_small_array_threshold(DEFAULT)


def test_golden_vectors_2048(forced_geometry, groups):
    tp.test_golden_vectors_through_c_abi(2048, groups)


@pytest.mark.parametrize("n", [257, 5000])
def test_seeded_arrays_2048(n, forced_geometry, groups, oracle_for):
    tp.test_seeded_arrays_against_gmp_oracle(2048, n, groups, oracle_for)


def test_worst_case_columns_2048(forced_geometry, vmn, gpu_ctx):
    tp.test_worst_case_column_magnitudes(2048, vmn, gpu_ctx)


def test_membership_and_bytetrees_2048(forced_geometry, groups):
    tp.test_subgroup_membership_by_jacobi_symbol(2048, groups)
    G, grp, _ = groups[2048]
    xs = [pow(grp["g"], 3 + k, grp["p"]) for k in range(70)]
    X = G.toElementArray(xs)
    assert G.toElementArrayFromByteTree(X.toByteTree()).toInts() == xs


def test_pos_transcript_2048(forced_geometry, vmn, gpu_ctx, mods):
    tpr.test_pos_transcript_matches_oracle(2048, 130, 1, (256, 256, 100), vmn, gpu_ctx, mods, mods["native"])


def test_geometries_agree_across_the_threshold(vmn, gpu_ctx, groups):
    """The same arrays through both geometries in one process: a fixed-base power, a power with per-element exponents, a
    multi-exponentiation and the two scans, element for element -- at a size above the default threshold too."""
    G, grp, _ = groups[2048]
    p, q, g = grp["p"], grp["q"], grp["g"]
    n = DEFAULT + 1000
    es = pyref.stream_ints(b"geom/e", n, q)
    fs = pyref.stream_ints(b"geom/f", n, 1 << 300)
    out = {}
    for name, thr in (("one-lane", 0), ("wide", 2 ** 63)):
        gpu_ctx.set_small_array_threshold(thr)
        try:
            E, F = G.ringArray(es), G.ringArray(fs)
            X = G.exp(g, E)
            Y = X.exp(F)
            x, d = E.recLin(F)
            out[name] = (X.toInts(), Y.toInts(), X.expProd(F), x.toInts(), d, F.prods().toInts(), X.prod())
        finally:
            gpu_ctx.set_small_array_threshold(DEFAULT)
    assert out["one-lane"] == out["wide"]
    assert out["wide"][0][:64] == [pow(g, e, p) for e in es[:64]]

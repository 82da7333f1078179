"""GPU suite: the two execution geometries of 2048- and 3072-bit moduli -- the base one (Cfg<74, 1> / Cfg<110, 2>) and the
WIDE ones for small arrays (Cfg<76, 4> / Cfg<112, 4>: the same rows, four lanes per element,
vmn_ctx_set_small_array_threshold; Cfg<80, 8>: eight lanes, vmn_ctx_set_tiny_array_threshold) -- must give the same bits.  The parity cases of the other modules run here once with every launch forced into each geometry (by default the
choice depends on the size of the array, so a suite of small cases would only ever see the wide one)."""
import os

import pytest

import test_gpu_parity as tp
import test_gpu_proofs as tpr
import test_gpu_configs as tcf
from test_gpu_parity import groups          # noqa: F401  (fixtures)
from test_gpu_proofs import mods            # noqa: F401
from oracle import pyref

pytestmark = pytest.mark.gpu

DEFAULT = int(os.environ.get("VMN_WIDE_MAX", 40960))
DEFAULT8 = int(os.environ.get("VMN_WIDE8_MAX", 6144))
FORCE = {"base": (0, 0), "wide": (2 ** 63, 0), "wide8": (2 ** 63, 2 ** 63)}     # (small, tiny) thresholds; wide8 exists at 2048 bits only


def force(gpu_ctx, name):
    small, tiny = FORCE[name]
    gpu_ctx.set_small_array_threshold(small)
    gpu_ctx.set_tiny_array_threshold(tiny)


def restore(gpu_ctx):
    gpu_ctx.set_small_array_threshold(DEFAULT)
    gpu_ctx.set_tiny_array_threshold(DEFAULT8)


@pytest.fixture(params=["base", "wide", "wide8"])
def forced_geometry(request, gpu_ctx):
    force(gpu_ctx, request.param)
    yield request.param
    restore(gpu_ctx)


@pytest.mark.parametrize("bits", [2048, 3072])
def test_golden_vectors(bits, forced_geometry, groups):
    tp.test_golden_vectors_through_c_abi(bits, groups)


@pytest.mark.parametrize("bits,n", [(2048, 257), (2048, 1500), (3072, 129), (3072, 600)])       # (the geometry is forced, not chosen by n)
def test_seeded_arrays(bits, n, forced_geometry, groups, oracle_for):
    tp.test_seeded_arrays_against_gmp_oracle(bits, n, groups, oracle_for)


@pytest.mark.parametrize("bits", [2048, 3072])
def test_worst_case_columns(bits, forced_geometry, vmn, gpu_ctx):
    tp.test_worst_case_column_magnitudes(bits, vmn, gpu_ctx)


def test_config2_3072bit_ccpos_flow_in_the_base_geometry(vmn, gpu_ctx, mods):
    """(tests/test_gpu_configs.py runs this flow with the default thresholds, i.e. wide at its size.)"""
    force(gpu_ctx, "base")
    try:
        tcf.test_config2_3072bit_precompute_shrink_ccpos("native", vmn, gpu_ctx, mods)
    finally:
        restore(gpu_ctx)


@pytest.mark.parametrize("bits", [2048, 3072])
def test_membership_and_bytetrees(bits, forced_geometry, groups):
    tp.test_subgroup_membership_by_jacobi_symbol(bits, groups)
    G, grp, _ = groups[bits]
    xs = [pow(grp["g"], 3 + k, grp["p"]) for k in range(70)]
    X = G.toElementArray(xs)
    assert G.toElementArrayFromByteTree(X.toByteTree()).toInts() == xs


@pytest.mark.parametrize("name", ["base", "wide"])
def test_pos_transcript_2048(name, vmn, gpu_ctx, mods):
    """(tests/test_gpu_proofs.py runs the transcripts with the default thresholds, i.e. eight lanes per element at their size.)"""
    force(gpu_ctx, name)
    try:
        tpr.test_pos_transcript_matches_oracle(2048, 70, 1, (256, 256, 100), vmn, gpu_ctx, mods, mods["native"])
    finally:
        restore(gpu_ctx)


@pytest.mark.parametrize("bits", [2048, 3072])
def test_geometries_agree_across_the_threshold(bits, vmn, gpu_ctx, groups):
    """The same arrays through both geometries in one process: a fixed-base power, a power with per-element exponents, a
    multi-exponentiation and the two scans, element for element -- at a size above the default threshold too."""
    G, grp, _ = groups[bits]
    p, q, g = grp["p"], grp["q"], grp["g"]
    n = DEFAULT + 1000
    es = pyref.stream_ints(b"geom/e", n, q)
    fs = pyref.stream_ints(b"geom/f", n, 1 << 300)
    out = {}
    for name in ("base", "wide") + (("wide8",) if bits == 2048 else ()):
        force(gpu_ctx, name)
        try:
            E, F = G.ringArray(es), G.ringArray(fs)
            X = G.exp(g, E)
            Y = X.exp(F)
            x, d = E.recLin(F)
            out[name] = (X.toInts(), Y.toInts(), X.expProd(F), x.toInts(), d, F.prods().toInts(), X.prod())
        finally:
            restore(gpu_ctx)
    assert out["base"] == out["wide"] and out.get("wide8", out["base"]) == out["base"]
    assert out["wide"][0][:64] == [pow(g, e, p) for e in es[:64]]

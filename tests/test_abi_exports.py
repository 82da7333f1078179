"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/vmnhip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vmnhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vmn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path_surface():
    syms = declared_symbols()
    for must in ("vmn_garray_exp_array", "vmn_garray_exp_scalar", "vmn_group_exp_fixed", "vmn_garray_expprod",
                 "vmn_garray_mul", "vmn_garray_prod", "vmn_garray_equals", "vmn_garray_permute", "vmn_rarray_rec_lin",
                 "vmn_rarray_prods", "vmn_rarray_inner_product", "vmn_group_mul_partials"):
        assert must in syms


def test_library_exports_every_declared_symbol(entry):
    lib = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnhip.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/vmnhip.h but not exported: {missing}"


def test_no_cpu_fallback_without_gpu(vmn):
    """Without a GPU the product must fail loudly (status VMN_ERR_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        vmn.Context(0)
    except vmn.VmnError as e:
        assert e.status == -2
    else:
        raise AssertionError("Context() succeeded without a GPU")


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "verificatum-vmn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "gmp.h" not in src and "libvmnoracle" not in src and "from oracle" not in src \
                    and "import oracle" not in src, f"{f} references the oracle"

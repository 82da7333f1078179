"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/vmnhip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols(header="vmnhip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vmn_[A-Za-z0-9_]+)\s*\(", text)))


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


def test_proof_library_exports_every_declared_symbol(entry):
    """include/vmnproofs.h (the proof-level seam S1): C++ drivers in libvmnproofs.so, linked against libvmnhip.so."""
    lib = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnproofs.so"))
    syms = declared_symbols("vmnproofs.h")
    for must in ("vmn_pos_commit", "vmn_pos_verify", "vmn_posc_verify", "vmn_ccpos_compute_ab", "vmn_shuffle_reencrypt",
                 "vmn_permutation_commitment", "vmn_msg_to_bytetree"):
        assert must in syms
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/vmnproofs.h but not exported: {missing}"


def test_proof_library_reports_misuse_without_a_gpu(entry):
    """Argument errors of the proof-level ABI come back as status codes with a message (no GPU needed)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnproofs.so"))
    hip = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnhip.so"))
    hip.vmn_last_error.restype = ctypes.c_char_p
    out = ctypes.c_void_p()
    assert lib.vmn_pos_create(None, 256, 256, 100, None, ctypes.byref(out)) == -1
    assert b"vmn_pos_create" in hip.vmn_last_error()
    assert lib.vmn_pos_commit(None, ctypes.byref(out)) == -1
    m = ctypes.c_void_p()
    assert lib.vmn_msg_create(ctypes.byref(m)) == 0
    assert lib.vmn_msg_push_ring(m, b"\x00\x01", ctypes.c_size_t(1), ctypes.c_size_t(2)) == 0
    lib.vmn_msg_items.restype = ctypes.c_size_t
    assert lib.vmn_msg_items(m) == 1 and lib.vmn_msg_item_kind(m, ctypes.c_size_t(0)) == 4
    lib.vmn_msg_bytetree_size.restype = ctypes.c_size_t
    size = lib.vmn_msg_bytetree_size(m)
    buf = ctypes.create_string_buffer(size)
    assert lib.vmn_msg_to_bytetree(m, buf) == 0
    assert buf.raw == bytes.fromhex("0000000001" "0100000002" "0001")      # node(1 child) | leaf(2 bytes)
    lib.vmn_msg_free(m)


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

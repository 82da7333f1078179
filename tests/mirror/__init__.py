"""TEST-ONLY: Python mirror of the reference's proof callers (hvzk.py, mixnet.py, elgamal.py, parallel_mirror.py).

The product's drivers are C++ (csrc/vmnproofs.cpp behind include/vmnproofs.h; Python side: the ctypes bindings of
``verificatum_vmn_amd.native``).  These modules restate the same callers in Python against the array interface, line by line
with the reference's classes, and exist to cross-check the C++ drivers (same tape -> same transcript) and to run the host logic
on the integer-backed stand-in of tests/fake_backend.py without a GPU.  Nothing under ``verificatum-vmn_amd/`` imports them.
"""
import importlib

MIRROR = ("hvzk", "mixnet", "elgamal", "parallel_mirror")


def load(entry, names):
    """{name: module} for mirror modules (this package) and product modules (``verificatum_vmn_amd.<name>``) alike."""
    entry.load_package()
    out = {}
    for name in names:
        out[name] = importlib.import_module(f"mirror.{name}" if name in MIRROR else f"verificatum_vmn_amd.{name}")
    return out

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def entry():
    import __graft_entry__ as e
    e.build()
    return e


@pytest.fixture(scope="session")
def vmn(entry):
    return entry.load_package()


def load_golden(bits):
    with open(os.path.join(ROOT, "tests", "golden", f"modp{bits}.json")) as f:
        rec = json.load(f)
    grp = {k: (int(v, 16) if k in ("p", "q", "g") else v) for k, v in rec["group"].items()}
    return grp, rec["cases"]


def ints(hex_list):
    return [int(h, 16) for h in hex_list]


@pytest.fixture(scope="session")
def oracle_for(entry):
    from oracle.cbind import Oracle
    cache = {}

    def get(p, q):
        if p not in cache:
            cache[p] = Oracle(p, q)
        return cache[p]
    return get


@pytest.fixture(scope="session")
def gpu_ctx(vmn):
    return vmn.Context(0)


@pytest.fixture(autouse=True)
def gmp_backed_pyref(request, monkeypatch):
    """GPU suite only: the four array exponentiations of oracle/pyref.py run in the C + GMP oracle (tests/fast_pyref.py).
    Not active in the CPU suite, which pins the two oracles against each other and against the golden vectors."""
    if request.node.get_closest_marker("gpu") is None:
        return
    import fast_pyref
    for name, fn in fast_pyref.replacements().items():
        monkeypatch.setattr(fast_pyref.pyref, name, fn)

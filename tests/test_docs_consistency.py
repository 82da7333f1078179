"""CPU suite: the measurement tables of DESIGN.md §6 are generated from the committed profiles (tools/design_tables.py);
they must be the ones the committed profiles give, and the PMC summary must belong to the kernels in the tree."""
import importlib.util
import json
import os

from conftest import ROOT


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_design_tables_are_those_of_the_committed_bench_line():
    dt = _load(os.path.join(ROOT, "tools", "design_tables.py"), "design_tables")
    text = dt.render(dt.DEFAULT)
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = s[s.index(dt.BEGIN) + len(dt.BEGIN):s.index(dt.END)].strip()
    assert block == text.strip(), "DESIGN.md §6 is stale: run python3 tools/design_tables.py"


def test_pmc_summary_belongs_to_the_kernels_in_the_tree():
    """Not fatal (a kernel change is allowed to precede the next profiling run; bench.py then leaves the PMC-derived fields
    null and says why) -- but said out loud."""
    import warnings
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_for_fingerprint")
    pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_kernels.json")))
    if pm["family_fingerprints"]["modp"] != bench.source_fingerprint():
        warnings.warn("profiles/r04_pmc_kernels.json was measured on another build of the headline kernel: rerun tools/profile_pmc.sh")

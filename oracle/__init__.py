"""CPU oracle for the Verificatum Mix-Net hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``verificatum-vmn_amd/``) never does.  See ``vmn_oracle.c`` for the
parity-pinning statement ("parity unpinned" by reference fixtures; pinned by GMP / Python
integers / committed golden vectors).
"""

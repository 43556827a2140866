"""CPU oracle for the PCR-CG hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``pcrcg_amd``) never does: it fails loudly when its HIP library is
missing instead of falling back to anything in here.
"""

"""CPU oracle for the low-bit linear hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker.  See DESIGN.md §3.
"""

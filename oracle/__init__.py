"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the MonoDETR hot path.  Nothing under
``monodetr_amd/`` may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg do, and only as the checker / reported baseline.
"""

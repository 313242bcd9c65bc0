"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference NW-head path (alanqrwang/nwhead @ 2024_08_07).
Nothing under ``nwhead_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker.
"""

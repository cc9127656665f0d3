"""CPU oracle for the FCN-ResNet-50 hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the CPU timing baseline.  The product path
(``neuralbarkcalculator_amd``) never imports this package and fails loudly
when its HIP library is missing.
"""

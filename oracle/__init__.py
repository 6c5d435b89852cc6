"""CPU oracle for the GP edge-tracing hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  The product
path (``gaussian_process_edge_trace_amd``) never imports this package and fails
loudly when the HIP library is missing.
"""

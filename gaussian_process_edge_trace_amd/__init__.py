"""MI355X-native GP edge tracing: drop-in for the hot path of ``gp_edge_tracing``.

Same three exports as the reference package (gp_edge_tracing/__init__.py:10-15), plus the batched tracer.
"""
from .gpet import GP_Edge_Tracing, GP_Edge_Tracing_Batch
from .sklearn_gpr import GaussianProcessRegressor
from . import gpet_utils
from . import _lib
from .sequence import SequenceTracer, trace_sequence
from .pipeline import run_in_flight

__all__ = ["GP_Edge_Tracing", "GaussianProcessRegressor", "gpet_utils", "GP_Edge_Tracing_Batch", "SequenceTracer",
           "trace_sequence", "run_in_flight"]

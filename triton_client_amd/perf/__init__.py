"""perf_analyzer-equivalent load generator (``python -m triton_client_amd.perf``)."""

from .analyzer import Analyzer, Measurement, build_parser, main  # noqa: F401

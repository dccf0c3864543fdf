"""Deprecated alias of ``tritonclient.grpc`` (kept for source compatibility, reference
``src/python/library/tritongrpcclient/__init__.py``)."""

import warnings

warnings.simplefilter("always", DeprecationWarning)
warnings.warn(
    "The package `tritongrpcclient` is deprecated and will be removed in a future version. "
    "Please use instead `tritonclient.grpc`",
    DeprecationWarning,
)

from tritonclient.grpc import *  # noqa: E402,F401,F403

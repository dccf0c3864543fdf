"""Deprecated alias of ``tritonclient.http`` (kept for source compatibility, reference
``src/python/library/tritonhttpclient/__init__.py``)."""

import warnings

warnings.simplefilter("always", DeprecationWarning)
warnings.warn(
    "The package `tritonhttpclient` is deprecated and will be removed in a future version. "
    "Please use instead `tritonclient.http`",
    DeprecationWarning,
)

from tritonclient.http import *  # noqa: E402,F401,F403

"""Deprecated alias of ``tritonclient.utils`` (kept for source compatibility, reference
``src/python/library/tritonclientutils/__init__.py``)."""

import warnings

warnings.simplefilter("always", DeprecationWarning)
warnings.warn(
    "The package `tritonclientutils` is deprecated and will be removed in a future version. "
    "Please use instead `tritonclient.utils`",
    DeprecationWarning,
)

from tritonclient.utils import *  # noqa: E402,F401,F403

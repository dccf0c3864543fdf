"""Deprecated: use ``tritonclient.utils.shared_memory``."""

from tritonclient.utils.shared_memory import *  # noqa: F401,F403

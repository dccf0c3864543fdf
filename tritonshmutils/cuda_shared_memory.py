"""Deprecated: use ``tritonclient.utils.cuda_shared_memory`` (hipshm on this stack)."""

from tritonclient.utils.cuda_shared_memory import *  # noqa: F401,F403

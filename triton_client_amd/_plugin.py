"""Plugin protocol: a callable that may edit request headers in place.

Reference: ``tritonclient/_plugin.py:31-48``.
"""

from abc import ABC, abstractmethod


class InferenceServerClientPlugin(ABC):
    """Subclass and implement ``__call__(request)``; ``request.headers`` is a
    dict that is sent with the network call after the plugin returns."""

    @abstractmethod
    def __call__(self, request):
        """Inspect / mutate ``request`` (a :class:`Request`)."""

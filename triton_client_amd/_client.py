"""Behaviour shared by the four client flavours: the single-plugin slot.

Reference: ``tritonclient/_client.py:31-85``.
"""

from ._plugin import InferenceServerClientPlugin
from ._request import Request
from .utils import raise_error


class InferenceServerClientBase:
    def __init__(self):
        self._plugin = None

    def _call_plugin(self, request):
        """Run the registered plugin (if any) on ``request``."""
        if self._plugin is not None:
            self._plugin(request)

    def _apply_plugin(self, headers):
        """Copy ``headers``, let the plugin edit the copy, return it."""
        request = Request(dict(headers) if headers else {})
        self._call_plugin(request)
        return request.headers

    def register_plugin(self, plugin):
        """Install ``plugin``; only one may be active at a time."""
        if self._plugin is not None:
            raise_error(
                "A plugin is already registered. Please unregister the previous plugin first before"
                " registering a new plugin."
            )
        if not callable(plugin):
            raise_error("plugin must be callable (see InferenceServerClientPlugin)")
        self._plugin = plugin

    def plugin(self):
        """The active plugin or ``None``."""
        return self._plugin

    def unregister_plugin(self):
        """Remove the active plugin."""
        if self._plugin is None:
            raise_error("No plugin has been registered.")
        self._plugin = None


__all__ = ["InferenceServerClientBase", "InferenceServerClientPlugin", "Request"]

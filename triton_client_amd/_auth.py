"""HTTP Basic authentication plugin (reference ``tritonclient/_auth.py:33-45``)."""

import base64

from ._plugin import InferenceServerClientPlugin


class BasicAuth(InferenceServerClientPlugin):
    """Adds ``authorization: Basic base64(username:password)`` to every call."""

    def __init__(self, username, password):
        token = base64.b64encode(f"{username}:{password}".encode("utf-8")).decode("ascii")
        self._header_value = "Basic " + token

    def __call__(self, request):
        request.headers["authorization"] = self._header_value

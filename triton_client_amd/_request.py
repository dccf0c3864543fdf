"""Mutable view of an outgoing request handed to plugins.

Reference: ``tritonclient/_request.py:29-39`` – only the headers are exposed.
"""


class Request:
    __slots__ = ("headers",)

    def __init__(self, headers):
        self.headers = headers

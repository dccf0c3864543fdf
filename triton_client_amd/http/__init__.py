"""HTTP/REST client (synchronous).  ``triton_client_amd.http.aio`` has the
asyncio flavour.  Export list parity: ``tritonclient/http/__init__.py:44-53``."""

from .._client import InferenceServerClientPlugin
from .._request import Request
from ..utils import *  # noqa: F401,F403  (the reference re-exports utils here)
from ..utils import InferenceServerException
from ._client import InferAsyncRequest, InferenceServerClient
from ._infer_input import InferInput
from ._infer_result import InferResult
from ._requested_output import InferRequestedOutput

__all__ = [
    "InferenceServerClientPlugin",
    "Request",
    "InferenceServerClient",
    "InferInput",
    "InferRequestedOutput",
    "InferResult",
    "InferAsyncRequest",
    "InferenceServerException",
]

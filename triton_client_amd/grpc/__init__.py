"""gRPC client (synchronous).  ``triton_client_amd.grpc.aio`` has the asyncio
flavour.  Export list parity: ``tritonclient/grpc/__init__.py:48-73``."""

from .._client import InferenceServerClientPlugin
from .._request import Request
from ..utils import *  # noqa: F401,F403
from ..utils import InferenceServerException
from . import model_config_pb2, service_pb2, service_pb2_grpc
from ._client import MAX_GRPC_MESSAGE_SIZE, CallContext, InferenceServerClient, KeepAliveOptions
from ._infer_input import InferInput
from ._infer_result import InferResult
from ._requested_output import InferRequestedOutput
from ._utils import raise_error_grpc

__all__ = [
    "InferenceServerClientPlugin",
    "Request",
    "InferenceServerClient",
    "InferInput",
    "InferRequestedOutput",
    "InferResult",
    "KeepAliveOptions",
    "InferenceServerException",
]

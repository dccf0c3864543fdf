"""Request/response messages of ``GRPCInferenceService`` (built at import time)."""

from . import _pb
from . import model_config_pb2 as model__config__pb2  # noqa: F401  (dependency, as protoc would import)

DESCRIPTOR = _pb.FILES["grpc_service.proto"]

globals().update(_pb.message_classes("grpc_service.proto"))

"""Deprecated: use ``tritonclient.grpc.service_pb2``."""

from tritonclient.grpc.service_pb2 import *  # noqa: F401,F403

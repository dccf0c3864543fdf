"""Deprecated: use ``tritonclient.grpc.service_pb2_grpc``."""

from tritonclient.grpc.service_pb2_grpc import *  # noqa: F401,F403

"""Deprecated: use ``tritonclient.grpc.model_config_pb2``."""

from tritonclient.grpc.model_config_pb2 import *  # noqa: F401,F403

"""Runtime construction of the protocol's protobuf classes.

The two ``.proto`` files in ``protos/`` are compiled by :mod:`._protoc`, added
to a private descriptor pool and turned into message classes with
``message_factory``.  ``model_config_pb2`` / ``service_pb2`` /
``service_pb2_grpc`` re-export the result under the names generated code would
have (reference: the wheel copies protoc output to
``tritonclient/grpc/{model_config_pb2,service_pb2,service_pb2_grpc}.py``,
``build_wheel.py:125-153``).
"""

from __future__ import annotations

import os

from google.protobuf import descriptor_pool, message_factory

from . import _protoc

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORDER = ("model_config.proto", "grpc_service.proto")


def _load():
    sources = {}
    for name in _ORDER:
        with open(os.path.join(_HERE, "protos", name), "r", encoding="utf-8") as fh:
            sources[name] = fh.read()
    pool = descriptor_pool.DescriptorPool()
    files = {}
    for fdp in _protoc.compile_protos(sources):
        pool.Add(fdp)
        files[fdp.name] = pool.FindFileByName(fdp.name)
    return pool, files


POOL, FILES = _load()


def message_classes(file_name):
    """``{MessageName: class}`` for the top-level messages of one file."""
    fd = FILES[file_name]
    return {name: message_factory.GetMessageClass(desc) for name, desc in fd.message_types_by_name.items()}


def enum_wrappers(file_name):
    from google.protobuf.internal import enum_type_wrapper

    fd = FILES[file_name]
    return {name: enum_type_wrapper.EnumTypeWrapper(desc) for name, desc in fd.enum_types_by_name.items()}


SERVICE = FILES["grpc_service.proto"].services_by_name["GRPCInferenceService"]

"""Client stub, servicer base and server registration for
``inference.GRPCInferenceService``, derived from the service descriptor.

Equivalent to what ``grpc_tools`` would emit: ``GRPCInferenceServiceStub``,
``GRPCInferenceServiceServicer`` and
``add_GRPCInferenceServiceServicer_to_server``.
"""

from __future__ import annotations

import grpc
from google.protobuf import message_factory

from . import _pb

_SERVICE_NAME = _pb.SERVICE.full_name  # "inference.GRPCInferenceService"


def _methods():
    for m in _pb.SERVICE.methods:
        req = message_factory.GetMessageClass(m.input_type)
        resp = message_factory.GetMessageClass(m.output_type)
        yield m.name, req, resp, m.client_streaming, m.server_streaming


_KIND = {
    (False, False): "unary_unary",
    (False, True): "unary_stream",
    (True, False): "stream_unary",
    (True, True): "stream_stream",
}


class GRPCInferenceServiceStub:
    """One multicallable attribute per RPC, bound to ``channel``."""

    def __init__(self, channel):
        for name, req, resp, cs, ss in _methods():
            factory = getattr(channel, _KIND[(cs, ss)])
            setattr(
                self,
                name,
                factory(
                    f"/{_SERVICE_NAME}/{name}",
                    request_serializer=req.SerializeToString,
                    response_deserializer=resp.FromString,
                ),
            )


def _unimplemented(name):
    def handler(self, request, context):
        context.set_code(grpc.StatusCode.UNIMPLEMENTED)
        context.set_details(f"Method {name} not implemented!")
        raise NotImplementedError(f"Method {name} not implemented!")

    handler.__name__ = name
    return handler


class GRPCInferenceServiceServicer:
    """Base class: every RPC answers UNIMPLEMENTED until overridden."""


for _name, *_ in _methods():
    setattr(GRPCInferenceServiceServicer, _name, _unimplemented(_name))
del _name, _


def add_GRPCInferenceServiceServicer_to_server(servicer, server):
    handlers = {}
    for name, req, resp, cs, ss in _methods():
        make = getattr(grpc, _KIND[(cs, ss)] + "_rpc_method_handler")
        handlers[name] = make(
            getattr(servicer, name),
            request_deserializer=req.FromString,
            response_serializer=resp.SerializeToString,
        )
    server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler(_SERVICE_NAME, handlers),))

#!/usr/bin/env python3
"""Talks to the server with the generated-style stub classes directly (no
client library): liveness, readiness, metadata, config and one inference with
raw input contents.  (Reference: ``src/python/examples/grpc_client.py``.)"""
import grpc
import numpy as np
from _common import fail, parser

from tritonclient.grpc import service_pb2, service_pb2_grpc

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    channel = grpc.insecure_channel(flags.url)
    stub = service_pb2_grpc.GRPCInferenceServiceStub(channel)
    print("server {}".format(stub.ServerLive(service_pb2.ServerLiveRequest())))
    print("server {}".format(stub.ServerReady(service_pb2.ServerReadyRequest())))
    print("model {}".format(stub.ModelReady(service_pb2.ModelReadyRequest(name="simple", version=""))))
    print("server metadata:\n{}".format(stub.ServerMetadata(service_pb2.ServerMetadataRequest())))
    print("model metadata:\n{}".format(stub.ModelMetadata(service_pb2.ModelMetadataRequest(name="simple"))))
    print("model config:\n{}".format(stub.ModelConfig(service_pb2.ModelConfigRequest(name="simple"))))

    request = service_pb2.ModelInferRequest(model_name="simple", model_version="", id="my request id")
    a = np.arange(16, dtype=np.int32)
    b = np.ones(16, dtype=np.int32)
    for name, data in (("INPUT0", a), ("INPUT1", b)):
        tensor = service_pb2.ModelInferRequest.InferInputTensor(name=name, datatype="INT32", shape=[1, 16])
        request.inputs.append(tensor)
        request.raw_input_contents.append(data.tobytes())
    for name in ("OUTPUT0", "OUTPUT1"):
        request.outputs.append(service_pb2.ModelInferRequest.InferRequestedOutputTensor(name=name))
    response = stub.ModelInfer(request)
    if response.id != "my request id":
        fail("request id was not echoed")
    outputs = {o.name: np.frombuffer(response.raw_output_contents[i], dtype=np.int32)
               for i, o in enumerate(response.outputs)}
    if not np.array_equal(outputs["OUTPUT0"], a + b) or not np.array_equal(outputs["OUTPUT1"], a - b):
        fail("incorrect result")
    print("model infer:\n{}".format([o.name for o in response.outputs]))
    print("PASS")

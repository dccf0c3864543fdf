#!/usr/bin/env python3
"""Inference through the raw gRPC stub with inputs carried in the typed
``contents.bytes_contents`` field instead of ``raw_input_contents``; also checks that a
non-numeric model version is rejected.  (Reference:
``src/python/examples/grpc_explicit_byte_content_client.py``.)"""
import grpc
import numpy as np
from _common import fail, parser

from tritonclient.grpc import service_pb2, service_pb2_grpc
from tritonclient.utils import deserialize_bytes_tensor

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    stub = service_pb2_grpc.GRPCInferenceServiceStub(grpc.insecure_channel(flags.url))
    request = service_pb2.ModelInferRequest(model_name="simple_string", model_version="")
    in0 = list(range(16))
    in1 = [1] * 16
    for name, values in (("INPUT0", in0), ("INPUT1", in1)):
        tensor = service_pb2.ModelInferRequest.InferInputTensor(name=name, datatype="BYTES", shape=[1, 16])
        tensor.contents.bytes_contents[:] = [str(v).encode() for v in values]
        request.inputs.append(tensor)
    for name in ("OUTPUT0", "OUTPUT1"):
        request.outputs.append(service_pb2.ModelInferRequest.InferRequestedOutputTensor(name=name))
    response = stub.ModelInfer(request)
    if len(response.outputs) != 2:
        fail("expected two output results")
    results = [deserialize_bytes_tensor(response.raw_output_contents[i]) for i in range(2)]
    for i in range(16):
        print(f"{in0[i]} + {in1[i]} = {results[0][i]}")
        print(f"{in0[i]} - {in1[i]} = {results[1][i]}")
        if in0[i] + in1[i] != int(results[0][i]):
            fail("sync infer error: incorrect sum")
        if in0[i] - in1[i] != int(results[1][i]):
            fail("sync infer error: incorrect difference")
    request.model_version = "wrong_specification"
    try:
        stub.ModelInfer(request)
        fail("expected the wrong version string to be rejected")
    except grpc.RpcError as e:
        if "failed to get model version from specified version string 'wrong_specification'" not in str(e):
            fail("unexpected error: " + str(e))
    print("PASS: explicit byte")

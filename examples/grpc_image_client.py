#!/usr/bin/env python3
"""Image classification through the raw gRPC stub (no client library): model
metadata/config via stub calls, tensors in ``raw_input_contents`` and the
classification extension requested with an output parameter; ``--streaming``
drives ``ModelStreamInfer`` with a request iterator.  (Reference:
``src/python/examples/grpc_image_client.py``.)"""
import grpc
import numpy as np
from _common import fail, parser
from _image_example import batches, decode, host_preprocess, list_images, parse_model

from tritonclient.grpc import service_pb2, service_pb2_grpc
from tritonclient.utils import deserialize_bytes_tensor

if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("-a", "--async", dest="async_set", action="store_true")
    ap.add_argument("--streaming", action="store_true")
    ap.add_argument("-m", "--model-name", required=True)
    ap.add_argument("-x", "--model-version", default="")
    ap.add_argument("-b", "--batch-size", type=int, default=1)
    ap.add_argument("-c", "--classes", type=int, default=1)
    ap.add_argument("-s", "--scaling", choices=["NONE", "INCEPTION", "VGG"], default="NONE")
    ap.add_argument("image_filename", nargs="?", default=None)
    flags = ap.parse_args()
    if flags.image_filename is None:
        fail("an image file or folder is required")
    stub = service_pb2_grpc.GRPCInferenceServiceStub(grpc.insecure_channel(flags.url))
    try:
        metadata = stub.ModelMetadata(service_pb2.ModelMetadataRequest(name=flags.model_name,
                                                                       version=flags.model_version))
        config = stub.ModelConfig(service_pb2.ModelConfigRequest(name=flags.model_name, version=flags.model_version))
    except grpc.RpcError as e:
        fail("failed to query the model: " + str(e))
    spec = parse_model(metadata, config)
    batched = spec.max_batch_size > 0
    if not batched and flags.batch_size != 1:
        fail("ERROR: This model doesn't support batching.")
    filenames = list_images(flags.image_filename)
    tensors = [host_preprocess(decode(f, spec.c), spec, flags.scaling) for f in filenames]
    dims = [spec.c, spec.h, spec.w] if spec.layout == "NCHW" else [spec.h, spec.w, spec.c]

    requests, names = [], []
    for sent, batch in enumerate(batches(filenames, flags.batch_size), start=1):
        request = service_pb2.ModelInferRequest(model_name=flags.model_name, model_version=flags.model_version,
                                                id=str(sent))
        request.inputs.append(service_pb2.ModelInferRequest.InferInputTensor(
            name=spec.input_name, datatype=spec.dtype, shape=([len(batch)] if batched else []) + dims))
        output = service_pb2.ModelInferRequest.InferRequestedOutputTensor(name=spec.output_name)
        output.parameters["classification"].int64_param = flags.classes
        request.outputs.append(output)
        data = np.stack([tensors[i] for i in batch]) if batched else tensors[batch[0]]
        request.raw_input_contents.append(data.tobytes())
        requests.append(request)
        names.append([filenames[i] for i in batch])

    responses = []
    try:
        if flags.streaming:
            for item in stub.ModelStreamInfer(iter(requests)):
                if item.error_message:
                    fail("inference failed: " + item.error_message)
                responses.append(item.infer_response)
        elif flags.async_set:
            futures = [stub.ModelInfer.future(r) for r in requests]
            responses = [f.result() for f in futures]
        else:
            responses = [stub.ModelInfer(r) for r in requests]
    except grpc.RpcError as e:
        fail("inference failed: " + str(e))

    for response in sorted(responses, key=lambda r: int(r.id)):
        if len(response.outputs) != 1:
            fail(f"expected 1 output, got {len(response.outputs)}")
        entries = deserialize_bytes_tensor(response.raw_output_contents[0]).reshape(-1, flags.classes)
        if batched and len(entries) != flags.batch_size:
            fail(f"expected {flags.batch_size} results, got {len(entries)}")
        print(f"Request {response.id}, batch size {flags.batch_size}")
        for row, filename in zip(entries, names[int(response.id) - 1]):
            print(f"Image '{filename}':")
            for entry in row:
                value, index, *label = entry.decode("utf-8").split(":")
                print(f"    {float(value):f} ({index}) = {label[0] if label else ''}")
    print("PASS")

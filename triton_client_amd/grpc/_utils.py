"""Error mapping, request assembly and compression selection for gRPC
(reference ``tritonclient/grpc/_utils.py:34-158``)."""

import grpc

from ..utils import InferenceServerException, raise_error
from . import service_pb2

_RESERVED = ("sequence_id", "sequence_start", "sequence_end", "priority", "binary_data_output")


def get_error_grpc(rpc_error):
    """``grpc.RpcError`` -> :class:`InferenceServerException`."""
    debug = None
    try:
        debug = rpc_error.debug_error_string()
    except Exception:
        pass
    return InferenceServerException(msg=rpc_error.details(), status=str(rpc_error.code()), debug_details=debug)


def get_cancelled_error(msg=None):
    return InferenceServerException(msg=msg or "Locally cancelled by application!", status="StatusCode.CANCELLED")


def raise_error_grpc(rpc_error):
    raise get_error_grpc(rpc_error) from None


def _set_parameter(param, key, value):
    # bool must be tested before int: bool is an int subclass
    if isinstance(value, str):
        param.string_param = value
    elif isinstance(value, bool):
        param.bool_param = value
    elif isinstance(value, int):
        param.int64_param = value
    elif isinstance(value, float):
        param.double_param = value
    else:
        raise_error(f'The parameter datatype "{type(value)}" for key "{key}" is not supported.')


def _get_inference_request(model_name, inputs, model_version, request_id, outputs, sequence_id, sequence_start,
                           sequence_end, priority, timeout, parameters):
    request = service_pb2.ModelInferRequest(model_name=model_name, model_version=model_version)
    if request_id != "":
        request.id = request_id
    for tensor in inputs:
        request.inputs.append(tensor._get_tensor())
        content = tensor._get_content()
        if content is not None:
            request.raw_input_contents.append(content)
    if outputs is not None:
        for tensor in outputs:
            request.outputs.append(tensor._get_tensor())
    if sequence_id != 0 and sequence_id != "":
        if isinstance(sequence_id, str):
            request.parameters["sequence_id"].string_param = sequence_id
        else:
            request.parameters["sequence_id"].int64_param = sequence_id
        request.parameters["sequence_start"].bool_param = sequence_start
        request.parameters["sequence_end"].bool_param = sequence_end
    if priority != 0:
        request.parameters["priority"].uint64_param = priority
    if timeout is not None:
        request.parameters["timeout"].int64_param = timeout
    if parameters:
        for key, value in parameters.items():
            if key in _RESERVED:
                raise_error(f'Parameter "{key}" is a reserved parameter and cannot be specified.')
            _set_parameter(request.parameters[key], key, value)
    return request


def _grpc_compression_type(algorithm_str):
    if algorithm_str is None:
        return grpc.Compression.NoCompression
    lowered = algorithm_str.lower()
    if lowered == "deflate":
        return grpc.Compression.Deflate
    if lowered == "gzip":
        return grpc.Compression.Gzip
    print("The provided client-side compression algorithm is not supported... using no compression")
    return grpc.Compression.NoCompression

"""Request assembly and error mapping shared by the sync and asyncio HTTP
clients (reference ``tritonclient/http/_utils.py:35-150``)."""

import json
from urllib.parse import quote_plus

from ..utils import InferenceServerException, raise_error

_RESERVED = ("sequence_id", "sequence_start", "sequence_end", "priority", "binary_data_output")


def _get_error(response):
    """``InferenceServerException`` for a non-200 response, else ``None``."""
    if response.status_code == 200:
        return None
    body = None
    try:
        body = response.read()
        if isinstance(body, (bytes, bytearray, memoryview)):
            body = bytes(body).decode("utf-8")
        payload = json.loads(body) if len(body) else {"error": "client received an empty response from the server."}
        return InferenceServerException(msg=payload["error"], status=str(response.status_code))
    except Exception as e:
        return InferenceServerException(
            msg=f"an exception occurred in the client while decoding the response: {e}",
            status=str(response.status_code),
            debug_details=body,
        )


def _raise_if_error(response):
    error = _get_error(response)
    if error is not None:
        raise error


def _get_query_string(query_params):
    pairs = []
    for key, value in query_params.items():
        values = value if isinstance(value, list) else [value]
        for item in values:
            pairs.append("%s=%s" % (quote_plus(key), quote_plus(str(item))))
    return "&".join(pairs)


def _build_header_dict(inputs, request_id, outputs, sequence_id, sequence_start, sequence_end, priority, timeout,
                       custom_parameters):
    request = {}
    params = {}
    if request_id != "":
        request["id"] = request_id
    if sequence_id != 0 and sequence_id != "":
        params["sequence_id"] = sequence_id
        params["sequence_start"] = sequence_start
        params["sequence_end"] = sequence_end
    if priority != 0:
        params["priority"] = priority
    if timeout is not None:
        params["timeout"] = timeout
    request["inputs"] = [t._get_tensor() for t in inputs]
    if outputs:
        request["outputs"] = [t._get_tensor() for t in outputs]
    else:
        # nothing requested explicitly: ask for every output in binary form
        params["binary_data_output"] = True
    if custom_parameters:
        for key, value in custom_parameters.items():
            if key in _RESERVED:
                raise_error(f'Parameter "{key}" is a reserved parameter and cannot be specified.')
            params[key] = value
    if params:
        request["parameters"] = params
    return request


def _get_inference_request_parts(inputs, request_id, outputs, sequence_id, sequence_start, sequence_end, priority,
                                 timeout, custom_parameters):
    """``([json_bytes, raw0, raw1, ...], json_size | None)`` – no concatenation."""
    header = _build_header_dict(inputs, request_id, outputs, sequence_id, sequence_start, sequence_end, priority,
                                timeout, custom_parameters)
    head = json.dumps(header, separators=(",", ":")).encode("utf-8")
    parts = [head]
    for tensor in inputs:
        raw = tensor._get_binary_data()
        if raw is not None:
            parts.append(raw)
    return parts, (len(head) if len(parts) > 1 else None)


def _get_inference_request(inputs, request_id, outputs, sequence_id, sequence_start, sequence_end, priority, timeout,
                           custom_parameters):
    """``(body_bytes, json_size | None)``; ``json_size`` is ``None`` when the
    body is JSON only (no ``Inference-Header-Content-Length`` needed)."""
    parts, json_size = _get_inference_request_parts(inputs, request_id, outputs, sequence_id, sequence_start,
                                                    sequence_end, priority, timeout, custom_parameters)
    if json_size is None:
        return parts[0], None
    return b"".join(parts), json_size

"""Inference response wrapper for the HTTP/REST transport.

API parity: ``tritonclient/http/_infer_result.py:41-242`` (``as_numpy``,
``get_output``, ``get_response``, ``from_response_body``).  The body is held as
one buffer; fixed-size outputs are returned as zero-copy ``np.frombuffer``
views into it.
"""

import gzip
import json
import zlib

import numpy as np

from ..utils import deserialize_bf16_tensor, deserialize_bytes_tensor, raise_error, triton_to_np_dtype


def _inflate(body, encoding):
    if encoding == "gzip":
        return gzip.decompress(body)
    if encoding == "deflate":
        return zlib.decompress(body)
    return body


class InferResult:
    """Holds a decoded inference response.

    ``response`` needs ``get(header_name)`` and ``read(length=-1)``.
    """

    def __init__(self, response, verbose):
        header_length = response.get("Inference-Header-Content-Length")
        body = _inflate(response.read(), response.get("Content-Encoding"))
        self._init_from_body(body, header_length, verbose)

    def _init_from_body(self, body, header_length, verbose):
        view = memoryview(body)
        if header_length is None:
            head, self._buffer = view, memoryview(b"")
        else:
            n = int(header_length)
            head, self._buffer = view[:n], view[n:]
        if verbose:
            print(bytes(head))
        try:
            self._result = json.loads(bytes(head))
        except UnicodeDecodeError as e:
            raise_error(
                "Failed to encode using UTF-8. Please use binary_data=True, if"
                f" you want to pass a byte array. UnicodeError: {e}"
            )
        # name -> (start, size) of the binary tail slice
        self._spans = {}
        cursor = 0
        for output in self._result.get("outputs", ()):
            params = output.get("parameters")
            if params is not None:
                size = params.get("binary_data_size")
                if size is not None:
                    self._spans[output["name"]] = (cursor, size)
                    cursor += size

    @classmethod
    def from_response_body(cls, response_body, verbose=False, header_length=None, content_encoding=None):
        """Build a result from raw bytes (no socket involved)."""
        self = cls.__new__(cls)
        self._init_from_body(_inflate(response_body, content_encoding), header_length, verbose)
        return self

    def as_numpy(self, name):
        """ndarray for output ``name`` or ``None`` when absent."""
        for output in self._result.get("outputs") or ():
            if output["name"] != name:
                continue
            datatype = output["datatype"]
            span = self._spans.get(name)
            if span is not None:
                start, size = span
                if size == 0:
                    arr = np.empty(0, dtype=triton_to_np_dtype(datatype))
                else:
                    chunk = self._buffer[start : start + size]
                    if datatype == "BYTES":
                        arr = deserialize_bytes_tensor(chunk)
                    elif datatype == "BF16":
                        arr = deserialize_bf16_tensor(chunk)
                    else:
                        arr = np.frombuffer(chunk, dtype=triton_to_np_dtype(datatype))
            elif "data" in output:
                arr = np.array(output["data"], dtype=triton_to_np_dtype(datatype))
            else:
                return None  # written to a shared memory region: nothing travelled in the response
            return arr.reshape(output["shape"])
        return None

    def get_output(self, name):
        """The JSON dict of output ``name`` or ``None``."""
        for output in self._result.get("outputs", ()):
            if output["name"] == name:
                return output
        return None

    def get_response(self):
        """The whole response header as a dict."""
        return self._result

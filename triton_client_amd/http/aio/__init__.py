"""asyncio HTTP/REST client on aiohttp.

Public surface parity: ``tritonclient/http/aio/__init__.py:92-775`` – same
methods as the synchronous client but ``async def``; there is no
``async_infer`` (just ``await infer(...)``).  Request bodies are streamed from
the part list (no concatenation); responses are decompressed here because the
session is opened with ``auto_decompress=False`` like the reference.
"""

from __future__ import annotations

import base64
import gzip
import json
import zlib
from urllib.parse import quote

import aiohttp

from ..._client import InferenceServerClientBase, InferenceServerClientPlugin  # noqa: F401
from ..._request import Request  # noqa: F401
from ...utils import *  # noqa: F401,F403
from ...utils import InferenceServerException, raise_error
from .._infer_input import InferInput  # noqa: F401
from .._infer_result import InferResult
from .._requested_output import InferRequestedOutput  # noqa: F401
from .._utils import _get_inference_request, _get_inference_request_parts, _get_query_string

__all__ = [
    "InferenceServerClientPlugin",
    "Request",
    "InferenceServerClient",
    "InferInput",
    "InferRequestedOutput",
    "InferResult",
    "InferenceServerException",
]


async def _get_error(response):
    if response.status == 200:
        return None
    body = None
    try:
        body = (await response.read()).decode("utf-8")
        payload = json.loads(body) if len(body) else {"error": "client received an empty response from the server."}
        return InferenceServerException(msg=payload["error"], status=str(response.status))
    except Exception as e:
        return InferenceServerException(
            msg=f"an exception occurred in the client while decoding the response: {e}",
            status=str(response.status),
            debug_details=body,
        )


async def _raise_if_error(response):
    error = await _get_error(response)
    if error is not None:
        raise error


class _BufferedResponse:
    """Adapter giving ``InferResult`` the ``get``/``read`` interface it expects."""

    def __init__(self, headers, body):
        self._headers = headers
        self._body = body

    def get(self, name):
        return self._headers.get(name)

    def read(self, length=-1):
        return self._body


async def _parts_stream(parts):
    for part in parts:
        yield bytes(part) if not isinstance(part, (bytes, bytearray)) else part


class InferenceServerClient(InferenceServerClientBase):
    """asyncio client; use as ``async with InferenceServerClient(url) as c``.

    Parameters
    ----------
    url : str
        ``host:port[/base-path]`` without scheme.
    verbose : bool
    conn_limit : int
        Maximum simultaneous connections of the session.
    conn_timeout : float
        Total timeout per request in seconds.
    ssl : bool
    ssl_context : ssl.SSLContext or None
    """

    def __init__(self, url, verbose=False, conn_limit=100, conn_timeout=60.0, ssl=False, ssl_context=None):
        super().__init__()
        if url.startswith("http://") or url.startswith("https://"):
            raise_error("url should not include the scheme")
        self._url = ("https://" if ssl else "http://") + url
        self._url = self._url.rstrip("/")
        self._conn = aiohttp.TCPConnector(ssl=ssl_context if ssl_context is not None else (None if ssl else False),
                                          limit=conn_limit)
        self._stub = aiohttp.ClientSession(connector=self._conn, timeout=aiohttp.ClientTimeout(total=conn_timeout),
                                           auto_decompress=False,
                                           skip_auto_headers=("Accept-Encoding",))
        self._verbose = verbose

    async def __aenter__(self):
        return self

    async def __aexit__(self, exc_type, exc, tb):
        await self.close()

    async def close(self):
        """Close the session and its connector."""
        await self._stub.close()
        await self._conn.close()

    # ------------------------------------------------------------------ I/O
    def _fix_header(self, headers):
        headers = self._apply_plugin(headers)
        if any(k.lower() == "transfer-encoding" for k in headers):
            raise_error(
                "Unsupported HTTP header: 'Transfer-Encoding' is not supported in the Python client library. "
                "Use raw HTTP request libraries or the C++ client instead for this header."
            )
        return {str(k): str(v) for k, v in headers.items()}

    def _full_url(self, request_uri, query_params):
        url = self._url + "/" + request_uri
        if query_params is not None:
            url = url + "?" + _get_query_string(query_params)
        return url

    async def _get(self, request_uri, headers, query_params):
        headers = self._fix_header(headers)
        url = self._full_url(request_uri, query_params)
        if self._verbose:
            print("GET {}, headers {}".format(url, headers))
        response = await self._stub.get(url=url, headers=headers)
        if self._verbose:
            print(response)
        return response

    async def _post(self, request_uri, request_body, headers, query_params):
        headers = self._fix_header(headers)
        url = self._full_url(request_uri, query_params)
        if self._verbose:
            print("POST {}, headers {}\n{}".format(url, headers, request_body))
        if isinstance(request_body, (list, tuple)):
            total = sum(len(p) for p in request_body)
            headers["Content-Length"] = str(total)
            data = _parts_stream(request_body) if len(request_body) > 1 else request_body[0]
        else:
            data = request_body.encode("utf-8") if isinstance(request_body, str) else request_body
        response = await self._stub.post(url=url, data=data, headers=headers)
        if self._verbose:
            print(response)
        return response

    async def _get_json(self, uri, headers, query_params):
        response = await self._get(uri, headers, query_params)
        await _raise_if_error(response)
        content = await response.read()
        result = json.loads(content) if content else {}
        if self._verbose:
            print(result)
        return result

    async def _post_json(self, uri, body, headers, query_params, want_result=True):
        response = await self._post(uri, body, headers, query_params)
        await _raise_if_error(response)
        content = await response.read()
        if not want_result:
            return None
        result = json.loads(content) if content else {}
        if self._verbose:
            print(result)
        return result

    @staticmethod
    def _model_uri(model_name, model_version, suffix=""):
        if type(model_version) != str:
            raise_error("model version must be a string")
        uri = "v2/models/" + quote(model_name)
        if model_version != "":
            uri += "/versions/" + model_version
        return uri + suffix

    # --------------------------------------------------------------- health
    async def is_server_live(self, headers=None, query_params=None):
        return (await self._get("v2/health/live", headers, query_params)).status == 200

    async def is_server_ready(self, headers=None, query_params=None):
        return (await self._get("v2/health/ready", headers, query_params)).status == 200

    async def is_model_ready(self, model_name, model_version="", headers=None, query_params=None):
        uri = self._model_uri(model_name, model_version, "/ready")
        return (await self._get(uri, headers, query_params)).status == 200

    # ------------------------------------------------------------- metadata
    async def get_server_metadata(self, headers=None, query_params=None):
        return await self._get_json("v2", headers, query_params)

    async def get_model_metadata(self, model_name, model_version="", headers=None, query_params=None):
        return await self._get_json(self._model_uri(model_name, model_version), headers, query_params)

    async def get_model_config(self, model_name, model_version="", headers=None, query_params=None):
        return await self._get_json(self._model_uri(model_name, model_version, "/config"), headers, query_params)

    # ----------------------------------------------------- model repository
    async def get_model_repository_index(self, headers=None, query_params=None):
        return await self._post_json("v2/repository/index", "", headers, query_params)

    async def load_model(self, model_name, headers=None, query_params=None, config=None, files=None):
        body = {}
        if config is not None:
            body.setdefault("parameters", {})["config"] = config
        if files is not None:
            for path, content in files.items():
                body.setdefault("parameters", {})[path] = base64.b64encode(content).decode("ascii")
        await self._post_json("v2/repository/models/{}/load".format(quote(model_name)), json.dumps(body), headers,
                              query_params, want_result=False)
        if self._verbose:
            print("Loaded model '{}'".format(model_name))

    async def unload_model(self, model_name, headers=None, query_params=None, unload_dependents=False):
        body = {"parameters": {"unload_dependents": unload_dependents}}
        await self._post_json("v2/repository/models/{}/unload".format(quote(model_name)), json.dumps(body), headers,
                              query_params, want_result=False)
        if self._verbose:
            print("Unloaded model '{}'".format(model_name))

    # ----------------------------------------------------------- statistics
    async def get_inference_statistics(self, model_name="", model_version="", headers=None, query_params=None):
        uri = self._model_uri(model_name, model_version, "/stats") if model_name != "" else "v2/models/stats"
        return await self._get_json(uri, headers, query_params)

    # ---------------------------------------------------------------- trace
    async def update_trace_settings(self, model_name=None, settings={}, headers=None, query_params=None):
        uri = "v2/models/{}/trace/setting".format(quote(model_name)) if model_name else "v2/trace/setting"
        return await self._post_json(uri, json.dumps(settings), headers, query_params)

    async def get_trace_settings(self, model_name=None, headers=None, query_params=None):
        uri = "v2/models/{}/trace/setting".format(quote(model_name)) if model_name else "v2/trace/setting"
        return await self._get_json(uri, headers, query_params)

    # ------------------------------------------------------------------ log
    async def update_log_settings(self, settings, headers=None, query_params=None):
        return await self._post_json("v2/logging", json.dumps(settings), headers, query_params)

    async def get_log_settings(self, headers=None, query_params=None):
        return await self._get_json("v2/logging", headers, query_params)

    # -------------------------------------------------- system shared memory
    async def get_system_shared_memory_status(self, region_name="", headers=None, query_params=None):
        uri = "v2/systemsharedmemory/region/{}/status".format(quote(region_name)) if region_name != "" \
            else "v2/systemsharedmemory/status"
        return await self._get_json(uri, headers, query_params)

    async def register_system_shared_memory(self, name, key, byte_size, offset=0, headers=None, query_params=None):
        body = {"key": key, "offset": offset, "byte_size": byte_size}
        await self._post_json("v2/systemsharedmemory/region/{}/register".format(quote(name)), json.dumps(body),
                              headers, query_params, want_result=False)
        if self._verbose:
            print("Registered system shared memory with name '{}'".format(name))

    async def unregister_system_shared_memory(self, name="", headers=None, query_params=None):
        uri = "v2/systemsharedmemory/region/{}/unregister".format(quote(name)) if name != "" \
            else "v2/systemsharedmemory/unregister"
        await self._post_json(uri, "", headers, query_params, want_result=False)

    # -------------------------------------------------- device shared memory
    async def get_cuda_shared_memory_status(self, region_name="", headers=None, query_params=None):
        uri = "v2/cudasharedmemory/region/{}/status".format(quote(region_name)) if region_name != "" \
            else "v2/cudasharedmemory/status"
        return await self._get_json(uri, headers, query_params)

    async def register_cuda_shared_memory(self, name, raw_handle, device_id, byte_size, headers=None,
                                          query_params=None):
        if isinstance(raw_handle, (bytes, bytearray)):
            raw_handle = bytes(raw_handle).decode("ascii")
        body = {"raw_handle": {"b64": raw_handle}, "device_id": device_id, "byte_size": byte_size}
        await self._post_json("v2/cudasharedmemory/region/{}/register".format(quote(name)), json.dumps(body),
                              headers, query_params, want_result=False)

    async def unregister_cuda_shared_memory(self, name="", headers=None, query_params=None):
        uri = "v2/cudasharedmemory/region/{}/unregister".format(quote(name)) if name != "" \
            else "v2/cudasharedmemory/unregister"
        await self._post_json(uri, "", headers, query_params, want_result=False)

    get_hip_shared_memory_status = get_cuda_shared_memory_status
    register_hip_shared_memory = register_cuda_shared_memory
    unregister_hip_shared_memory = unregister_cuda_shared_memory

    # ------------------------------------------------------------ inference
    @staticmethod
    def generate_request_body(inputs, outputs=None, request_id="", sequence_id=0, sequence_start=False,
                              sequence_end=False, priority=0, timeout=None, parameters=None):
        return _get_inference_request(inputs=inputs, request_id=request_id, outputs=outputs,
                                      sequence_id=sequence_id, sequence_start=sequence_start,
                                      sequence_end=sequence_end, priority=priority, timeout=timeout,
                                      custom_parameters=parameters)

    @staticmethod
    def parse_response_body(response_body, verbose=False, header_length=None, content_encoding=None):
        return InferResult.from_response_body(response_body, verbose, header_length, content_encoding)

    async def infer(self, model_name, inputs, model_version="", outputs=None, request_id="", sequence_id=0,
                    sequence_start=False, sequence_end=False, priority=0, timeout=None, headers=None,
                    query_params=None, request_compression_algorithm=None, response_compression_algorithm=None,
                    parameters=None):
        """Run an inference; returns an :class:`InferResult`."""
        parts, json_size = _get_inference_request_parts(inputs, request_id, outputs, sequence_id, sequence_start,
                                                        sequence_end, priority, timeout, parameters)
        headers = dict(headers) if headers is not None else {}
        if request_compression_algorithm == "gzip":
            headers["Content-Encoding"] = "gzip"
            parts = [gzip.compress(b"".join(parts))]
        elif request_compression_algorithm == "deflate":
            headers["Content-Encoding"] = "deflate"
            parts = [zlib.compress(b"".join(parts))]
        if response_compression_algorithm in ("gzip", "deflate"):
            headers["Accept-Encoding"] = response_compression_algorithm
        if json_size is not None:
            headers["Inference-Header-Content-Length"] = json_size
        uri = self._model_uri(model_name, model_version, "/infer")
        response = await self._post(uri, parts, headers, query_params)
        await _raise_if_error(response)
        body = await response.read()
        return InferResult(_BufferedResponse(response.headers, body), self._verbose)

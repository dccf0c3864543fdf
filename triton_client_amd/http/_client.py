"""Synchronous HTTP/REST client for the KServe-v2 / Triton protocol.

Public surface parity: ``tritonclient/http/_client.py:46-1659`` (every method,
argument order and default).  Transport differences by design:

* sockets + keep-alive pool (:mod:`._transport`) instead of geventhttpclient;
* ``async_infer`` runs on a thread pool sized by ``concurrency`` and returns
  immediately (the reference yields with a 10 ms ``gevent.sleep`` per submit);
* request bodies are scatter-gather sent, responses are single buffers that
  ``InferResult.as_numpy`` views without copying.
"""

from __future__ import annotations

import base64
import gzip
import json
import ssl as _ssl
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor
from concurrent.futures import TimeoutError as _FutureTimeout
from urllib.parse import quote

from .._client import InferenceServerClientBase
from ..utils import raise_error
from ._infer_result import InferResult
from ._transport import ConnectionPool
from ._utils import _get_inference_request, _get_inference_request_parts, _get_query_string, _raise_if_error


class InferAsyncRequest:
    """Handle of an in-flight ``async_infer`` call."""

    def __init__(self, future, verbose=False):
        self._future = future
        self._verbose = verbose

    def get_result(self, block=True, timeout=None):
        """Wait for and decode the response.

        ``block=False`` raises immediately if the response has not arrived;
        ``timeout`` (seconds) bounds a blocking wait.
        """
        try:
            response = self._future.result(timeout=timeout if block else 0)
        except _FutureTimeout:
            raise_error("failed to obtain inference response")
        _raise_if_error(response)
        return InferResult(response, self._verbose)


def _split_url(url):
    hostport, slash, path = url.partition("/")
    base = ("/" + path).rstrip("/") if slash else ""
    if hostport.startswith("["):  # [v6]:port
        host, _, rest = hostport[1:].partition("]")
        port = rest[1:] if rest.startswith(":") else ""
    else:
        host, _, port = hostport.partition(":")
    return host, port, base


class InferenceServerClient(InferenceServerClientBase):
    """Client for one server reachable at ``host:port[/base-path]``.

    Not thread safe, like the reference: use one client per thread (the
    internal pool used by ``async_infer`` is managed by the client itself).

    Parameters
    ----------
    url : str
        ``host:port`` with optional base path, **without** scheme.
    verbose : bool
    concurrency : int
        Number of persistent connections (= max parallel ``async_infer``).
    connection_timeout, network_timeout : float
        Seconds for connect and for each socket operation.
    max_greenlets : int or None
        Upper bound of queued + running ``async_infer`` calls (name kept for
        API compatibility; workers are threads).
    ssl, ssl_options, ssl_context_factory, insecure
        TLS controls; ``ssl_options`` accepts ``keyfile``, ``certfile``,
        ``ca_certs``.
    """

    def __init__(
        self,
        url,
        verbose=False,
        concurrency=1,
        connection_timeout=60.0,
        network_timeout=60.0,
        max_greenlets=None,
        ssl=False,
        ssl_options=None,
        ssl_context_factory=None,
        insecure=False,
    ):
        super().__init__()
        if url.startswith("http://") or url.startswith("https://"):
            raise_error("url should not include the scheme")
        host, port, self._base_uri = _split_url(url)
        port = int(port) if port else (443 if ssl else 80)
        context = None
        if ssl:
            if ssl_context_factory is not None:
                context = ssl_context_factory()
            else:
                context = _ssl.create_default_context()
                opts = ssl_options or {}
                if opts.get("ca_certs"):
                    context.load_verify_locations(opts["ca_certs"])
                if opts.get("certfile"):
                    context.load_cert_chain(opts["certfile"], opts.get("keyfile"))
            if insecure:
                context.check_hostname = False
                context.verify_mode = _ssl.CERT_NONE
        self._pool = ConnectionPool(host, port, concurrency, connection_timeout, network_timeout, context, host)
        self._concurrency = max(1, int(concurrency))
        self._executor = None
        self._slots = threading.BoundedSemaphore(max_greenlets) if max_greenlets else None
        self._verbose = verbose
        self._closed = False

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        """Wait for outstanding async calls and drop every connection."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        if self._executor is not None:
            self._executor.shutdown(wait=True)
        self._pool.close()

    # ------------------------------------------------------------------ I/O
    def _prepare(self, request_uri, headers, query_params):
        headers = self._apply_plugin(headers)
        self._validate_headers(headers)
        target = self._base_uri + "/" + request_uri
        if query_params is not None:
            target = target + "?" + _get_query_string(query_params)
        return target, headers

    def _get(self, request_uri, headers, query_params):
        target, headers = self._prepare(request_uri, headers, query_params)
        if self._verbose:
            print("GET {}, headers {}".format(target, headers))
        response = self._pool.request("GET", target, headers)
        if self._verbose:
            print(response)
        return response

    def _post(self, request_uri, request_body, headers, query_params):
        target, headers = self._prepare(request_uri, headers, query_params)
        if self._verbose:
            print("POST {}, headers {}\n{}".format(target, headers, request_body))
        if isinstance(request_body, str):
            request_body = request_body.encode("utf-8")
        parts = request_body if isinstance(request_body, (list, tuple)) else [request_body]
        response = self._pool.request("POST", target, headers, parts)
        if self._verbose:
            print(response)
        return response

    def _validate_headers(self, headers):
        if not headers:
            return
        if any(k.lower() == "transfer-encoding" for k in headers):
            raise_error(
                "Unsupported HTTP header: 'Transfer-Encoding' is not supported in the Python client library. "
                "Use raw HTTP request libraries or the C++ client instead for this header."
            )

    @staticmethod
    def _json(response):
        content = bytes(response.read())
        return json.loads(content) if content else {}

    def _get_json(self, uri, headers, query_params):
        response = self._get(uri, headers, query_params)
        _raise_if_error(response)
        result = self._json(response)
        if self._verbose:
            print(result)
        return result

    def _post_json(self, uri, body, headers, query_params, want_result=True):
        response = self._post(uri, body, headers, query_params)
        _raise_if_error(response)
        if not want_result:
            return None
        result = self._json(response)
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
    def is_server_live(self, headers=None, query_params=None):
        """True when ``GET v2/health/live`` answers 200."""
        return self._get("v2/health/live", headers, query_params).status_code == 200

    def is_server_ready(self, headers=None, query_params=None):
        """True when ``GET v2/health/ready`` answers 200."""
        return self._get("v2/health/ready", headers, query_params).status_code == 200

    def is_model_ready(self, model_name, model_version="", headers=None, query_params=None):
        """True when the model (version) is ready for inference."""
        uri = self._model_uri(model_name, model_version, "/ready")
        return self._get(uri, headers, query_params).status_code == 200

    # ------------------------------------------------------------- metadata
    def get_server_metadata(self, headers=None, query_params=None):
        return self._get_json("v2", headers, query_params)

    def get_model_metadata(self, model_name, model_version="", headers=None, query_params=None):
        return self._get_json(self._model_uri(model_name, model_version), headers, query_params)

    def get_model_config(self, model_name, model_version="", headers=None, query_params=None):
        return self._get_json(self._model_uri(model_name, model_version, "/config"), headers, query_params)

    # ----------------------------------------------------- model repository
    def get_model_repository_index(self, headers=None, query_params=None):
        return self._post_json("v2/repository/index", "", headers, query_params)

    def load_model(self, model_name, headers=None, query_params=None, config=None, files=None):
        """Ask the server to (re)load ``model_name``.

        ``config`` is a JSON string overriding the stored configuration;
        ``files`` maps ``"file:<path>"`` to raw bytes (requires ``config``).
        """
        body = {}
        if config is not None:
            body.setdefault("parameters", {})["config"] = config
        if files is not None:
            for path, content in files.items():
                body.setdefault("parameters", {})[path] = base64.b64encode(content).decode("ascii")
        self._post_json("v2/repository/models/{}/load".format(quote(model_name)), json.dumps(body), headers,
                        query_params, want_result=False)
        if self._verbose:
            print("Loaded model '{}'".format(model_name))

    def unload_model(self, model_name, headers=None, query_params=None, unload_dependents=False):
        body = {"parameters": {"unload_dependents": unload_dependents}}
        self._post_json("v2/repository/models/{}/unload".format(quote(model_name)), json.dumps(body), headers,
                        query_params, want_result=False)
        if self._verbose:
            print("Unloaded model '{}'".format(model_name))

    # ----------------------------------------------------------- statistics
    def get_inference_statistics(self, model_name="", model_version="", headers=None, query_params=None):
        if model_name != "":
            uri = self._model_uri(model_name, model_version, "/stats")
        else:
            uri = "v2/models/stats"
        return self._get_json(uri, headers, query_params)

    # ---------------------------------------------------------------- trace
    def update_trace_settings(self, model_name=None, settings={}, headers=None, query_params=None):
        """Set trace settings (a ``None`` value clears that setting)."""
        uri = "v2/models/{}/trace/setting".format(quote(model_name)) if model_name else "v2/trace/setting"
        return self._post_json(uri, json.dumps(settings), headers, query_params)

    def get_trace_settings(self, model_name=None, headers=None, query_params=None):
        uri = "v2/models/{}/trace/setting".format(quote(model_name)) if model_name else "v2/trace/setting"
        return self._get_json(uri, headers, query_params)

    # ------------------------------------------------------------------ log
    def update_log_settings(self, settings, headers=None, query_params=None):
        return self._post_json("v2/logging", json.dumps(settings), headers, query_params)

    def get_log_settings(self, headers=None, query_params=None):
        return self._get_json("v2/logging", headers, query_params)

    # -------------------------------------------------- system shared memory
    def get_system_shared_memory_status(self, region_name="", headers=None, query_params=None):
        uri = "v2/systemsharedmemory/region/{}/status".format(quote(region_name)) if region_name != "" \
            else "v2/systemsharedmemory/status"
        return self._get_json(uri, headers, query_params)

    def register_system_shared_memory(self, name, key, byte_size, offset=0, headers=None, query_params=None):
        body = {"key": key, "offset": offset, "byte_size": byte_size}
        self._post_json("v2/systemsharedmemory/region/{}/register".format(quote(name)), json.dumps(body), headers,
                        query_params, want_result=False)
        if self._verbose:
            print("Registered system shared memory with name '{}'".format(name))

    def unregister_system_shared_memory(self, name="", headers=None, query_params=None):
        uri = "v2/systemsharedmemory/region/{}/unregister".format(quote(name)) if name != "" \
            else "v2/systemsharedmemory/unregister"
        self._post_json(uri, "", headers, query_params, want_result=False)
        if self._verbose:
            print("Unregistered system shared memory" + (" with name '{}'".format(name) if name else " regions"))

    # -------------------------------------------------- device shared memory
    # The endpoint is spelled "cudasharedmemory" by the protocol; on this stack
    # the handle is a HIP IPC handle produced by utils.hip_shared_memory.
    def get_cuda_shared_memory_status(self, region_name="", headers=None, query_params=None):
        uri = "v2/cudasharedmemory/region/{}/status".format(quote(region_name)) if region_name != "" \
            else "v2/cudasharedmemory/status"
        return self._get_json(uri, headers, query_params)

    def register_cuda_shared_memory(self, name, raw_handle, device_id, byte_size, headers=None, query_params=None):
        """Register a device region.  ``raw_handle`` is the base64 handle from
        ``hip_shared_memory.get_raw_handle``."""
        if isinstance(raw_handle, (bytes, bytearray)):
            raw_handle = bytes(raw_handle).decode("ascii")
        body = {"raw_handle": {"b64": raw_handle}, "device_id": device_id, "byte_size": byte_size}
        self._post_json("v2/cudasharedmemory/region/{}/register".format(quote(name)), json.dumps(body), headers,
                        query_params, want_result=False)
        if self._verbose:
            print("Registered device shared memory with name '{}'".format(name))

    def unregister_cuda_shared_memory(self, name="", headers=None, query_params=None):
        uri = "v2/cudasharedmemory/region/{}/unregister".format(quote(name)) if name != "" \
            else "v2/cudasharedmemory/unregister"
        self._post_json(uri, "", headers, query_params, want_result=False)
        if self._verbose:
            print("Unregistered device shared memory" + (" with name '{}'".format(name) if name else " regions"))

    # hipshm spellings of the same three calls
    get_hip_shared_memory_status = get_cuda_shared_memory_status
    register_hip_shared_memory = register_cuda_shared_memory
    unregister_hip_shared_memory = unregister_cuda_shared_memory

    # ------------------------------------------------------------ inference
    @staticmethod
    def generate_request_body(inputs, outputs=None, request_id="", sequence_id=0, sequence_start=False,
                              sequence_end=False, priority=0, timeout=None, parameters=None):
        """Build an inference request body without sending it.

        Returns ``(body_bytes, json_size)``; ``json_size`` is ``None`` when the
        body is JSON only.
        """
        return _get_inference_request(inputs=inputs, request_id=request_id, outputs=outputs,
                                      sequence_id=sequence_id, sequence_start=sequence_start,
                                      sequence_end=sequence_end, priority=priority, timeout=timeout,
                                      custom_parameters=parameters)

    @staticmethod
    def parse_response_body(response_body, verbose=False, header_length=None, content_encoding=None):
        """Decode a response body obtained out-of-band into an ``InferResult``."""
        return InferResult.from_response_body(response_body, verbose, header_length, content_encoding)

    def _infer_wire(self, model_name, inputs, model_version, outputs, request_id, sequence_id, sequence_start,
                    sequence_end, priority, timeout, headers, request_compression_algorithm,
                    response_compression_algorithm, parameters):
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
        return self._model_uri(model_name, model_version, "/infer"), parts, headers

    def infer(self, model_name, inputs, model_version="", outputs=None, request_id="", sequence_id=0,
              sequence_start=False, sequence_end=False, priority=0, timeout=None, headers=None, query_params=None,
              request_compression_algorithm=None, response_compression_algorithm=None, parameters=None):
        """Run a synchronous inference; returns an :class:`InferResult`."""
        uri, parts, headers = self._infer_wire(model_name, inputs, model_version, outputs, request_id, sequence_id,
                                               sequence_start, sequence_end, priority, timeout, headers,
                                               request_compression_algorithm, response_compression_algorithm,
                                               parameters)
        response = self._post(uri, parts, headers, query_params)
        _raise_if_error(response)
        return InferResult(response, self._verbose)

    def async_infer(self, model_name, inputs, model_version="", outputs=None, request_id="", sequence_id=0,
                    sequence_start=False, sequence_end=False, priority=0, timeout=None, headers=None,
                    query_params=None, request_compression_algorithm=None, response_compression_algorithm=None,
                    parameters=None):
        """Submit an inference and return an :class:`InferAsyncRequest` at once.

        Up to ``concurrency`` requests are on the wire simultaneously; more are
        queued inside the client.
        """
        uri, parts, headers = self._infer_wire(model_name, inputs, model_version, outputs, request_id, sequence_id,
                                               sequence_start, sequence_end, priority, timeout, headers,
                                               request_compression_algorithm, response_compression_algorithm,
                                               parameters)
        # plugin + URL work happens on the caller thread so errors surface here
        target, headers = self._prepare(uri, headers, query_params)
        if self._verbose:
            print("POST {}, headers {}".format(target, headers))
        if self._executor is None:
            self._executor = ThreadPoolExecutor(max_workers=self._concurrency, thread_name_prefix="tca-http")
        slots = self._slots
        if slots is not None:
            slots.acquire()

        def call():
            try:
                return self._pool.request("POST", target, headers, parts)
            finally:
                if slots is not None:
                    slots.release()

        return InferAsyncRequest(self._executor.submit(call), self._verbose)

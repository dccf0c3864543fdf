"""Synchronous gRPC client for the KServe-v2 / Triton protocol.

Public surface parity: ``tritonclient/grpc/_client.py:57-1936`` –
``KeepAliveOptions``, ``CallContext``, every control RPC with the
``headers / as_json / client_timeout`` tail, ``infer``, callback based
``async_infer`` and the single bidirectional stream
(``start_stream`` / ``async_stream_infer`` / ``stop_stream``).

Internals are organised around one ``_call`` helper plus the request builders
of :mod:`._requests` (shared with the asyncio client) rather than one
hand-expanded method per RPC.
"""

from __future__ import annotations

import grpc

from .._client import InferenceServerClientBase
from ..utils import raise_error
from . import _requests as rq
from . import service_pb2_grpc
from ._infer_result import InferResult
from ._infer_stream import _InferStream
from ._utils import _get_inference_request, _grpc_compression_type, get_cancelled_error, get_error_grpc, \
    raise_error_grpc

INT32_MAX = 2**31 - 1
MAX_GRPC_MESSAGE_SIZE = INT32_MAX


class KeepAliveOptions:
    """gRPC keep-alive knobs (see grpc's ``doc/keepalive.md``).

    Parameters
    ----------
    keepalive_time_ms : int
        Ping period; default ``INT32_MAX`` (effectively off).
    keepalive_timeout_ms : int
        Time to wait for a ping ack before closing; default 20000.
    keepalive_permit_without_calls : bool
        Ping even when no call is in flight; default False.
    http2_max_pings_without_data : int
        Pings allowed without data frames (0 = unlimited); default 2.
    """

    def __init__(self, keepalive_time_ms=INT32_MAX, keepalive_timeout_ms=20000,
                 keepalive_permit_without_calls=False, http2_max_pings_without_data=2):
        self.keepalive_time_ms = keepalive_time_ms
        self.keepalive_timeout_ms = keepalive_timeout_ms
        self.keepalive_permit_without_calls = keepalive_permit_without_calls
        self.http2_max_pings_without_data = http2_max_pings_without_data


class CallContext:
    """Cancellation handle returned by ``async_infer``."""

    def __init__(self, grpc_future):
        self.__grpc_future = grpc_future

    def cancel(self):
        """Cancel the underlying RPC (the callback receives a CANCELLED error)."""
        self.__grpc_future.cancel()


def _channel_options(keepalive_options, channel_args):
    # an explicit (even empty) channel_args list is used verbatim
    if channel_args is not None:
        return channel_args
    ka = keepalive_options or KeepAliveOptions()
    return [
        ("grpc.max_send_message_length", MAX_GRPC_MESSAGE_SIZE),
        ("grpc.max_receive_message_length", MAX_GRPC_MESSAGE_SIZE),
        ("grpc.keepalive_time_ms", ka.keepalive_time_ms),
        ("grpc.keepalive_timeout_ms", ka.keepalive_timeout_ms),
        ("grpc.keepalive_permit_without_calls", ka.keepalive_permit_without_calls),
        ("grpc.http2.max_pings_without_data", ka.http2_max_pings_without_data),
    ]


def _read_file(path):
    if path is None:
        return None
    with open(path, "rb") as fh:
        return fh.read()


def _make_credentials(root_certificates, private_key, certificate_chain):
    return grpc.ssl_channel_credentials(
        root_certificates=_read_file(root_certificates),
        private_key=_read_file(private_key),
        certificate_chain=_read_file(certificate_chain),
    )


class InferenceServerClient(InferenceServerClientBase):
    """gRPC client bound to ``url`` (``host:port``).

    Thread-safety matches the reference: everything except the streaming trio
    (``start_stream`` / ``async_stream_infer`` / ``stop_stream``) may be called
    from several threads.

    Parameters
    ----------
    url : str
    verbose : bool
    ssl : bool
    root_certificates, private_key, certificate_chain : str or None
        PEM file paths, used when ``ssl`` is set and ``creds`` is not.
    creds : grpc.ChannelCredentials or None
    keepalive_options : KeepAliveOptions or None
    channel_args : list of (key, value) or None
        Raw channel arguments; replaces the defaults *and* ``keepalive_options``.
    """

    def __init__(self, url, verbose=False, ssl=False, root_certificates=None, private_key=None,
                 certificate_chain=None, creds=None, keepalive_options=None, channel_args=None):
        super().__init__()
        options = _channel_options(keepalive_options, channel_args)
        if creds:
            self._channel = grpc.secure_channel(url, creds, options=options)
        elif ssl:
            self._channel = grpc.secure_channel(
                url, _make_credentials(root_certificates, private_key, certificate_chain), options=options)
        else:
            self._channel = grpc.insecure_channel(url, options=options)
        self._client_stub = service_pb2_grpc.GRPCInferenceServiceStub(self._channel)
        self._verbose = verbose
        self._stream = None

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
        """Stop the stream (if any) and close the channel."""
        if getattr(self, "_channel", None) is None:
            return
        self.stop_stream()
        self._channel.close()
        self._channel = None

    # ------------------------------------------------------------------ core
    def _get_metadata(self, headers):
        headers = self._apply_plugin(headers)
        return tuple(headers.items())

    def _call(self, built, headers, client_timeout, as_json=None, label=None):
        """Invoke a unary RPC described by ``built = (rpc_name, request)``."""
        rpc_name, request = built
        metadata = self._get_metadata(headers)
        if self._verbose:
            print("{}, metadata {}\n{}".format(label or rpc_name, metadata, request))
        try:
            response = getattr(self._client_stub, rpc_name)(request=request, metadata=metadata,
                                                            timeout=client_timeout)
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)
        if self._verbose:
            print(response)
        if as_json:
            return rq.to_json(response)
        return response

    # --------------------------------------------------------------- health
    def is_server_live(self, headers=None, client_timeout=None):
        return self._call(rq.server_live(), headers, client_timeout, label="is_server_live").live

    def is_server_ready(self, headers=None, client_timeout=None):
        return self._call(rq.server_ready(), headers, client_timeout, label="is_server_ready").ready

    def is_model_ready(self, model_name, model_version="", headers=None, client_timeout=None):
        return self._call(rq.model_ready(model_name, model_version), headers, client_timeout,
                          label="is_model_ready").ready

    # ------------------------------------------------------------- metadata
    def get_server_metadata(self, headers=None, as_json=False, client_timeout=None):
        return self._call(rq.server_metadata(), headers, client_timeout, as_json, "get_server_metadata")

    def get_model_metadata(self, model_name, model_version="", headers=None, as_json=False, client_timeout=None):
        return self._call(rq.model_metadata(model_name, model_version), headers, client_timeout, as_json,
                          "get_model_metadata")

    def get_model_config(self, model_name, model_version="", headers=None, as_json=False, client_timeout=None):
        return self._call(rq.model_config(model_name, model_version), headers, client_timeout, as_json,
                          "get_model_config")

    # ----------------------------------------------------- model repository
    def get_model_repository_index(self, headers=None, as_json=False, client_timeout=None):
        return self._call(rq.repository_index(), headers, client_timeout, as_json, "get_model_repository_index")

    def load_model(self, model_name, headers=None, config=None, files=None, client_timeout=None):
        """Ask the server to (re)load ``model_name``; ``config`` is a JSON
        string, ``files`` maps ``"file:<path>"`` to bytes."""
        verbose, self._verbose = self._verbose, False  # never dump file payloads
        try:
            self._call(rq.model_load(model_name, config, files), headers, client_timeout)
        finally:
            self._verbose = verbose
        if self._verbose:
            print("Loaded model '{}'".format(model_name))

    def unload_model(self, model_name, headers=None, unload_dependents=False, client_timeout=None):
        self._call(rq.model_unload(model_name, unload_dependents), headers, client_timeout, label="unload_model")
        if self._verbose:
            print("Unloaded model '{}'".format(model_name))

    # ----------------------------------------------------------- statistics
    def get_inference_statistics(self, model_name="", model_version="", headers=None, as_json=False,
                                 client_timeout=None):
        return self._call(rq.model_statistics(model_name, model_version), headers, client_timeout, as_json,
                          "get_inference_statistics")

    # ---------------------------------------------------------------- trace
    def update_trace_settings(self, model_name=None, settings={}, headers=None, as_json=False,
                              client_timeout=None):
        return self._call(rq.trace_setting(model_name, settings), headers, client_timeout, as_json,
                          "update_trace_settings")

    def get_trace_settings(self, model_name=None, headers=None, as_json=False, client_timeout=None):
        return self._call(rq.trace_setting(model_name, None), headers, client_timeout, as_json,
                          "get_trace_settings")

    # ------------------------------------------------------------------ log
    def update_log_settings(self, settings, headers=None, as_json=False, client_timeout=None):
        return self._call(rq.log_settings(settings), headers, client_timeout, as_json, "update_log_settings")

    def get_log_settings(self, headers=None, as_json=False, client_timeout=None):
        return self._call(rq.log_settings(None), headers, client_timeout, as_json, "get_log_settings")

    # -------------------------------------------------- system shared memory
    def get_system_shared_memory_status(self, region_name="", headers=None, as_json=False, client_timeout=None):
        return self._call(rq.sysshm_status(region_name), headers, client_timeout, as_json,
                          "get_system_shared_memory_status")

    def register_system_shared_memory(self, name, key, byte_size, offset=0, headers=None, client_timeout=None):
        self._call(rq.sysshm_register(name, key, byte_size, offset), headers, client_timeout,
                   label="register_system_shared_memory")
        if self._verbose:
            print("Registered system shared memory with name '{}'".format(name))

    def unregister_system_shared_memory(self, name="", headers=None, client_timeout=None):
        self._call(rq.sysshm_unregister(name), headers, client_timeout, label="unregister_system_shared_memory")
        if self._verbose:
            print("Unregistered system shared memory" + (" with name '{}'".format(name) if name else " regions"))

    # -------------------------------------------------- device shared memory
    def get_cuda_shared_memory_status(self, region_name="", headers=None, as_json=False, client_timeout=None):
        return self._call(rq.devshm_status(region_name), headers, client_timeout, as_json,
                          "get_cuda_shared_memory_status")

    def register_cuda_shared_memory(self, name, raw_handle, device_id, byte_size, headers=None,
                                    client_timeout=None):
        """Register a device region; ``raw_handle`` is the base64 HIP IPC handle
        from ``hip_shared_memory.get_raw_handle``."""
        self._call(rq.devshm_register(name, raw_handle, device_id, byte_size), headers, client_timeout,
                   label="register_cuda_shared_memory")
        if self._verbose:
            print("Registered device shared memory with name '{}'".format(name))

    def unregister_cuda_shared_memory(self, name="", headers=None, client_timeout=None):
        self._call(rq.devshm_unregister(name), headers, client_timeout, label="unregister_cuda_shared_memory")
        if self._verbose:
            print("Unregistered device shared memory" + (" with name '{}'".format(name) if name else " regions"))

    get_hip_shared_memory_status = get_cuda_shared_memory_status
    register_hip_shared_memory = register_cuda_shared_memory
    unregister_hip_shared_memory = unregister_cuda_shared_memory

    # ------------------------------------------------------------ inference
    def infer(self, model_name, inputs, model_version="", outputs=None, request_id="", sequence_id=0,
              sequence_start=False, sequence_end=False, priority=0, timeout=None, client_timeout=None,
              headers=None, compression_algorithm=None, parameters=None):
        """Run a synchronous inference; returns an :class:`InferResult`."""
        metadata = self._get_metadata(headers)
        rq.check_version(model_version)
        request = _get_inference_request(model_name, inputs, model_version, request_id, outputs, sequence_id,
                                         sequence_start, sequence_end, priority, timeout, parameters)
        if self._verbose:
            print("infer, metadata {}\n{}".format(metadata, request))
        try:
            response = self._client_stub.ModelInfer(request=request, metadata=metadata, timeout=client_timeout,
                                                    compression=_grpc_compression_type(compression_algorithm))
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)
        if self._verbose:
            print(response)
        return InferResult(response)

    def async_infer(self, model_name, inputs, callback, model_version="", outputs=None, request_id="",
                    sequence_id=0, sequence_start=False, sequence_end=False, priority=0, timeout=None,
                    client_timeout=None, headers=None, compression_algorithm=None, parameters=None):
        """Submit an inference; ``callback(result=, error=)`` fires on completion.

        Returns a :class:`CallContext` that can cancel the request.
        """
        verbose = self._verbose

        def on_done(call_future):
            error = result = None
            try:
                result = InferResult(call_future.result())
                if verbose:
                    print(result.get_response())
            except grpc.RpcError as rpc_error:
                error = get_error_grpc(rpc_error)
            except grpc.FutureCancelledError:
                error = get_cancelled_error()
            callback(result=result, error=error)

        metadata = self._get_metadata(headers)
        rq.check_version(model_version)
        request = _get_inference_request(model_name, inputs, model_version, request_id, outputs, sequence_id,
                                         sequence_start, sequence_end, priority, timeout, parameters)
        if self._verbose:
            print("async_infer, metadata {}\n{}".format(metadata, request))
        try:
            future = self._client_stub.ModelInfer.future(
                request=request, metadata=metadata, timeout=client_timeout,
                compression=_grpc_compression_type(compression_algorithm))
            future.add_done_callback(on_done)
            if verbose:
                print("Sent request")
            return CallContext(future)
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)

    # ------------------------------------------------------------ streaming
    def start_stream(self, callback, stream_timeout=None, headers=None, compression_algorithm=None):
        """Open the client's single ``ModelStreamInfer`` stream.

        ``callback(result=, error=)`` is invoked (on a reader thread) once per
        response.  ``stream_timeout`` (seconds) bounds the stream's lifetime.
        """
        if self._stream is not None:
            raise_error("cannot start another stream with one already running. "
                        "'InferenceServerClient' supports only a single active stream at a given time.")
        metadata = self._get_metadata(headers)
        self._stream = _InferStream(callback, self._verbose)
        try:
            responses = self._client_stub.ModelStreamInfer(
                self._stream.request_iterator(), metadata=metadata, timeout=stream_timeout,
                compression=_grpc_compression_type(compression_algorithm))
            self._stream.start(responses)
        except grpc.RpcError as rpc_error:
            self._stream = None
            raise_error_grpc(rpc_error)

    def stop_stream(self, cancel_requests=False):
        """Close the stream.  ``cancel_requests=True`` cancels what is still in
        flight instead of waiting for it."""
        if getattr(self, "_stream", None) is not None:
            self._stream.close(cancel_requests)
        self._stream = None

    def async_stream_infer(self, model_name, inputs, model_version="", outputs=None, request_id="",
                           sequence_id=0, sequence_start=False, sequence_end=False,
                           enable_empty_final_response=False, priority=0, timeout=None, parameters=None):
        """Send one request over the active stream (responses reach the
        ``start_stream`` callback)."""
        if self._stream is None:
            raise_error("stream not available, use start_stream() to make one available.")
        rq.check_version(model_version)
        request = _get_inference_request(model_name, inputs, model_version, request_id, outputs, sequence_id,
                                         sequence_start, sequence_end, priority, timeout, parameters)
        # only meaningful on streams: ask decoupled models for an explicit
        # (possibly empty) final response flagged triton_final_response
        if enable_empty_final_response:
            request.parameters["triton_enable_empty_final_response"].bool_param = True
        if self._verbose:
            print("async_stream_infer\n{}".format(request))
        self._stream.enqueue(request)
        if self._verbose:
            print("enqueued request {} to stream...".format(request_id))

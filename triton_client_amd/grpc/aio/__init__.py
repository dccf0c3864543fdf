"""asyncio gRPC client on ``grpc.aio``.

Public surface parity: ``tritonclient/grpc/aio/__init__.py:50-810`` – the
synchronous client's methods as coroutines plus ``stream_infer`` (an async
iterator of ``(InferResult, error)`` with ``.cancel()``).  Request construction
is shared with the synchronous client through :mod:`.._requests`.
"""

from __future__ import annotations

import grpc

from ..._client import InferenceServerClientBase, InferenceServerClientPlugin  # noqa: F401
from ..._request import Request  # noqa: F401
from ...utils import *  # noqa: F401,F403
from ...utils import InferenceServerException, raise_error
from .. import _requests as rq
from .. import model_config_pb2, service_pb2, service_pb2_grpc  # noqa: F401
from .._client import MAX_GRPC_MESSAGE_SIZE, KeepAliveOptions, _channel_options, _make_credentials  # noqa: F401
from .._infer_input import InferInput  # noqa: F401
from .._infer_result import InferResult
from .._requested_output import InferRequestedOutput  # noqa: F401
from .._utils import _get_inference_request, _grpc_compression_type, raise_error_grpc

__all__ = [
    "InferenceServerClientPlugin",
    "Request",
    "InferenceServerClient",
    "InferInput",
    "InferRequestedOutput",
    "InferResult",
    "KeepAliveOptions",
    "InferenceServerException",
]

_STREAM_DEFAULTS = {
    "model_version": "",
    "outputs": None,
    "request_id": "",
    "sequence_id": 0,
    "sequence_start": False,
    "sequence_end": False,
    "priority": 0,
    "timeout": None,
    "parameters": None,
}


class _ResponseIterator:
    """Async iterator over a ``ModelStreamInfer`` call; ``cancel()`` aborts it."""

    def __init__(self, grpc_call, verbose):
        self._grpc_call = grpc_call
        self._it = grpc_call.__aiter__()
        self._verbose = verbose

    def __aiter__(self):
        return self

    async def __anext__(self):
        try:
            response = await self._it.__anext__()
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)
        if self._verbose:
            print(response)
        if response.error_message != "":
            return None, InferenceServerException(msg=response.error_message)
        return InferResult(response.infer_response), None

    def cancel(self):
        return self._grpc_call.cancel()


class InferenceServerClient(InferenceServerClientBase):
    """asyncio gRPC client; constructor arguments as the synchronous client."""

    def __init__(self, url, verbose=False, ssl=False, root_certificates=None, private_key=None,
                 certificate_chain=None, creds=None, keepalive_options=None, channel_args=None):
        super().__init__()
        options = _channel_options(keepalive_options, channel_args)
        if creds:
            self._channel = grpc.aio.secure_channel(url, creds, options=options)
        elif ssl:
            self._channel = grpc.aio.secure_channel(
                url, _make_credentials(root_certificates, private_key, certificate_chain), options=options)
        else:
            self._channel = grpc.aio.insecure_channel(url, options=options)
        self._client_stub = service_pb2_grpc.GRPCInferenceServiceStub(self._channel)
        self._verbose = verbose

    async def __aenter__(self):
        return self

    async def __aexit__(self, exc_type, exc, tb):
        await self.close()

    async def close(self):
        await self._channel.close()

    def _get_metadata(self, headers):
        headers = self._apply_plugin(headers)
        return tuple(headers.items())

    async def _call(self, built, headers, client_timeout, as_json=None, label=None):
        rpc_name, request = built
        metadata = self._get_metadata(headers)
        if self._verbose:
            print("{}, metadata {}\n{}".format(label or rpc_name, metadata, request))
        try:
            response = await getattr(self._client_stub, rpc_name)(request=request, metadata=metadata,
                                                                  timeout=client_timeout)
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)
        if self._verbose:
            print(response)
        if as_json:
            return rq.to_json(response)
        return response

    # --------------------------------------------------------------- health
    async def is_server_live(self, headers=None, client_timeout=None):
        return (await self._call(rq.server_live(), headers, client_timeout, label="is_server_live")).live

    async def is_server_ready(self, headers=None, client_timeout=None):
        return (await self._call(rq.server_ready(), headers, client_timeout, label="is_server_ready")).ready

    async def is_model_ready(self, model_name, model_version="", headers=None, client_timeout=None):
        return (await self._call(rq.model_ready(model_name, model_version), headers, client_timeout,
                                 label="is_model_ready")).ready

    # ------------------------------------------------------------- metadata
    async def get_server_metadata(self, headers=None, as_json=False, client_timeout=None):
        return await self._call(rq.server_metadata(), headers, client_timeout, as_json, "get_server_metadata")

    async def get_model_metadata(self, model_name, model_version="", headers=None, as_json=False,
                                 client_timeout=None):
        return await self._call(rq.model_metadata(model_name, model_version), headers, client_timeout, as_json,
                                "get_model_metadata")

    async def get_model_config(self, model_name, model_version="", headers=None, as_json=False,
                               client_timeout=None):
        return await self._call(rq.model_config(model_name, model_version), headers, client_timeout, as_json,
                                "get_model_config")

    # ----------------------------------------------------- model repository
    async def get_model_repository_index(self, headers=None, as_json=False, client_timeout=None):
        return await self._call(rq.repository_index(), headers, client_timeout, as_json,
                                "get_model_repository_index")

    async def load_model(self, model_name, headers=None, config=None, files=None, client_timeout=None):
        verbose, self._verbose = self._verbose, False
        try:
            await self._call(rq.model_load(model_name, config, files), headers, client_timeout)
        finally:
            self._verbose = verbose
        if self._verbose:
            print("Loaded model '{}'".format(model_name))

    async def unload_model(self, model_name, headers=None, unload_dependents=False, client_timeout=None):
        await self._call(rq.model_unload(model_name, unload_dependents), headers, client_timeout,
                         label="unload_model")
        if self._verbose:
            print("Unloaded model '{}'".format(model_name))

    # ----------------------------------------------------------- statistics
    async def get_inference_statistics(self, model_name="", model_version="", headers=None, as_json=False,
                                       client_timeout=None):
        return await self._call(rq.model_statistics(model_name, model_version), headers, client_timeout, as_json,
                                "get_inference_statistics")

    # ---------------------------------------------------------------- trace
    async def update_trace_settings(self, model_name=None, settings={}, headers=None, as_json=False,
                                    client_timeout=None):
        return await self._call(rq.trace_setting(model_name, settings), headers, client_timeout, as_json,
                                "update_trace_settings")

    async def get_trace_settings(self, model_name=None, headers=None, as_json=False, client_timeout=None):
        return await self._call(rq.trace_setting(model_name, None), headers, client_timeout, as_json,
                                "get_trace_settings")

    # ------------------------------------------------------------------ log
    async def update_log_settings(self, settings, headers=None, as_json=False, client_timeout=None):
        return await self._call(rq.log_settings(settings), headers, client_timeout, as_json, "update_log_settings")

    async def get_log_settings(self, headers=None, as_json=False, client_timeout=None):
        return await self._call(rq.log_settings(None), headers, client_timeout, as_json, "get_log_settings")

    # -------------------------------------------------- system shared memory
    async def get_system_shared_memory_status(self, region_name="", headers=None, as_json=False,
                                              client_timeout=None):
        return await self._call(rq.sysshm_status(region_name), headers, client_timeout, as_json,
                                "get_system_shared_memory_status")

    async def register_system_shared_memory(self, name, key, byte_size, offset=0, headers=None,
                                            client_timeout=None):
        await self._call(rq.sysshm_register(name, key, byte_size, offset), headers, client_timeout,
                         label="register_system_shared_memory")

    async def unregister_system_shared_memory(self, name="", headers=None, client_timeout=None):
        await self._call(rq.sysshm_unregister(name), headers, client_timeout,
                         label="unregister_system_shared_memory")

    # -------------------------------------------------- device shared memory
    async def get_cuda_shared_memory_status(self, region_name="", headers=None, as_json=False,
                                            client_timeout=None):
        return await self._call(rq.devshm_status(region_name), headers, client_timeout, as_json,
                                "get_cuda_shared_memory_status")

    async def register_cuda_shared_memory(self, name, raw_handle, device_id, byte_size, headers=None,
                                          client_timeout=None):
        await self._call(rq.devshm_register(name, raw_handle, device_id, byte_size), headers, client_timeout,
                         label="register_cuda_shared_memory")

    async def unregister_cuda_shared_memory(self, name="", headers=None, client_timeout=None):
        await self._call(rq.devshm_unregister(name), headers, client_timeout,
                         label="unregister_cuda_shared_memory")

    get_hip_shared_memory_status = get_cuda_shared_memory_status
    register_hip_shared_memory = register_cuda_shared_memory
    unregister_hip_shared_memory = unregister_cuda_shared_memory

    # ------------------------------------------------------------ inference
    async def infer(self, model_name, inputs, model_version="", outputs=None, request_id="", sequence_id=0,
                    sequence_start=False, sequence_end=False, priority=0, timeout=None, client_timeout=None,
                    headers=None, compression_algorithm=None, parameters=None):
        metadata = self._get_metadata(headers)
        rq.check_version(model_version)
        request = _get_inference_request(model_name, inputs, model_version, request_id, outputs, sequence_id,
                                         sequence_start, sequence_end, priority, timeout, parameters)
        if self._verbose:
            print("infer, metadata {}\n{}".format(metadata, request))
        try:
            response = await self._client_stub.ModelInfer(
                request=request, metadata=metadata, timeout=client_timeout,
                compression=_grpc_compression_type(compression_algorithm))
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)
        if self._verbose:
            print(response)
        return InferResult(response)

    def stream_infer(self, inputs_iterator, stream_timeout=None, headers=None, compression_algorithm=None):
        """Bidirectional streaming inference.

        ``inputs_iterator`` is an async iterator yielding dicts with the keyword
        arguments of the synchronous ``async_stream_infer`` (``model_name`` and
        ``inputs`` mandatory).  Returns an async iterator of
        ``(InferResult | None, InferenceServerException | None)`` that also
        offers ``cancel()``.
        """
        metadata = self._get_metadata(headers)

        async def requests():
            async for spec in inputs_iterator:
                if type(spec) != dict:
                    raise_error("inputs_iterator is not yielding a dict")
                if "model_name" not in spec or "inputs" not in spec:
                    raise_error("model_name and/or inputs is missing from inputs_iterator's yielded dict")
                args = dict(_STREAM_DEFAULTS)
                args.update(spec)
                if type(args["model_version"]) != str:
                    raise_error("model_version must be a string")
                final = args.pop("enable_empty_final_response", False)
                request = _get_inference_request(
                    model_name=args["model_name"], inputs=args["inputs"], model_version=args["model_version"],
                    request_id=args["request_id"], outputs=args["outputs"], sequence_id=args["sequence_id"],
                    sequence_start=args["sequence_start"], sequence_end=args["sequence_end"],
                    priority=args["priority"], timeout=args["timeout"], parameters=args["parameters"])
                if final:
                    request.parameters["triton_enable_empty_final_response"].bool_param = True
                yield request

        try:
            call = self._client_stub.ModelStreamInfer(
                requests(), metadata=metadata, timeout=stream_timeout,
                compression=_grpc_compression_type(compression_algorithm))
            return _ResponseIterator(call, self._verbose)
        except grpc.RpcError as rpc_error:
            raise_error_grpc(rpc_error)

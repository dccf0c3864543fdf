"""gRPC front of the in-repo v2 server (``inference.GRPCInferenceService``).

Every RPC the clients call is served (reference client side:
``tritonclient/grpc/_client.py:298-1790``), including the bidirectional
``ModelStreamInfer`` used for sequences and decoupled models.
"""

from __future__ import annotations

import queue
import threading
from concurrent import futures

import grpc

from ..grpc import service_pb2 as pb
from ..grpc import service_pb2_grpc
from .core import InferRequest, InputTensor, RequestedOutput, ServerError, ShmRef, encode_wire

import numpy as np
from google.protobuf import json_format

_CODE = {400: grpc.StatusCode.INVALID_ARGUMENT, 404: grpc.StatusCode.NOT_FOUND, 503: grpc.StatusCode.UNAVAILABLE,
         500: grpc.StatusCode.INTERNAL}

_TYPED = {
    "BOOL": "bool_contents", "INT8": "int_contents", "INT16": "int_contents", "INT32": "int_contents",
    "INT64": "int64_contents", "UINT8": "uint_contents", "UINT16": "uint_contents", "UINT32": "uint_contents",
    "UINT64": "uint64_contents", "FP32": "fp32_contents", "FP64": "fp64_contents", "BYTES": "bytes_contents",
}

_NP = {"BOOL": np.bool_, "INT8": np.int8, "INT16": np.int16, "INT32": np.int32, "INT64": np.int64,
       "UINT8": np.uint8, "UINT16": np.uint16, "UINT32": np.uint32, "UINT64": np.uint64, "FP32": np.float32,
       "FP64": np.float64}


def _param_value(p):
    kind = p.WhichOneof("parameter_choice")
    return getattr(p, kind) if kind else None


def _shm_ref(params):
    """``ShmRef`` out of a tensor's parameter map, or None when it names no region."""
    region = params.get("shared_memory_region")
    if region is None:
        return None
    size = params.get("shared_memory_byte_size")
    offset = params.get("shared_memory_offset")
    return ShmRef(_param_value(region), int(_param_value(size) or 0) if size is not None else 0,
                  int(_param_value(offset) or 0) if offset is not None else 0)


def request_from_proto(msg, streaming=False, raw=None):
    """``ModelInferRequest`` -> :class:`InferRequest`.  ``raw`` replaces the
    message's ``raw_input_contents`` (buffers a transport kept out of the parsed
    message to avoid copying them)."""
    top = msg.parameters
    req = InferRequest(model_name=msg.model_name, model_version=msg.model_version, id=msg.id, streaming=streaming,
                       parameters={k: _param_value(v) for k, v in top.items()} if len(top) else {})
    raw_contents = msg.raw_input_contents if raw is None else raw
    raw_iter = iter(raw_contents)
    use_raw = len(raw_contents) > 0
    inputs = req.inputs
    for t in msg.inputs:
        it = InputTensor(name=t.name, datatype=t.datatype, shape=list(t.shape))
        params = t.parameters
        shm = _shm_ref(params) if len(params) else None
        if shm is not None:
            it.shm = shm
            inputs.append(it)
            continue
        has_contents = t.HasField("contents") and t.contents.ByteSize() > 0
        if use_raw:
            if has_contents:
                raise ServerError(
                    f"contents field must not be specified when using raw_input_contents for '{t.name}' for model "
                    f"'{msg.model_name}'")
            try:
                it.raw = next(raw_iter)
                if t.datatype == "BYTES" and not isinstance(it.raw, bytes):
                    it.raw = bytes(it.raw)  # the length-prefix decoder works on bytes
            except StopIteration:
                raise ServerError(
                    f"expected raw_input_contents for every non shared-memory input of model '{msg.model_name}'"
                ) from None
        elif has_contents:
            field = _TYPED.get(t.datatype)
            if field is None:
                raise ServerError(f"datatype '{t.datatype}' must be sent through raw_input_contents")
            values = getattr(t.contents, field)
            if t.datatype == "BYTES":
                arr = np.empty(len(values), dtype=np.object_)
                for i, v in enumerate(values):
                    arr[i] = v
            else:
                arr = np.array(values, dtype=_NP[t.datatype])
            try:
                it.data = arr.reshape(list(t.shape))
            except ValueError:
                raise ServerError(
                    f"unexpected number of elements {arr.size} in contents for input '{t.name}'") from None
        else:
            raise ServerError(f"input '{t.name}' for model '{msg.model_name}' has no data")
        inputs.append(it)
    if len(msg.outputs):
        req.outputs = []
        for o in msg.outputs:
            params = o.parameters
            if len(params):
                k = params.get("classification")
                ro = RequestedOutput(name=o.name, class_count=int(_param_value(k) or 0) if k is not None else 0)
                ro.shm = _shm_ref(params)
            else:
                ro = RequestedOutput(name=o.name)
            req.outputs.append(ro)
    return req


def mark_final(req, response):
    """A one-response model on a stream: its response is the final one, and says so when the
    request asked for the final-response marker (``triton_enable_empty_final_response``)."""
    if req.parameters.get("triton_enable_empty_final_response"):
        response.parameters["triton_final_response"] = True


def _set_param(p, value):
    if isinstance(value, bool):
        p.bool_param = value
    elif isinstance(value, int):
        p.int64_param = value
    elif isinstance(value, float):
        p.double_param = value
    else:
        p.string_param = str(value)


def response_to_proto(resp):
    """:class:`InferResponse` -> ``ModelInferResponse``."""
    msg = pb.ModelInferResponse(model_name=resp.model_name, model_version=resp.model_version, id=resp.id)
    for k, v in resp.parameters.items():
        _set_param(msg.parameters[k], v)
    for out in resp.outputs:
        o = msg.outputs.add(name=out.name, datatype=out.datatype, shape=[int(s) for s in out.shape])
        if out.in_shm is not None:
            o.parameters["shared_memory_region"].string_param = out.in_shm.region
            o.parameters["shared_memory_byte_size"].int64_param = out.in_shm.byte_size
            if out.in_shm.offset:
                o.parameters["shared_memory_offset"].int64_param = out.in_shm.offset
        else:
            msg.raw_output_contents.append(encode_wire(out.datatype, out.data))
    return msg


def _abort(context, err: ServerError):
    context.abort(_CODE.get(err.code, grpc.StatusCode.INTERNAL), err.message)


class Servicer(service_pb2_grpc.GRPCInferenceServiceServicer):
    def __init__(self, core, stream_workers=16):
        self.core = core
        self._stream_pool = futures.ThreadPoolExecutor(max_workers=stream_workers,
                                                       thread_name_prefix="tca-grpc-stream")

    # -- health / metadata --
    def ServerLive(self, request, context):
        return pb.ServerLiveResponse(live=self.core.live)

    def ServerReady(self, request, context):
        return pb.ServerReadyResponse(ready=self.core.ready)

    def ModelReady(self, request, context):
        return pb.ModelReadyResponse(ready=self.core.is_model_ready(request.name, request.version))

    def ServerMetadata(self, request, context):
        md = self.core.server_metadata()
        return pb.ServerMetadataResponse(name=md["name"], version=md["version"], extensions=md["extensions"])

    def ModelMetadata(self, request, context):
        try:
            md = self.core.model_metadata(request.name, request.version)
        except ServerError as e:
            _abort(context, e)
        resp = pb.ModelMetadataResponse(name=md["name"], versions=md["versions"], platform=md["platform"])
        for t in md["inputs"]:
            resp.inputs.add(name=t["name"], datatype=t["datatype"], shape=t["shape"])
        for t in md["outputs"]:
            resp.outputs.add(name=t["name"], datatype=t["datatype"], shape=t["shape"])
        return resp

    def ModelConfig(self, request, context):
        try:
            cfg = self.core.model_config(request.name, request.version)
        except ServerError as e:
            _abort(context, e)
        resp = pb.ModelConfigResponse()
        json_format.ParseDict(cfg, resp.config, ignore_unknown_fields=True)
        return resp

    def ModelStatistics(self, request, context):
        try:
            stats = self.core.statistics(request.name, request.version)
        except ServerError as e:
            _abort(context, e)
        resp = pb.ModelStatisticsResponse()
        json_format.ParseDict(stats, resp, ignore_unknown_fields=True)
        return resp

    # -- repository --
    def RepositoryIndex(self, request, context):
        resp = pb.RepositoryIndexResponse()
        for m in self.core.repository_index(request.ready):
            resp.models.add(**m)
        return resp

    def RepositoryModelLoad(self, request, context):
        config = None
        files = {}
        for key, p in request.parameters.items():
            kind = p.WhichOneof("parameter_choice")
            if key == "config" and kind == "string_param":
                config = p.string_param
            elif key.startswith("file:") and kind == "bytes_param":
                files[key] = p.bytes_param
        try:
            self.core.load_model(request.model_name, config, files or None)
        except ServerError as e:
            _abort(context, e)
        return pb.RepositoryModelLoadResponse()

    def RepositoryModelUnload(self, request, context):
        dependents = False
        if "unload_dependents" in request.parameters:
            dependents = request.parameters["unload_dependents"].bool_param
        try:
            self.core.unload_model(request.model_name, dependents)
        except ServerError as e:
            _abort(context, e)
        return pb.RepositoryModelUnloadResponse()

    # -- shared memory --
    def SystemSharedMemoryStatus(self, request, context):
        try:
            regions = self.core.sysshm.status(request.name)
        except ServerError as e:
            _abort(context, e)
        resp = pb.SystemSharedMemoryStatusResponse()
        for r in regions:
            entry = resp.regions[r["name"]]
            entry.name, entry.key, entry.offset, entry.byte_size = r["name"], r["key"], r["offset"], r["byte_size"]
        return resp

    def SystemSharedMemoryRegister(self, request, context):
        try:
            self.core.sysshm.register(request.name, request.key, request.offset, request.byte_size)
        except ServerError as e:
            _abort(context, e)
        return pb.SystemSharedMemoryRegisterResponse()

    def SystemSharedMemoryUnregister(self, request, context):
        self.core.sysshm.unregister(request.name)
        return pb.SystemSharedMemoryUnregisterResponse()

    def CudaSharedMemoryStatus(self, request, context):
        try:
            regions = self.core.devshm.status(request.name)
        except ServerError as e:
            _abort(context, e)
        resp = pb.CudaSharedMemoryStatusResponse()
        for r in regions:
            entry = resp.regions[r["name"]]
            entry.name, entry.device_id, entry.byte_size = r["name"], r["device_id"], r["byte_size"]
        return resp

    def CudaSharedMemoryRegister(self, request, context):
        try:
            self.core.devshm.register(request.name, request.raw_handle, request.device_id, request.byte_size)
        except ServerError as e:
            _abort(context, e)
        return pb.CudaSharedMemoryRegisterResponse()

    def CudaSharedMemoryUnregister(self, request, context):
        self.core.devshm.unregister(request.name)
        return pb.CudaSharedMemoryUnregisterResponse()

    # -- trace / log --
    def TraceSetting(self, request, context):
        try:
            if len(request.settings):
                settings = {k: (list(v.value) if len(v.value) else None) for k, v in request.settings.items()}
                merged = self.core.update_trace(request.model_name or None, settings)
            else:
                merged = self.core.get_trace(request.model_name or None)
        except ServerError as e:
            _abort(context, e)
        resp = pb.TraceSettingResponse()
        for k, v in merged.items():
            resp.settings[k].value.extend(v)
        return resp

    def LogSettings(self, request, context):
        try:
            settings = {}
            for k, v in request.settings.items():
                kind = v.WhichOneof("parameter_choice")
                settings[k] = getattr(v, kind) if kind else None
            merged = self.core.update_log(settings) if settings else dict(self.core.log_settings)
        except ServerError as e:
            _abort(context, e)
        resp = pb.LogSettingsResponse()
        for k, v in merged.items():
            if isinstance(v, bool):
                resp.settings[k].bool_param = v
            elif isinstance(v, int):
                resp.settings[k].uint32_param = v
            else:
                resp.settings[k].string_param = str(v)
        return resp

    # -- inference --
    def ModelInfer(self, request, context):
        try:
            resp = self.core.infer(request_from_proto(request))
            if not hasattr(resp, "outputs"):
                raise ServerError("ModelInfer RPC doesn't support models with decoupled transaction policy", 400)
            return response_to_proto(resp)
        except ServerError as e:
            _abort(context, e)

    def ModelStreamInfer(self, request_iterator, context):
        """Requests are executed concurrently (one worker each, so interleaved
        sequences and decoupled models make progress independently) while
        responses are written from this generator in completion order."""
        out = queue.Queue()
        pending = [0]
        lock = threading.Lock()
        done_reading = threading.Event()
        _EOS = object()

        def run(msg):
            try:
                try:
                    req = request_from_proto(msg, streaming=True)
                    result = self.core.infer(req)
                    if hasattr(result, "outputs"):
                        mark_final(req, result)
                        out.put(pb.ModelStreamInferResponse(infer_response=response_to_proto(result)))
                    else:
                        for item in result:
                            out.put(pb.ModelStreamInferResponse(infer_response=response_to_proto(item)))
                except ServerError as e:
                    err = pb.ModelStreamInferResponse(error_message=e.message)
                    err.infer_response.id = msg.id
                    err.infer_response.model_name = msg.model_name
                    out.put(err)
                except Exception as e:  # model bug: report, keep the stream alive
                    out.put(pb.ModelStreamInferResponse(error_message=f"internal error: {type(e).__name__}: {e}"))
            finally:
                with lock:
                    pending[0] -= 1
                    last = pending[0] == 0 and done_reading.is_set()
                if last:
                    out.put(_EOS)

        def reader():
            # sequence requests must execute in arrival order per sequence id:
            # chain them through a per-sequence lock-step by running those inline
            try:
                for msg in request_iterator:
                    with lock:
                        pending[0] += 1
                    if "sequence_id" in msg.parameters:
                        run(msg)
                    else:
                        self._stream_pool.submit(run, msg)
            except Exception:
                pass
            finally:
                done_reading.set()
                with lock:
                    idle = pending[0] == 0
                if idle:
                    out.put(_EOS)

        threading.Thread(target=reader, name="tca-grpc-stream-reader", daemon=True).start()
        written = {}  # model name -> responses written on this stream (fault injection only)
        while True:
            item = out.get()
            if item is _EOS:
                return
            yield item
            name = item.infer_response.model_name
            limit = self.core.stream_drop_after(name) if name else 0
            if limit:
                written[name] = written.get(name, 0) + 1
                if written[name] >= limit:
                    context.abort(grpc.StatusCode.UNAVAILABLE,
                                  f"injected stream failure for model '{name}' (drop_stream_after_n={limit})")


class GrpcFront:
    def __init__(self, core, host="127.0.0.1", port=8001, workers=32, tls=None):
        self.core = core
        self.host = host
        self.port = port
        self._tls = tls  # (certificate chain, private key, client CA or None) as PEM file paths
        self.server = grpc.server(
            futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tca-grpc-srv"),
            options=[("grpc.max_send_message_length", 2**31 - 1), ("grpc.max_receive_message_length", 2**31 - 1),
                     ("grpc.so_reuseport", 0)],
        )
        service_pb2_grpc.add_GRPCInferenceServiceServicer_to_server(Servicer(core), self.server)

    def start(self):
        if self._tls:
            cert, key, client_ca = self._tls
            with open(key, "rb") as kf, open(cert, "rb") as cf:
                pair = (kf.read(), cf.read())
            roots = None
            if client_ca:
                with open(client_ca, "rb") as rf:
                    roots = rf.read()
            creds = grpc.ssl_server_credentials([pair], root_certificates=roots, require_client_auth=bool(client_ca))
            self.port = self.server.add_secure_port(f"{self.host}:{self.port}", creds)
        else:
            self.port = self.server.add_insecure_port(f"{self.host}:{self.port}")
        if self.port == 0:
            raise RuntimeError("could not bind the gRPC port")
        self.server.start()
        return self.port

    def stop(self, grace=0.5):
        self.server.stop(grace)

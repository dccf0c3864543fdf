"""Transport independent heart of the in-repo KServe-v2 server.

The reference tree has no server at all (every example needs a live Triton,
``src/c++/tests/cc_client_test.cc:39-40``); this one exists so the client
stack, the examples and the load generator run hermetically, and so that one
PyTorch-ROCm model instance per MI355X can be driven through the real wire
protocol.

``InferenceCore`` owns

* the model repository (zoo -> loaded instances, load/unload/index, config
  override),
* the shared-memory registries (system shm by key, device shm by HIP IPC
  handle),
* per-model statistics, trace and log settings,
* sequence state, the dynamic batcher for GPU models and fault injection knobs
  (``delay_ms`` / ``fail_every_n`` / ``drop_stream_after_n`` model parameters).

The HTTP and gRPC fronts translate their wire format to :class:`InferRequest`
and back from :class:`InferResponse`.
"""

from __future__ import annotations

import base64
import contextlib
import copy
import json
import math
import os
import threading
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from ..models import MODEL_ZOO, ExecContext, ModelError, SequenceInfo
from ..utils import (
    deserialize_bf16_tensor,
    deserialize_bytes_tensor,
    serialize_bf16_tensor,
    serialize_byte_tensor,
    triton_to_np_dtype,
)
from ..utils import shared_memory as sysshm
from ..utils import trace as _trace

SERVER_NAME = "triton"
SERVER_VERSION = "2.50.0-mi355x"
EXTENSIONS = [
    "classification", "sequence", "model_repository", "model_repository(unload_dependents)", "schedule_policy",
    "model_configuration", "system_shared_memory", "cuda_shared_memory", "binary_tensor_data", "parameters",
    "statistics", "trace", "logging",
]


class ServerError(Exception):
    """An error to report to the client.  ``code`` is the HTTP status; the gRPC
    front maps 400->INVALID_ARGUMENT, 404->NOT_FOUND, 503->UNAVAILABLE, others
    -> INTERNAL."""

    def __init__(self, message, code=400):
        super().__init__(message)
        self.message = message
        self.code = code


@dataclass
class ShmRef:
    region: str
    byte_size: int
    offset: int = 0

    def __post_init__(self):
        # both values arrive as signed integers (JSON numbers, int64 protobuf fields): a
        # negative one would pass every "offset + size <= region" check and address memory
        # in front of the region
        if self.byte_size < 0 or self.offset < 0:
            raise ServerError(
                f"shared memory offset and byte size must not be negative for region '{self.region}' "
                f"(offset {self.offset}, byte size {self.byte_size})", 400)


@dataclass
class InputTensor:
    name: str
    datatype: str
    shape: List[int]
    data: object = None  # np.ndarray (decoded) or None when shm is set
    raw: object = None  # undecoded wire bytes (decoded lazily against the spec)
    json_data: object = None  # JSON "data" list
    shm: Optional[ShmRef] = None


@dataclass
class RequestedOutput:
    name: str
    binary: Optional[bool] = None  # HTTP only; None = use request default
    class_count: int = 0
    shm: Optional[ShmRef] = None


@dataclass
class InferRequest:
    model_name: str
    model_version: str = ""
    id: str = ""
    parameters: Dict[str, object] = field(default_factory=dict)
    inputs: List[InputTensor] = field(default_factory=list)
    outputs: Optional[List[RequestedOutput]] = None
    streaming: bool = False


class _RequestPlan:
    """Everything a repeat of one raw request message needs (see ``ServerCore.try_planned``)."""

    __slots__ = ("req", "lm", "inputs", "batch", "stable", "requested", "requested_names", "version", "stats",
                 "regions", "dests")

    def __init__(self, req, lm, inputs, batch, stable, requested, requested_names, version, stats, regions, dests):
        self.req, self.lm, self.inputs, self.batch, self.stable = req, lm, inputs, batch, stable
        self.requested, self.requested_names, self.version, self.stats = requested, requested_names, version, stats
        self.regions, self.dests = regions, dests


@dataclass
class OutputTensor:
    name: str
    datatype: str
    shape: List[int]
    data: object = None  # np.ndarray, or None when written to shm
    in_shm: Optional[ShmRef] = None
    binary: Optional[bool] = None


@dataclass
class InferResponse:
    model_name: str
    model_version: str
    id: str = ""
    outputs: List[OutputTensor] = field(default_factory=list)
    parameters: Dict[str, object] = field(default_factory=dict)


# --------------------------------------------------------------------------
# statistics
# --------------------------------------------------------------------------
class _Duration:
    __slots__ = ("count", "ns")

    def __init__(self):
        self.count = 0
        self.ns = 0

    def add(self, ns, count=1):
        self.count += count
        self.ns += int(ns)

    def as_dict(self):
        return {"count": self.count, "ns": self.ns}


class ModelStats:
    FIELDS = ("success", "fail", "queue", "compute_input", "compute_infer", "compute_output", "cache_hit",
              "cache_miss")

    def __init__(self):
        self.lock = threading.Lock()
        self.last_inference = 0
        self.inference_count = 0
        self.execution_count = 0
        self.d = {k: _Duration() for k in self.FIELDS}
        self.batch = {}

    def record(self, batch, queue_ns, in_ns, infer_ns, out_ns, executions=1):
        total = queue_ns + in_ns + infer_ns + out_ns
        with self.lock:
            self.last_inference = int(time.time() * 1000)
            self.inference_count += max(1, batch)
            self.execution_count += executions
            self.d["success"].add(total)
            self.d["queue"].add(queue_ns)
            self.d["compute_input"].add(in_ns)
            self.d["compute_infer"].add(infer_ns)
            self.d["compute_output"].add(out_ns)
            b = self.batch.setdefault(max(1, batch), {k: _Duration() for k in
                                                       ("compute_input", "compute_infer", "compute_output")})
            b["compute_input"].add(in_ns)
            b["compute_infer"].add(infer_ns)
            b["compute_output"].add(out_ns)

    def record_fail(self, ns):
        with self.lock:
            self.d["fail"].add(ns)

    def merge_native(self, taken):
        """Adds what the native execution path counted since the last merge (``take_stats``)."""
        with self.lock:
            self.last_inference = max(self.last_inference, int(taken["last_ms"]))
            self.inference_count += int(taken["inferences"])
            self.execution_count += int(taken["requests"])
            self.d["queue"].add(taken["queue_ns"], taken["requests"])
            total = int(taken["queue_ns"])
            for batch, (count, in_ns, infer_ns, out_ns) in taken["by_batch"].items():
                total += in_ns + infer_ns + out_ns
                self.d["compute_input"].add(in_ns, count)
                self.d["compute_infer"].add(infer_ns, count)
                self.d["compute_output"].add(out_ns, count)
                b = self.batch.setdefault(max(1, batch), {k: _Duration() for k in
                                                           ("compute_input", "compute_infer", "compute_output")})
                b["compute_input"].add(in_ns, count)
                b["compute_infer"].add(infer_ns, count)
                b["compute_output"].add(out_ns, count)
            self.d["success"].add(total, taken["requests"])

    def as_dict(self, name, version):
        with self.lock:
            return {
                "name": name,
                "version": version,
                "last_inference": self.last_inference,
                "inference_count": self.inference_count,
                "execution_count": self.execution_count,
                "inference_stats": {k: v.as_dict() for k, v in self.d.items()},
                "batch_stats": [
                    {"batch_size": bs, **{k: v.as_dict() for k, v in d.items()}} for bs, d in sorted(self.batch.items())
                ],
            }


# --------------------------------------------------------------------------
# shared memory registries
# --------------------------------------------------------------------------
class _Leased:
    """Lease bookkeeping shared by both shared-memory registries.

    A request that names a region holds a lease on it from validation until its answer has
    been produced (queued in the batcher, running on the GPU, being copied out).  Unregister
    first stops new leases, then waits for the outstanding ones; a region that is still busy
    after ``drain_timeout_s`` stays registered and the call fails with an "in use" error."""

    def __init__(self, drain_timeout_s=10.0):
        self._lock = threading.Lock()
        self._idle = threading.Condition(self._lock)
        self._regions = {}
        self.drain_timeout_s = drain_timeout_s

    def acquire(self, name):
        """Lease on region ``name`` (None when it is not registered or is being unregistered).
        Every successful acquire is paired with one :meth:`release`."""
        with self._lock:
            entry = self._regions.get(name)
            if entry is None or entry.closing:
                return None
            entry.inflight += 1
            return entry

    def release(self, entry):
        with self._lock:
            entry.inflight -= 1
            if entry.inflight <= 0:
                self._idle.notify_all()

    def _drain(self, name):
        """Under ``self._lock``: mark the named region (all when empty) as closing, wait for
        their leases and remove the idle ones.  Returns ``(removed [(name, entry)], busy [name])``."""
        names = [name] if name else list(self._regions)
        entries = []
        for n in names:
            entry = self._regions.get(n)
            if entry is not None and not entry.closing:
                entry.closing = True  # no new leases from here on
                entries.append((n, entry))
        deadline = time.monotonic() + self.drain_timeout_s
        removed, busy = [], []
        for n, entry in entries:
            while entry.inflight > 0:
                remaining = deadline - time.monotonic()
                if remaining <= 0:
                    break
                self._idle.wait(remaining)
            if entry.inflight > 0:
                entry.closing = False  # still serving requests: leave it registered
                busy.append(n)
            else:
                del self._regions[n]
                removed.append((n, entry))
        return removed, busy

    @staticmethod
    def _raise_busy(busy):
        if busy:
            raise ServerError(
                f"Cannot unregister shared memory region '{busy[0]}': it is in use by in-flight inference requests",
                400)


class _SystemRegion:
    __slots__ = ("key", "offset", "byte_size", "handle", "inflight", "closing")

    def __init__(self, key, offset, byte_size, handle):
        self.key, self.offset, self.byte_size, self.handle = key, int(offset), int(byte_size), handle
        self.inflight = 0
        self.closing = False

    def __getitem__(self, i):
        return (self.key, self.offset, self.byte_size, self.handle)[i]


class SystemShmRegistry(_Leased):
    native = None  # NativeExecutor that mirrors the registrations (its own mapping of every region)

    def register(self, name, key, offset, byte_size):
        if int(offset) < 0 or int(byte_size) < 0:
            raise ServerError(f"Unable to open shared memory region: '{key}': negative offset or byte size", 400)
        with self._lock:
            if name in self._regions:
                raise ServerError(f"shared memory region '{name}' already in manager", 400)
            try:
                handle = sysshm.attach_shared_memory_region(name, key, byte_size, offset)
            except sysshm.SharedMemoryException as e:
                raise ServerError(f"Unable to open shared memory region: '{key}': {e}", 400) from None
            self._regions[name] = _SystemRegion(key, offset, byte_size, handle)
        if self.native is not None:
            try:
                self.native.register_region(name, key if key.startswith("/") else "/" + key, int(offset),
                                            int(byte_size))
            except ValueError:
                pass  # requests naming this region take the general path

    def unregister(self, name=""):
        if self.native is not None:
            self.native.unregister_region(name or "")  # first: no request may start on a region that is going away
        with self._lock:
            removed, busy = self._drain(name)
            still = [(n, self._regions[n]) for n in busy if n in self._regions]
        for _, entry in removed:
            try:
                sysshm.destroy_shared_memory_region(entry.handle)
            except sysshm.SharedMemoryException:
                pass
        for n, entry in still:
            # still leased by requests of the general path: it stays registered, so the executor serves it again
            try:
                self.native.register_region(n, entry.key if entry.key.startswith("/") else "/" + entry.key,
                                            entry.offset, entry.byte_size)
            except (ValueError, AttributeError):
                pass
        self._raise_busy(busy)

    def status(self, name=""):
        with self._lock:
            if name:
                if name not in self._regions:
                    raise ServerError(f"Unable to find system shared memory region: '{name}'", 400)
                items = {name: self._regions[name]}
            else:
                items = dict(self._regions)
        return [{"name": n, "key": e[0], "offset": e[1], "byte_size": e[2]} for n, e in items.items()]

    def view(self, ref: ShmRef):
        """Writable memoryview of ``ref`` inside its region."""
        with self._lock:
            entry = self._regions.get(ref.region)
        if entry is None:
            return None
        size = entry[2]
        if ref.offset < 0 or ref.byte_size < 0 or ref.offset + ref.byte_size > size:
            raise ServerError(
                f"Invalid offset + byte size for shared memory region: '{ref.region}'", 400)
        return entry[3].buffer()[ref.offset: ref.offset + ref.byte_size]


class _DeviceRegion:
    """One registered device region: ``(device_id, byte_size, mapping)`` for the callers that
    index it like the tuple it used to be, plus the number of requests that currently hold it."""

    __slots__ = ("device_id", "byte_size", "mapping", "inflight", "closing")

    def __init__(self, device_id, byte_size, mapping):
        self.device_id, self.byte_size, self.mapping = int(device_id), int(byte_size), mapping
        self.inflight = 0
        self.closing = False

    def __getitem__(self, i):
        return (self.device_id, self.byte_size, self.mapping)[i]

    def __iter__(self):
        return iter((self.device_id, self.byte_size, self.mapping))


class DeviceShmRegistry(_Leased):
    """Regions opened from HIP IPC handles (protocol name: cudasharedmemory).

    Unregister waits for the leases of in-flight requests (see :class:`_Leased`), drops every
    cached view, asks the loaded models to drop captured graphs that address the region
    (``on_close`` hooks) and only then closes the IPC mapping: the GPU never replays a graph
    or runs a copy on an unmapped address."""

    native = None  # NativeExecutor that mirrors the registrations (name -> device address range)

    def __init__(self, drain_timeout_s=10.0):
        super().__init__(drain_timeout_s)
        self._views = {}  # (region, datatype, shape, offset) -> torch view; building one costs ~20 us
        self.on_close = []  # callables (base address, byte size): a mapping is about to be closed

    def _mirror(self, name, entry):
        """Tells the native executor about a region (its device BYTES path addresses it directly)."""
        if self.native is None or entry.mapping.host_backed:
            return
        try:
            self.native.register_device_region(name, int(entry.mapping.address), int(entry.byte_size),
                                               int(entry.device_id))
        except ValueError:
            pass  # requests naming this region take the general path

    def tensor_view(self, name, mapping, datatype, shape, offset):
        """Torch view of a slice of a registered region, cached: requests of a
        closed-loop client name the same slices over and over."""
        key = (name, datatype, tuple(shape), offset)
        view = self._views.get(key)
        if view is None:
            from ..utils import hip_shared_memory as hipshm

            try:
                view = hipshm.mapping_as_torch(mapping, datatype, shape, offset)
            except ValueError as e:
                raise ServerError(f"invalid tensor in shared memory region '{name}': {e}", 400) from None
            with self._lock:
                if name in self._regions and self._regions[name].mapping is mapping:
                    if len(self._views) > 65536:
                        self._views.clear()
                    self._views[key] = view
        return view

    def register(self, name, raw_handle: bytes, device_id, byte_size):
        from ..utils import hip_shared_memory as hipshm

        if int(byte_size) < 0 or int(device_id) < 0:
            raise ServerError(f"failed to register device shared memory region '{name}': negative size or device", 400)
        with self._lock:
            if name in self._regions:
                raise ServerError(f"shared memory region '{name}' already in manager", 400)
            try:
                with _trace.range(f"hipIpcOpenMemHandle {name}"):
                    mapping = hipshm.open_shared_memory_region(name, raw_handle, byte_size, device_id)
            except Exception as e:
                raise ServerError(f"failed to register device shared memory region '{name}': {e}", 400) from None
            entry = self._regions[name] = _DeviceRegion(device_id, byte_size, mapping)
        self._mirror(name, entry)

    def unregister(self, name=""):
        from ..utils import hip_shared_memory as hipshm

        if self.native is not None:
            # first: no natively executed request may start on a region that is going away, and the
            # queued ones finish before the mapping is closed
            if not self.native.unregister_device_region(name or "", int(self.drain_timeout_s * 1000)):
                self._raise_busy([name or "*"])
        with self._lock:
            removed, busy = self._drain(name)
            for n, _ in removed:
                for key in [k for k in self._views if k[0] == n]:
                    del self._views[key]
            still = [(n, self._regions[n]) for n in busy if n in self._regions]
        for n, entry in still:
            self._mirror(n, entry)  # still leased by the general path: stays registered, natively too
        for n, entry in removed:
            base, size = entry.mapping.address, entry.byte_size
            for hook in list(self.on_close):
                try:
                    hook(base, size)
                except Exception:
                    pass
            try:
                hipshm.close_shared_memory_region(entry.mapping)
            except Exception:
                pass
        self._raise_busy(busy)

    def status(self, name=""):
        with self._lock:
            if name:
                if name not in self._regions:
                    raise ServerError(f"Unable to find cuda shared memory region: '{name}'", 400)
                items = {name: self._regions[name]}
            else:
                items = dict(self._regions)
        return [{"name": n, "device_id": e.device_id, "byte_size": e.byte_size} for n, e in items.items()]

    def get(self, name):
        """Entry of region ``name`` for a caller that holds a lease on it (see ``acquire``)."""
        with self._lock:
            return self._regions.get(name)


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _shape_matches(dims, shape):
    if len(dims) != len(shape):
        return False
    return all(d == -1 or d == s for d, s in zip(dims, shape))


def _decode_wire(datatype, shape, raw, name, model):
    """Wire bytes -> ndarray of ``shape``."""
    count = math.prod(shape) if len(shape) else 1
    if datatype == "BYTES":
        arr = deserialize_bytes_tensor(raw)
        if arr.size != count:
            raise ServerError(
                f"expected {count} string elements for inference input '{name}' for model '{model}', got {arr.size}")
        return arr.reshape(shape)
    if datatype == "BF16":
        arr = deserialize_bf16_tensor(raw)
    else:
        npdt = triton_to_np_dtype(datatype)
        if npdt is None:
            raise ServerError(f"invalid datatype '{datatype}' for input '{name}'")
        itemsize = np.dtype(npdt).itemsize
        if len(raw) != count * itemsize:
            raise ServerError(
                f"unexpected total byte size {len(raw)} for input '{name}', expecting {count * itemsize}")
        arr = np.frombuffer(raw, dtype=npdt)
    if arr.size != count:
        raise ServerError(f"unexpected element count {arr.size} for input '{name}', expecting {count}")
    return arr.reshape(shape)


def _decode_json(datatype, shape, data, name):
    flat = np.array(data, dtype=object).reshape(-1).tolist() if isinstance(data, list) else [data]
    count = math.prod(shape) if len(shape) else 1
    if len(flat) != count:
        raise ServerError(f"unexpected number of elements {len(flat)} in 'data' for input '{name}', expecting {count}")
    if datatype == "BYTES":
        arr = np.empty(count, dtype=np.object_)
        for i, v in enumerate(flat):
            arr[i] = v.encode("utf-8") if isinstance(v, str) else bytes(v)
        return arr.reshape(shape)
    if datatype in ("FP16", "BF16"):
        npdt = np.float16 if datatype == "FP16" else np.float32
    else:
        npdt = triton_to_np_dtype(datatype)
    return np.array(flat, dtype=npdt).reshape(shape)


ASYNC = object()  # returned by InferenceCore.infer when the answer comes through on_done

_DEVICE_BYTES = None


def _device_bytes_class():
    global _DEVICE_BYTES
    if _DEVICE_BYTES is None:
        from ..ops.bytes_pipe import DeviceBytes

        _DEVICE_BYTES = DeviceBytes
    return _DEVICE_BYTES


class _UnstableDeviceTensor:
    """Marks a device tensor that was copied out of a shared-memory region (so its address is
    not that of the region and must not key a captured graph)."""

    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = tensor


class PackedBytes:
    """A BYTES tensor that already is in wire form (``<u32 len><bytes>`` per
    element), e.g. classification strings formatted by the GPU kernel.  Fronts
    that need the elements (JSON output) unpack it on demand."""

    __slots__ = ("blob", "shape")
    dtype = np.dtype(np.object_)

    def __init__(self, blob, shape):
        self.blob = blob
        self.shape = tuple(shape)

    def unpack(self):
        return deserialize_bytes_tensor(self.blob).reshape(self.shape)

    def reshape(self, *shape):
        return self.unpack().reshape(*shape)


def encode_wire(datatype, arr):
    """ndarray -> wire bytes."""
    if isinstance(arr, PackedBytes):
        return arr.blob
    if datatype == "BYTES":
        packed = serialize_byte_tensor(arr)
        return packed.item() if packed.size else b""
    if datatype == "BF16":
        packed = serialize_bf16_tensor(np.ascontiguousarray(arr, dtype=np.float32))
        return packed.item() if packed.size else b""
    return np.ascontiguousarray(arr).tobytes()


def format_classification(values, labels, k):
    """Top-k ``"value:index[:label]"`` strings for each row of ``values``."""
    rows = values.reshape(-1, values.shape[-1]) if values.ndim > 1 else values.reshape(1, -1)
    k = min(k, rows.shape[1])
    idx = np.argsort(-rows, axis=1, kind="stable")[:, :k]
    out = np.empty((rows.shape[0], k), dtype=np.object_)
    for r in range(rows.shape[0]):
        for c in range(k):
            i = int(idx[r, c])
            text = "%f:%d" % (float(rows[r, i]), i)
            if labels is not None and i < len(labels):
                text += ":" + labels[i]
            out[r, c] = text.encode("utf-8")
    return out if values.ndim > 1 else out.reshape(k)


# --------------------------------------------------------------------------
# loaded model wrapper
# --------------------------------------------------------------------------
class LoadedModel:
    def __init__(self, model, device_index=None):
        self.model = model
        self.stats = {v: ModelStats() for v in model.versions}
        self.label_table = None  # class labels resident on the GPU, built on first classification request
        self.sequences = {}
        self.seq_lock = threading.Lock()
        self.exec_lock = threading.Lock()  # CPU models run serially per instance like a single instance group
        self.request_counter = 0
        self.device_index = device_index
        self.batcher = None
        # signature lookups of the request path (the model's tensor lists do not change after load)
        self.in_specs = {t.name: t for t in model.inputs}
        self.out_specs = {t.name: t for t in model.outputs}


class InferenceCore:
    """See module docstring."""

    def __init__(self, models=None, gpu=None, model_control="none", strict_ready=True):
        self.gpu = gpu
        self.model_control = model_control
        self._lock = threading.RLock()
        self._loaded: Dict[str, LoadedModel] = {}
        self._plans = {}  # raw request message -> _RequestPlan (see try_planned)
        self._plan_counts = {"hits": 0, "stored": 0}
        self._failed: Dict[str, str] = {}
        self.sysshm = SystemShmRegistry()
        self.devshm = DeviceShmRegistry()
        self.devshm.on_close.append(self._drop_plans_of_region)
        self.native = None  # NativeExecutor once attach_native() has run
        self.trace_global = {
            "trace_file": [""], "trace_level": ["OFF"], "trace_rate": ["1000"], "trace_count": ["-1"],
            "log_frequency": ["0"], "trace_mode": ["triton"],
        }
        self.trace_models: Dict[str, dict] = {}
        self.log_settings = {
            "log_file": "", "log_info": True, "log_warning": True, "log_error": True, "log_verbose_level": 0,
            "log_format": "default",
        }
        self.live = True
        self.ready = True
        self._available = list(models) if models is not None else list(MODEL_ZOO)
        if model_control != "explicit":
            for name in self._available:
                try:
                    self.load_model(name)
                except ServerError as e:
                    self._failed[name] = e.message

    # ------------------------------------------------------ native execution
    def attach_native(self, executor):
        """Hands models with a C++ kernel (``native_kernel`` on the model class) and the system
        shared-memory registrations to ``executor``; the fronts then answer ordinary requests for
        those models without the interpreter.  Model load / unload and region registration keep
        it in step from here on."""
        self.native = executor
        self.sysshm.native = executor
        for entry in self.sysshm.status():
            try:
                key = entry["key"]
                executor.register_region(entry["name"], key if key.startswith("/") else "/" + key,
                                         int(entry["offset"]), int(entry["byte_size"]))
            except ValueError:
                pass
        self._attach_device_bytes(executor)
        with self._lock:
            loaded = list(self._loaded.values())
        for lm in loaded:
            if hasattr(lm.model, "capture_gate"):
                lm.model.capture_gate = executor  # stream captures and the executor's graph launches take turns
            self._sync_native(lm)

    def _attach_device_bytes(self, executor):
        """With a GPU: gives the executor a device BYTES pipeline of its own (the kernels of
        ``ops/bytes_pipe.py``, driven from its batch thread) and the device shared-memory
        registrations, so BYTES models with a device path are answered without the interpreter."""
        if self.gpu is None or os.environ.get("TCA_NATIVE_DEVICE_BYTES", "1") == "0":
            return
        try:
            import ctypes

            from ..ops import _device
            from ..ops.bytes_pipe import BytesPipe, max_stream_bytes

            max_requests, max_count = 64, 4096
            pipe = BytesPipe(int(self.gpu), max_requests, max_count)
            stream = _device.Stream(int(self.gpu))
            lib = _device.lib()
            address = lambda fn: ctypes.cast(fn, ctypes.c_void_p).value  # noqa: E731
            executor.enable_device_bytes(address(lib.tca_bytes_pipe_submit), address(lib.tca_bytes_pipe_wait),
                                         int(pipe._h.value), int(stream.handle), int(self.gpu), max_requests, max_count,
                                         int(max_stream_bytes()))
        except Exception:  # noqa: BLE001 - no HIP library / no device: the general path serves these models
            return
        self._native_pipe = (pipe, stream)  # alive as long as the executor's batch thread
        self.devshm.native = executor
        with self.devshm._lock:
            entries = list(self.devshm._regions.items())
        for name, entry in entries:
            self.devshm._mirror(name, entry)

    def _sync_native(self, lm, present=True):
        executor = getattr(self, "native", None)
        if executor is None:
            return
        model = lm.model
        kernel = getattr(model, "native_kernel", None)
        if kernel == "graph":
            self._sync_native_graph(lm, present)
            return
        stateful = kernel in ("sequence", "dyna_sequence")
        eligible = present and kernel and model.device == "cpu" and not model.parameters \
            and bool(getattr(model, "sequence", False)) == stateful and (not model.decoupled or kernel == "repeat") \
            and (not kernel.startswith("bytes_") or self.devshm.native is not None)
        lm.native_sequences = False
        if not eligible:
            executor.remove_model(model.name)
            return
        try:
            executor.set_model(model.name, model.versions[-1], int(model.max_batch_size),
                               [(t.name, t.datatype, [int(d) for d in t.dims]) for t in model.inputs],
                               [(t.name, t.datatype, [int(d) for d in t.dims]) for t in model.outputs], kernel)
            # the table of open sequences (numeric correlation ids) lives in the executor from here on:
            # requests answered there and requests that come through infer() see the same one
            lm.native_sequences = bool(stateful)
        except ValueError:
            executor.remove_model(model.name)

    def _sync_native_graph(self, lm, present):
        """GPU models served from captured HIP graphs (``native_kernel = "graph"``): the executor gets the
        model's signature, a set of graphs captured over buffers of its own for every execution instance, the
        instances' streams and the batching policy of the dynamic batcher, and from then on answers requests
        whose tensors are in device shared memory or inside the messages without the interpreter (queue,
        leases, batch forming, gather / graph launch / scatter, completion and response are all native; tensors
        inside messages pass through page-locked memory the kernels address directly).  Everything else -
        classification, system shared memory, JSON tensors, errors - keeps taking the general path."""
        executor, model = self.native, lm.model
        name = model.name
        eligible = (present and lm.batcher is not None and self.devshm.native is not None and not model.parameters
                    and getattr(model, "use_graphs", False) and len(model.inputs) == 1 and len(model.outputs) == 1
                    and os.environ.get("TCA_NATIVE_GRAPHS", "1") != "0")
        if not eligible:
            executor.remove_model(name)  # stops its threads first
            lm.native_graphs = None
            return
        try:
            import ctypes

            from ..ops import _device

            batcher = lm.batcher
            warmed, model.warmed_up = getattr(model, "warmed_up", False), False  # these captures are load-time ones
            try:
                plan = model.native_graph_plan(batcher.instances)
            finally:
                model.warmed_up = warmed
            lib = _device.lib()
            address = lambda fn: ctypes.cast(fn, ctypes.c_void_p).value  # noqa: E731
            entry_points = tuple(address(getattr(lib, fn)) for fn in (
                "tca_hip_graph_launch", "tca_hip_event_create", "tca_hip_event_record", "tca_hip_event_sync",
                "tca_hip_event_destroy", "tca_hip_set_device", "tca_batched_copy_launch", "tca_hip_stream_sync"))
            executor.set_model(name, model.versions[-1], int(model.max_batch_size),
                               [(t.name, t.datatype, [int(d) for d in t.dims]) for t in model.inputs],
                               [(t.name, t.datatype, [int(d) for d in t.dims]) for t in model.outputs], "graph")
            # page-locked memory for requests whose tensors travel inside the messages: the receiving thread
            # copies the input bytes in, the gather / scatter kernels address it over the bus
            arena_mb = int(os.environ.get("TCA_NATIVE_WIRE_ARENA_MB", "1536"))
            arena = _device.PinnedBuffer(arena_mb << 20) if arena_mb > 0 else None
            # ... and a staging thread moves them on to device memory on a stream of its own, ahead of the
            # batches, with the DMA engines (hipMemcpyAsync).  Measured with both tensors in system shared
            # memory, batch 8, 64 requests in flight: 71k img/s this way; 59k when a copy kernel does the
            # staging (TCA_NATIVE_WIRE_STAGE=kernel: its waves wait on the bus while holding compute units,
            # and the convolutions next to it ran 30 % slower); 52k when the gather kernels in front of each
            # batch read the page-locked memory themselves (TCA_NATIVE_WIRE_STAGE=0).
            stage_mode = os.environ.get("TCA_NATIVE_WIRE_STAGE", "sdma")
            staged = None
            if arena is not None and stage_mode != "0":
                import torch

                staged = (torch.empty(arena.nbytes, dtype=torch.uint8, device=f"cuda:{int(self.gpu)}"),
                          torch.cuda.Stream(device=int(self.gpu)))
            executor.enable_graph_model(
                name, entry_points, int(self.gpu), plan["streams"], plan["graphs"], plan["in_row_bytes"],
                plan["out_row_bytes"], plan["max_batch"], int(os.environ.get("TCA_MAX_INFLIGHT_BATCHES", 0)) or int(
                    getattr(model, "max_inflight_batches", 2 * batcher.instances)),
                int(batcher.fill_window_s * 1e6), int(batcher.burst_gap_s * 1e6), int(batcher.burst_window_s * 1e6),
                int(batcher.split_min_batch),
                # measured with the request path in C++ (tools/dev/low_concurrency_probe.py, batch-8 requests): one
                # merged batch per instance (depth 1) beats two of half the size from 3 to 12 requests in flight
                # (4: 32.2k -> 42.3k infer/s, p50 0.99 -> 0.74 ms; 8: 45.7k -> 52.7k), and from 16 on merging
                # everything queued is better still (16: 59.9k -> 67.2k), hence the limit of 0.8 x the largest batch
                (max(2, int(round(10 * float(os.environ.get("TCA_ADAPTIVE_CAP_DEPTH", "1.0"))))) if batcher._adaptive_cap else 0),
                int(arena.ptr) if arena else 0, int(arena.nbytes) if arena else 0, 256 << 10,
                int(staged[0].data_ptr()) if staged else 0, int(staged[1].cuda_stream) if staged else 0,
                address(lib.tca_hip_memcpy_async) if stage_mode == "sdma" else 0,
                max(1, int(round(10 * float(os.environ.get("TCA_ADAPTIVE_CAP_LIMIT", "0.8"))))))
            lm.native_graphs = (plan["keep"], arena, staged)
        except Exception:  # noqa: BLE001 - the general path serves the model
            import traceback

            traceback.print_exc()
            executor.remove_model(name)
            lm.native_graphs = None

    def _merge_native_stats(self, name, lm):
        executor = getattr(self, "native", None)
        if executor is None:
            return
        taken = executor.take_stats(name)
        if taken:
            lm.stats[lm.model.versions[-1]].merge_native(taken)

    # ----------------------------------------------------------- repository
    def repository_index(self, ready_only=False):
        out = []
        with self._lock:
            for name in self._available:
                lm = self._loaded.get(name)
                if lm is not None:
                    for v in lm.model.versions:
                        out.append({"name": name, "version": v, "state": "READY", "reason": ""})
                elif not ready_only:
                    entry = {"name": name}
                    if name in self._failed:
                        entry.update({"version": "1", "state": "UNAVAILABLE", "reason": self._failed[name]})
                    out.append(entry)
        return out

    def load_model(self, name, config=None, files=None):
        if name not in MODEL_ZOO or name not in self._available:
            raise ServerError(f"failed to load '{name}', failed to poll from model repository", 400)
        cls = MODEL_ZOO[name]
        model = cls()
        if config is not None:
            try:
                override = json.loads(config) if isinstance(config, (str, bytes)) else dict(config)
            except ValueError as e:
                raise ServerError(f"failed to load '{name}', unable to parse config override: {e}", 400) from None
            model.apply_config_override(override)
        elif files:
            raise ServerError(
                f"failed to load '{name}', override model directory requires a config override as well", 400)
        if files:
            model.override_files = {k: (base64.b64decode(v) if isinstance(v, str) else v) for k, v in files.items()}
        if model.device == "gpu" and self.gpu is None:
            raise ServerError(f"failed to load '{name}', no GPU is assigned to this server instance", 400)
        try:
            model.load(self.gpu if model.device == "gpu" else None)
        except ModelError as e:
            raise ServerError(f"failed to load '{name}': {e}", 400) from None
        lm = LoadedModel(model, self.gpu)
        if hasattr(model, "capture_gate") and self.native is not None:
            model.capture_gate = self.native
        if hasattr(model, "evict_region"):
            self.devshm.on_close.append(model.evict_region)
        model.core = self  # ensembles run their steps through infer_tensors()
        if model.device == "gpu" and model.dynamic_batching:
            from .batcher import DynamicBatcher

            lm.batcher = DynamicBatcher(self, lm)
        with self._lock:
            old = self._loaded.get(name)
            self._loaded[name] = lm
            self._failed.pop(name, None)
        self._sync_native(lm)
        if hasattr(model, "online_captures"):
            model.warmed_up = True  # graphs captured from here on count as online captures
        if old is not None:
            self._retire(old)

    def _retire(self, lm):
        if getattr(lm, "native_graphs", None) is not None and self.native is not None:
            # the executor replays this model's graphs: stop that before they go (unless a newer instance of
            # the model has already replaced them there)
            with self._lock:
                current = self._loaded.get(lm.model.name)
            if current is None or current is lm:
                self.native.remove_model(lm.model.name)
            lm.native_graphs = None
        hook = getattr(lm.model, "evict_region", None)
        if hook in self.devshm.on_close:
            self.devshm.on_close.remove(hook)
        if lm.batcher is not None:
            lm.batcher.stop()
        try:
            lm.model.unload()
        except Exception:
            pass

    def unload_model(self, name, unload_dependents=False):
        with self._lock:
            lm = self._loaded.pop(name, None)
        if lm is None and name not in self._available:
            raise ServerError(f"failed to unload '{name}', model is not found", 400)
        if lm is not None:
            self._sync_native(lm, present=False)
            self._retire(lm)

    def _get(self, name, version="") -> LoadedModel:
        with self._lock:
            lm = self._loaded.get(name)
        if lm is None:
            raise ServerError(f"Request for unknown model: '{name}' is not found", 400)
        if version and not version.isdigit():
            raise ServerError(f"failed to get model version from specified version string '{version}' "
                              "(details: stoll): the model version must be an integral number", 400)
        if version and version not in lm.model.versions:
            raise ServerError(f"Request for unknown model: '{name}' version {version} is not found", 400)
        return lm

    def is_model_ready(self, name, version=""):
        try:
            self._get(name, version)
            return True
        except ServerError:
            return False

    def server_metadata(self):
        return {"name": SERVER_NAME, "version": SERVER_VERSION, "extensions": list(EXTENSIONS)}

    def model_metadata(self, name, version=""):
        lm = self._get(name, version)
        md = lm.model.metadata()
        if version:
            md["versions"] = [version]
        return md

    def model_config(self, name, version=""):
        return self._get(name, version).model.config()

    def statistics(self, name="", version=""):
        out = []
        with self._lock:
            items = list(self._loaded.items())
        if name:
            lm = self._get(name, version)
            items = [(name, lm)]
        for n, lm in items:
            self._merge_native_stats(n, lm)
            for v, st in lm.stats.items():
                if version and v != version:
                    continue
                out.append(st.as_dict(n, v))
        return {"model_stats": out}

    def metrics_text(self):
        """Prometheus text exposition of the per-model counters (what a Triton server offers on its
        metrics port, here under ``GET /metrics`` of the HTTP front), plus how many requests the
        native execution path answered or handed on."""
        rows = self.statistics()["model_stats"]
        series = (
            ("nv_inference_request_success", "counter", "Number of successful inference requests",
             lambda r: r["inference_stats"]["success"]["count"]),
            ("nv_inference_request_failure", "counter", "Number of failed inference requests",
             lambda r: r["inference_stats"]["fail"]["count"]),
            ("nv_inference_count", "counter", "Number of inferences performed (a batch of n counts n)",
             lambda r: r["inference_count"]),
            ("nv_inference_exec_count", "counter", "Number of model executions performed",
             lambda r: r["execution_count"]),
            ("nv_inference_request_duration_us", "counter", "Cumulative inference request duration in microseconds",
             lambda r: r["inference_stats"]["success"]["ns"] // 1000),
            ("nv_inference_queue_duration_us", "counter", "Cumulative inference queuing duration in microseconds",
             lambda r: r["inference_stats"]["queue"]["ns"] // 1000),
            ("nv_inference_compute_input_duration_us", "counter", "Cumulative compute input duration in microseconds",
             lambda r: r["inference_stats"]["compute_input"]["ns"] // 1000),
            ("nv_inference_compute_infer_duration_us", "counter", "Cumulative compute inference duration in microseconds",
             lambda r: r["inference_stats"]["compute_infer"]["ns"] // 1000),
            ("nv_inference_compute_output_duration_us", "counter", "Cumulative compute output duration in microseconds",
             lambda r: r["inference_stats"]["compute_output"]["ns"] // 1000),
        )
        lines = []
        for metric, kind, text, value in series:
            lines.append(f"# HELP {metric} {text}")
            lines.append(f"# TYPE {metric} {kind}")
            for r in rows:
                lines.append(f'{metric}{{model="{r["name"]}",version="{r["version"]}"}} {int(value(r))}')
        executor = getattr(self, "native", None)
        if executor is not None:
            counters = executor.counters()
            for key, text in (("handled", "Requests answered by the native execution path"),
                              ("declined", "Requests the native execution path looked at and handed on")):
                lines.append(f"# HELP tca_native_{key}_total {text}")
                lines.append(f"# TYPE tca_native_{key}_total counter")
                lines.append(f"tca_native_{key}_total {int(counters[key])}")
            with self._lock:
                loaded = list(self._loaded.items())
            graph_rows = [(name, executor.graph_stats(name)) for name, lm in loaded
                          if getattr(lm, "native_graphs", None) is not None]
            for key, text in (("batches", "Batches the native graph path launched"),
                              ("requests", "Requests merged into those batches"),
                              ("images", "Batch entries of those requests"),
                              ("padded_images", "Rows computed beyond the merged batches (next captured size)"),
                              ("wire_requests", "Requests with a tensor inside the message, staged in page-locked memory"),
                              ("arena_full", "Such requests handed to the general path because the staging memory was taken")):
                lines.append(f"# HELP tca_native_graph_{key}_total {text}")
                lines.append(f"# TYPE tca_native_graph_{key}_total counter")
                for name, row in graph_rows:
                    if row:
                        lines.append(f'tca_native_graph_{key}_total{{model="{name}"}} {int(row[key])}')
        with self._lock:
            models = list(self._loaded.items())
        lines.append("# HELP tca_graph_captures_online_total HIP graphs captured while serving (after load-time warm-up)")
        lines.append("# TYPE tca_graph_captures_online_total counter")
        for name, lm in models:
            if hasattr(lm.model, "online_captures"):
                lines.append(f'tca_graph_captures_online_total{{model="{name}"}} {int(lm.model.online_captures)}')
        for key, text in (("hits", "Requests queued from the stored plan of their raw message"),
                          ("stored", "Request plans stored")):
            lines.append(f"# HELP tca_request_plan_{key}_total {text}")
            lines.append(f"# TYPE tca_request_plan_{key}_total counter")
            lines.append(f"tca_request_plan_{key}_total {self._plan_counts[key]}")
        return "\n".join(lines) + "\n"

    # ----------------------------------------------------------- trace / log
    def update_trace(self, model_name, settings):
        if model_name:
            self._get(model_name)
            target = self.trace_models.setdefault(model_name, {})
        else:
            target = self.trace_global
        for key, value in (settings or {}).items():
            if key not in self.trace_global:
                raise ServerError(f"unsupported trace setting '{key}'", 400)
            if value is None or value == []:
                if model_name:
                    target.pop(key, None)  # fall back to the global value
                continue
            values = value if isinstance(value, list) else [value]
            target[key] = [str(v) for v in values]
        return self.get_trace(model_name)

    def get_trace(self, model_name=None):
        merged = {k: list(v) for k, v in self.trace_global.items()}
        if model_name:
            self._get(model_name)
            for k, v in self.trace_models.get(model_name, {}).items():
                merged[k] = list(v)
        return merged

    def update_log(self, settings):
        for key, value in (settings or {}).items():
            if key not in self.log_settings:
                raise ServerError(f"unsupported log setting '{key}'", 400)
            if value is None:
                continue
            self.log_settings[key] = value
        return dict(self.log_settings)

    # -------------------------------------------------------------- infer
    def _prepare_inputs(self, lm: LoadedModel, req: InferRequest):
        """Validate against the model signature and materialise the inputs.

        Returns ``(inputs: dict, batch: int, device_views: set)``."""
        model = lm.model
        specs = lm.in_specs
        seen = set()
        inputs = {}
        batch = 0
        gpu = model.device == "gpu"
        stable = set()
        deferred = []
        for t in req.inputs:
            spec = specs.get(t.name)
            if spec is None:
                raise ServerError(f"unexpected inference input '{t.name}' for model '{model.name}'")
            if t.name in seen:
                raise ServerError(f"input '{t.name}' already exists in request")
            seen.add(t.name)
            if t.datatype != spec.datatype:
                raise ServerError(
                    f"inference input '{t.name}' data-type is '{t.datatype}', but model '{model.name}' expects "
                    f"'{spec.datatype}'")
            shape = [int(s) for s in t.shape]
            dims = list(spec.dims)
            if model.max_batch_size > 0:
                if len(shape) < 1 or not _shape_matches(dims, shape[1:]):
                    raise ServerError(
                        f"unexpected shape for input '{t.name}' for model '{model.name}'. Expected "
                        f"{[-1] + dims}, got {shape}")
                if shape[0] > model.max_batch_size:
                    raise ServerError(
                        f"inference request batch-size must be <= {model.max_batch_size} for '{model.name}'")
                batch = max(batch, shape[0])
            elif not _shape_matches(dims, shape):
                raise ServerError(
                    f"unexpected shape for input '{t.name}' for model '{model.name}'. Expected {dims}, got {shape}")
            if t.shm is not None:
                arr = self._read_shm(t, shape, model, gpu, deferred)
                if isinstance(arr, _UnstableDeviceTensor):
                    arr = arr.tensor
                elif gpu and not isinstance(arr, np.ndarray) and arr is not None:
                    stable.add(t.name)
            elif t.data is not None:
                arr = t.data
            elif t.raw is not None:
                arr = _decode_wire(t.datatype, shape, t.raw, t.name, model.name)
            elif t.json_data is not None:
                arr = _decode_json(t.datatype, shape, t.json_data, t.name)
            else:
                raise ServerError(f"input '{t.name}' for model '{model.name}' has no data")
            inputs[t.name] = arr
        if deferred:
            # tensors that go through the host although they sit in device regions (BYTES,
            # inputs of CPU models): all ranges of the request move in one native call
            from ..utils import hip_shared_memory as hipshm

            hosts = hipshm.read_mappings([(mapping, ref.offset, ref.byte_size) for _, _, mapping, ref in deferred])
            for (t, shape, _, _), host in zip(deferred, hosts):
                inputs[t.name] = _decode_wire(t.datatype, shape, host, t.name, model.name)
        missing = [n for n, s in specs.items() if n not in seen and not s.optional]
        if missing:
            raise ServerError(
                f"expected {len(specs)} inputs but got {len(seen)} inputs for model '{model.name}'. "
                f"Got input(s) {sorted(seen)}, but missing required input(s) {missing}. "
                "Please provide all required input(s).")
        return inputs, batch, stable

    def _read_shm(self, t: InputTensor, shape, model, gpu, deferred):
        ref = t.shm
        view = self.sysshm.view(ref)
        if view is not None:
            return _decode_wire(t.datatype, shape, view, t.name, model.name)
        entry = self.devshm.get(ref.region)
        if entry is None:
            raise ServerError(f"Unable to find shared memory region: '{ref.region}'", 400)
        from ..utils import hip_shared_memory as hipshm

        device_id, size, mapping = entry
        if ref.offset + ref.byte_size > size:
            raise ServerError(f"Invalid offset + byte size for shared memory region: '{ref.region}'", 400)
        if gpu and t.datatype != "BYTES":
            itemsize = 2 if t.datatype == "BF16" else np.dtype(triton_to_np_dtype(t.datatype)).itemsize
            need = (math.prod(shape) if len(shape) else 1) * itemsize
            if need > ref.byte_size:
                raise ServerError(
                    f"input '{t.name}' for model '{model.name}' needs {need} bytes, but the request gives it "
                    f"{ref.byte_size} bytes of shared memory region '{ref.region}'", 400)
            if (mapping.address + ref.offset) % itemsize:
                raise ServerError(
                    f"shared memory offset {ref.offset} of input '{t.name}' is not a multiple of the element size "
                    f"({itemsize})", 400)
            view = self.devshm.tensor_view(ref.region, mapping, t.datatype, shape, ref.offset)
            if (mapping.address + ref.offset) % 16:
                # the fused kernels load 4- to 16-byte words from the input: a tensor at an odd
                # offset is copied to an aligned buffer (and is no longer a stable address)
                import torch

                copy = view.clone()
                torch.cuda.current_stream(copy.device).synchronize()  # consumers run on other streams
                return _UnstableDeviceTensor(copy)
            return view
        if (t.datatype == "BYTES" and getattr(model, "device_bytes", False) and not mapping.host_backed
                and ref.byte_size > 0):
            # the model can walk the stream on the GPU: hand over the range, not a host copy
            return _device_bytes_class()(mapping.address + ref.offset, ref.byte_size, shape, mapping, ref.offset,
                                         device_id)
        deferred.append((t, shape, mapping, ref))  # read together with the request's other ranges
        return None

    def _output_destination(self, lm, ro: RequestedOutput, spec, batch):
        """For GPU models: a torch view of the device shm destination, if any."""
        if ro.shm is None or ro.class_count:
            return None
        entry = self.devshm.get(ro.shm.region)
        if entry is None or spec.datatype == "BYTES" or any(d < 0 for d in spec.dims):
            return None
        from ..utils import hip_shared_memory as hipshm

        shape = ([batch] if lm.model.max_batch_size > 0 else []) + list(spec.dims)
        need = math.prod(shape) * np.dtype(triton_to_np_dtype(spec.datatype)).itemsize
        if need > ro.shm.byte_size or ro.shm.offset + need > entry[1]:
            raise ServerError(
                f"shared memory size specified with the request for output '{ro.name}' ({ro.shm.byte_size} bytes) "
                f"should be at least {need} bytes to hold the results")
        if (entry[2].address + ro.shm.offset) % 16:
            return None  # odd offset: the kernels write an aligned buffer, the result is copied into the region
        return self.devshm.tensor_view(ro.shm.region, entry[2], spec.datatype, shape, ro.shm.offset)

    def _sequence_ctx(self, lm, req):
        params = req.parameters
        seq_id = params.get("sequence_id", 0)
        if not lm.model.sequence:
            return None
        if seq_id in (0, "", None):
            raise ServerError(
                f"inference request to model '{lm.model.name}' must specify a non-zero or non-empty correlation ID")
        start = bool(params.get("sequence_start", False))
        end = bool(params.get("sequence_end", False))
        if getattr(lm, "native_sequences", False) and type(seq_id) is int and 0 < seq_id < (1 << 63):
            if not self.native.sequence_step(lm.model.name, seq_id, start, end):
                raise ServerError(
                    f"inference request for sequence {seq_id} to model '{lm.model.name}' must specify the START "
                    "flag on the first request of the sequence")
            return SequenceInfo(sequence_id=seq_id, start=start, end=end, state={})
        with lm.seq_lock:
            state = lm.sequences.get(seq_id)
            if start or state is None:
                if state is None and not start:
                    raise ServerError(
                        f"inference request for sequence {seq_id} to model '{lm.model.name}' must specify the START "
                        "flag on the first request of the sequence")
                state = {}
                lm.sequences[seq_id] = state
            if end:
                lm.sequences.pop(seq_id, None)
        return SequenceInfo(sequence_id=seq_id, start=start, end=end, state=state)

    def _inject_faults(self, lm, req):
        params = lm.model.parameters
        delay = params.get("delay_ms")
        if delay:
            time.sleep(float(delay) / 1000.0)
        every = params.get("fail_every_n")
        if every:
            lm.request_counter += 1
            if lm.request_counter % int(every) == 0:
                raise ServerError(f"injected failure for model '{lm.model.name}' (fail_every_n={every})", 500)

    def stream_drop_after(self, name):
        """Fault injection for streaming clients: ``drop_stream_after_n`` (model
        parameter) makes a gRPC front abort a stream with UNAVAILABLE once it has
        written that many responses of this model on it.  0 = never."""
        with self._lock:
            lm = self._loaded.get(name)
        if lm is None:
            return 0
        try:
            return int(lm.model.parameters.get("drop_stream_after_n") or 0)
        except (TypeError, ValueError):
            return 0

    def uses_batcher(self, name):
        """Whether requests for ``name`` complete asynchronously (GPU model
        behind the dynamic batcher)."""
        with self._lock:
            lm = self._loaded.get(name)
        return lm is not None and lm.batcher is not None

    def runs_inline(self, name):
        """Whether a transport's event loop may run ``name`` itself instead of handing the
        request to a worker thread: a stateless CPU model that answers in microseconds
        (``inline = True`` on the model class) with no delay / fault-injection parameters."""
        with self._lock:
            lm = self._loaded.get(name)
        if lm is None:
            return False
        model = lm.model
        return bool(getattr(model, "inline", False)) and model.device == "cpu" and not model.parameters \
            and not model.decoupled

    def _lease_regions(self, req):
        """Take a lease on every shared-memory region the request names, so none of them
        can be unmapped until :meth:`release_request` runs (after the answer is complete)."""
        names = []
        for t in req.inputs:
            if t.shm is not None and t.shm.region not in names:
                names.append(t.shm.region)
        for ro in req.outputs or ():
            if ro.shm is not None and ro.shm.region not in names:
                names.append(ro.shm.region)
        if not names:
            return
        leases = req._leases = []
        req._lease_names = names
        for n in names:
            entry = self.sysshm.acquire(n)
            registry = self.sysshm
            if entry is None:
                entry = self.devshm.acquire(n)
                registry = self.devshm
            if entry is None:
                self.release_request(req)
                raise ServerError(f"Unable to find shared memory region: '{n}'", 400)
            leases.append((registry, entry))

    def release_request(self, req):
        """Give back the request's region leases (idempotent)."""
        leases = getattr(req, "_leases", None)
        if leases:
            req._leases = None
            for registry, entry in leases:
                registry.release(entry)

    # ------------------------------------------------------------ request plans
    # A closed-loop client that keeps its tensors in registered device regions sends the SAME bytes for
    # every request.  What those bytes mean - the parsed message, the validated tensor views of the
    # regions, the output destinations - is kept under the raw message as key, so a repeat costs a
    # dictionary lookup, the region leases and the queue hand-over instead of a parse and a
    # validation (2-3x less interpreter time per request on the zero-copy path).  A plan is only valid
    # against the very region entries and the very loaded model it was built for: a re-registered
    # region or a reloaded model has new objects, and the plan is dropped on the first mismatch.
    _PLAN_CAPACITY = 4096

    def try_planned(self, key, on_done=None, on_done_for=None):
        """Queue the request whose raw message is ``key`` from its stored plan.  True when it
        was queued (``on_done``, or ``on_done_for(request)`` for a front whose answer depends on
        the request, will be called); False when there is no valid plan (the caller parses the
        message and calls :meth:`infer` with ``plan_key``)."""
        plan = self._plans.get(key)
        if plan is None:
            return False
        t_arrive = time.perf_counter_ns()
        with self._lock:
            lm = self._loaded.get(plan.req.model_name)
        leases = []
        if lm is plan.lm and lm.batcher is not None:
            for registry, name, entry in plan.regions:
                got = registry.acquire(name)
                if got is not entry:
                    if got is not None:
                        registry.release(got)
                    break
                leases.append((registry, got))
            else:
                req = copy.copy(plan.req)
                req._leases = leases
                ctx = ExecContext(request_id=req.id, parameters=req.parameters, version=plan.version,
                                  requested_outputs=plan.requested_names)
                ctx.stable_inputs = plan.stable
                if on_done is None:
                    on_done = on_done_for(req)
                lm.batcher.submit_async(req, plan.inputs, plan.batch, plan.stable, plan.requested, ctx, plan.stats,
                                        t_arrive, plan.version, on_done, dests=plan.dests)
                self._plan_counts["hits"] += 1
                return True
        for registry, entry in leases:
            registry.release(entry)
        self._plans.pop(key, None)
        return False

    def _drop_plans_of_region(self, base, nbytes):
        """``DeviceShmRegistry.on_close`` hook: forget the plans whose views alias a mapping that is being closed
        (they would be refused on their next use anyway; this only keeps dead views from piling up)."""
        for key, plan in list(self._plans.items()):
            for _, _, entry in plan.regions:
                mapping = getattr(entry, "mapping", None)
                if mapping is not None and mapping.address == base:
                    self._plans.pop(key, None)
                    break

    def _store_plan(self, key, lm, req, inputs, batch, stable, requested, version, stats):
        """Keep the plan of a request that is about to be queued, if every repeat of it would be
        handled the same way: a batched GPU model without fault-injection parameters, no request
        parameters, every input a zero-copy view of a registered device region."""
        model = lm.model
        if (model.parameters or req.parameters or req.streaming or model.sequence or model.decoupled
                or not req.inputs or any(t.shm is None or t.name not in stable for t in req.inputs)):
            return
        leases, names = getattr(req, "_leases", None), getattr(req, "_lease_names", None)
        if not leases or not names or len(leases) != len(names):
            return
        try:
            dests = {ro.name: self._output_destination(lm, ro, lm.out_specs[ro.name], batch) for ro in requested}
        except ServerError:
            return  # the launch reports it
        if len(self._plans) >= self._PLAN_CAPACITY:
            self._plans.clear()
        template = copy.copy(req)
        template._leases = None
        self._plan_counts["stored"] += 1
        self._plans[key] = _RequestPlan(template, lm, inputs, batch, stable, list(requested),
                                        [r.name for r in requested], version, stats,
                                        [(registry, name, entry) for name, (registry, entry) in zip(names, leases)],
                                        dests)

    def infer(self, req: InferRequest, on_done=None, plan_key=None):
        """Run one request.  Returns an :class:`InferResponse` for ordinary
        models, or an iterator of them for decoupled models.

        With ``on_done`` and a model behind the dynamic batcher the request is
        only queued: :data:`ASYNC` is returned and ``on_done(response, error)``
        is called later from the completer thread, so transport threads never
        block on the GPU."""
        t_arrive = time.perf_counter_ns()
        lm = self._get(req.model_name, req.model_version)
        model = lm.model
        version = req.model_version or model.versions[-1]
        stats = lm.stats[version]
        handed_over = False  # the leases travel with the request (batcher item / response generator)
        try:
            self._lease_regions(req)
            if model.decoupled and not req.streaming:
                raise ServerError(
                    f"doesn't support models with decoupled transaction policy: model '{model.name}' must be called "
                    "through the streaming API", 400)
            self._inject_faults(lm, req)
            t0 = time.perf_counter_ns()
            inputs, batch, stable = self._prepare_inputs(lm, req)
            out_specs = lm.out_specs
            requested = req.outputs
            if requested:
                for ro in requested:
                    if ro.name not in out_specs:
                        raise ServerError(f"unexpected inference output '{ro.name}' for model '{model.name}'")
            else:
                requested = [RequestedOutput(name=t.name) for t in model.outputs]
            ctx = ExecContext(request_id=req.id, parameters=req.parameters, version=version,
                              requested_outputs=[r.name for r in requested])
            ctx.sequence = self._sequence_ctx(lm, req)
            ctx.stable_inputs = stable
            if model.device == "gpu":
                if lm.batcher is not None:
                    if on_done is not None:
                        if plan_key is not None:
                            self._store_plan(plan_key, lm, req, inputs, batch, stable, requested, version, stats)
                        handed_over = True  # released by the batcher when the answer is complete
                        lm.batcher.submit_async(req, inputs, batch, stable, requested, ctx, stats, t_arrive, version,
                                                on_done)
                        return ASYNC
                    return lm.batcher.submit(req, inputs, batch, stable, requested, ctx, stats, t_arrive, version)
                return self._run_gpu_single(lm, req, inputs, batch, stable, requested, ctx, stats, t_arrive, version)
            t1 = time.perf_counter_ns()
            if getattr(model, "device_bytes", False):
                queued = self._run_device_bytes(lm, req, inputs, ctx, requested, out_specs, version, batch, stats,
                                                t_arrive, t0, t1, on_done)
                if queued is ASYNC:
                    handed_over = True
                    return ASYNC
                if queued is not None:
                    return queued
            with lm.exec_lock:
                t_exec = time.perf_counter_ns()
                result = model.execute(inputs, ctx)
                t2 = time.perf_counter_ns()
            if model.decoupled:
                handed_over = True  # released when the response generator ends
                return self._decoupled_iter(lm, req, result, requested, out_specs, version, stats, t_arrive)
            response = self._build_response(lm, req, result, requested, out_specs, version, batch)
            t3 = time.perf_counter_ns()
            stats.record(batch, (t0 - t_arrive) + (t_exec - t1), t1 - t0, t2 - t_exec, t3 - t2)
            return response
        except ModelError as e:
            stats.record_fail(time.perf_counter_ns() - t_arrive)
            raise ServerError(str(e), 400) from None
        except ServerError:
            stats.record_fail(time.perf_counter_ns() - t_arrive)
            raise
        finally:
            if not handed_over:
                self.release_request(req)

    def _run_device_bytes(self, lm, req, inputs, ctx, requested, out_specs, version, batch, stats, t_arrive, t0, t1,
                          on_done):
        """BYTES tensors that sit in device shared memory, for a model with a device path: the
        request is queued for the model's batch thread.  With ``on_done`` the answer is delivered
        from there (returns :data:`ASYNC`); without, this thread waits for it.  Returns None when
        the request does not qualify (the caller then runs the host path)."""
        DeviceBytes = _device_bytes_class()
        if not any(type(v) is DeviceBytes for v in inputs.values()):
            return None
        for ro in requested:
            if ro.shm is None or ro.class_count or out_specs[ro.name].datatype != "BYTES":
                continue
            entry = self.devshm.get(ro.shm.region)
            if entry is None or entry[2].host_backed:
                continue
            if ro.shm.offset + ro.shm.byte_size > entry[1]:
                raise ServerError(f"Invalid offset + byte size for shared memory region: '{ro.shm.region}'", 400)
            ctx.output_destinations[ro.name] = DeviceBytes(entry[2].address + ro.shm.offset, ro.shm.byte_size, (),
                                                           entry[2], ro.shm.offset, entry[0])
        waiter = threading.Event() if on_done is None else None
        box = []

        def finish(result, error):
            response = None
            try:
                if error is None:
                    response = self._build_response(lm, req, result, requested, out_specs, version, batch)
                    now = time.perf_counter_ns()
                    stats.record(batch, t0 - t_arrive, t1 - t0, now - t1, 0)
                else:
                    stats.record_fail(time.perf_counter_ns() - t_arrive)
                    error = error if isinstance(error, ServerError) else ServerError(str(error), 400)
            except ServerError as e:
                error = e
            except Exception as e:  # noqa: BLE001
                error = ServerError(f"internal error: {type(e).__name__}: {e}", 500)
            if waiter is None:
                self.release_request(req)
                on_done(response if error is None else None, error)
            else:
                box.append((response, error))
                waiter.set()

        if not lm.model.execute_async(inputs, ctx, finish):
            return None
        if waiter is None:
            return ASYNC
        waiter.wait()
        response, error = box[0]
        if error is not None:
            raise error
        return response

    # -- GPU helpers (torch imported lazily: CPU-only servers never need it) --
    def infer_tensors(self, model_name, inputs, batch):
        """A step of an ensemble: run ``inputs`` (name -> device tensor or array with a leading batch axis)
        through a loaded model the way a request goes - behind the dynamic batcher it is merged with
        whatever else is queued for that model, whoever sent it - and return ``{output name: array}``.
        Blocks the calling (worker) thread; counted in the step model's statistics."""
        lm = self._get(model_name, "")
        model = lm.model
        version = model.versions[-1]
        requested = [RequestedOutput(name=t.name) for t in model.outputs]
        ctx = ExecContext(request_id="", parameters={}, version=version, requested_outputs=[r.name for r in requested])
        ctx.stable_inputs = set()
        req = InferRequest(model_name)
        t_arrive = time.perf_counter_ns()
        if lm.batcher is not None:
            resp = lm.batcher.submit(req, inputs, int(batch), set(), requested, ctx, lm.stats[version], t_arrive, version)
        else:
            resp = self._run_gpu_single(lm, req, inputs, int(batch), set(), requested, ctx, lm.stats[version], t_arrive,
                                        version)
        return {o.name: o.data for o in resp.outputs}

    def _to_device(self, lm, inputs, stable):
        import torch

        device = torch.device("cuda", lm.device_index or 0)
        out = {}
        for name, arr in inputs.items():
            if isinstance(arr, np.ndarray):
                if arr.dtype == np.object_:
                    out[name] = arr
                    continue
                host = torch.from_numpy(np.ascontiguousarray(arr))
                out[name] = host.to(device, non_blocking=True)
            else:
                out[name] = arr
        return out

    def _run_gpu_single(self, lm, req, inputs, batch, stable, requested, ctx, stats, t_arrive, version):
        model = lm.model
        out_specs = {t.name: t for t in model.outputs}
        t0 = time.perf_counter_ns()
        # models that keep their own locks (an ensemble: its requests wait side by side for the step models'
        # batchers) run their requests concurrently
        with (contextlib.nullcontext() if getattr(model, "concurrent", False) else lm.exec_lock):
            t_exec = time.perf_counter_ns()
            dev_inputs = self._to_device(lm, inputs, stable)
            for ro in requested:
                dest = self._output_destination(lm, ro, out_specs[ro.name], batch)
                if dest is not None:
                    dev_inputs["__out__" + ro.name] = dest
            t1 = time.perf_counter_ns()
            result = model.execute(dev_inputs, ctx)
            if not getattr(model, "concurrent", False):
                model.stream.synchronize()
            t2 = time.perf_counter_ns()
        response = self._build_response(lm, req, result, requested, out_specs, version, batch, dev_inputs)
        t3 = time.perf_counter_ns()
        stats.record(batch, (t_exec - t_arrive), t1 - t_exec, t2 - t1, t3 - t2)
        return response

    def _decoupled_iter(self, lm, req, generator, requested, out_specs, version, stats, t_arrive):
        want_final = bool(req.parameters.get("triton_enable_empty_final_response", False))

        def gen():
            count = 0
            try:
                for result in generator:
                    resp = self._build_response(lm, req, result, requested, out_specs, version, 0)
                    if want_final:
                        resp.parameters["triton_final_response"] = False
                    count += 1
                    yield resp
                if want_final:
                    yield InferResponse(model_name=req.model_name, model_version=version, id=req.id,
                                        parameters={"triton_final_response": True})
                stats.record(1, 0, 0, time.perf_counter_ns() - t_arrive, 0)
            except ModelError as e:
                stats.record_fail(time.perf_counter_ns() - t_arrive)
                raise ServerError(str(e), 400) from None
            finally:
                self.release_request(req)

        return gen()

    def _build_response(self, lm, req, result, requested, out_specs, version, batch, dev_inputs=None):
        model = lm.model
        response = InferResponse(model_name=req.model_name, model_version=version, id=req.id)
        device_writes = []  # (mapping, offset, blob) of outputs bound for device regions: one native call
        for ro in requested:
            spec = out_specs[ro.name]
            if ro.name not in result:
                continue
            value = result[ro.name]
            preplaced = dev_inputs is not None and ("__out__" + ro.name) in dev_inputs and \
                value is dev_inputs["__out__" + ro.name]
            if preplaced:
                response.outputs.append(OutputTensor(ro.name, spec.datatype, list(value.shape), None, ro.shm))
                continue
            if type(value).__name__ == "DeviceBytesResult":  # written into its device destination by the model
                response.outputs.append(OutputTensor(ro.name, spec.datatype, list(value.shape), None, ro.shm))
                continue
            datatype = spec.datatype
            if ro.class_count:
                value = self._classify(lm, value, ro.class_count)
                datatype = "BYTES"
            elif not isinstance(value, np.ndarray):
                value = self._to_host(value)
            if datatype not in ("BYTES", "BF16"):
                want = triton_to_np_dtype(datatype)
                if value.dtype != want:
                    value = value.astype(want)
            shape = list(value.shape)
            if ro.shm is not None:
                self._write_shm(ro, datatype, value, device_writes)
                response.outputs.append(OutputTensor(ro.name, datatype, shape, None, ro.shm))
            else:
                response.outputs.append(OutputTensor(ro.name, datatype, shape, value, None, ro.binary))
        if device_writes:
            from ..utils import hip_shared_memory as hipshm

            hipshm.write_mappings(device_writes)
        return response

    def _to_host(self, value):
        return value.detach().cpu().numpy()

    def _classify(self, lm, value, k):
        labels = lm.model.labels
        if not isinstance(value, np.ndarray):
            from ..ops import classification

            table = lm.label_table
            if table is None and labels:
                table = lm.label_table = classification.LabelTable(labels, value.device)
            blob, shape = classification.classification_payload(value, k, table)
            return PackedBytes(blob, shape)
        return format_classification(value, labels, k)

    def _write_shm(self, ro: RequestedOutput, datatype, value, device_writes=None):
        ref = ro.shm
        blob = encode_wire(datatype, value)
        if len(blob) > ref.byte_size:
            raise ServerError(
                f"shared memory size specified with the request for output '{ro.name}' ({ref.byte_size} bytes) "
                f"should be at least {len(blob)} bytes to hold the results")
        view = self.sysshm.view(ShmRef(ref.region, len(blob), ref.offset))
        if view is not None:
            view[:] = blob
            return
        entry = self.devshm.get(ref.region)
        if entry is None:
            raise ServerError(f"Unable to find shared memory region: '{ref.region}'", 400)
        from ..utils import hip_shared_memory as hipshm

        if ref.offset + len(blob) > entry[1]:
            raise ServerError(f"Invalid offset + byte size for shared memory region: '{ref.region}'", 400)
        if device_writes is not None:
            device_writes.append((entry[2], ref.offset, blob))
        else:
            hipshm.write_mapping(entry[2], ref.offset, blob)

    def shutdown(self):
        with self._lock:
            loaded = list(self._loaded.values())
            self._loaded.clear()
        for lm in loaded:
            self._retire(lm)
        self.sysshm.unregister()
        try:
            self.devshm.unregister()
        except Exception:
            pass
        pipe = getattr(self, "_native_pipe", None)
        if pipe is not None:
            self.native.disable_device_bytes()  # its batch thread uses the pipeline released below
            self.devshm.native = None
            self._native_pipe = None
            pipe[0].close()
            pipe[1].close()

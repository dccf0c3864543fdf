"""CPU reference models the client examples and tests talk to.

Contracts (reference examples, ``src/python/examples``):

=============================  ==============================================
``simple``                     INT32 add / sub of two ``[batch,16]`` tensors
``simple_string``              same on decimal strings (BYTES)
``simple_identity``            BYTES identity (embedded NULs survive)
``custom_identity_int32``      INT32 identity; ``execute_delay_ms`` parameter
``onnx_int32_int32_int32``     add/sub, versions 2 and 3 swap the outputs
``simple_sequence`` & co.      stateful sequence models
``repeat_int32``               decoupled: one response per input element
=============================  ==============================================
"""

from __future__ import annotations

import math
import os
import time

import numpy as np

from .base import Model, ModelError, TensorSpec, register


def _np_bytes(values, shape):
    out = np.empty(len(values), dtype=np.object_)
    for i, v in enumerate(values):
        out[i] = v
    return out.reshape(shape)


@register
class Simple(Model):
    name = "simple"
    inline = True  # microseconds of host work: transports may run it on their own thread
    native_kernel = "add_sub"  # the fronts run ordinary requests in C++ (tca/native_executor.h)
    max_batch_size = 8
    inputs = [TensorSpec("INPUT0", "INT32", [16]), TensorSpec("INPUT1", "INT32", [16])]
    outputs = [TensorSpec("OUTPUT0", "INT32", [16]), TensorSpec("OUTPUT1", "INT32", [16])]

    def execute(self, inputs, ctx):
        a, b = inputs["INPUT0"], inputs["INPUT1"]
        return {"OUTPUT0": a + b, "OUTPUT1": a - b}


@register
class SimpleInt8(Model):
    """INT8 flavour of ``simple`` (reference
    ``src/python/examples/grpc_explicit_int8_content_client.py:59``)."""

    name = "simple_int8"
    inline = True  # microseconds of host work: transports may run it on their own thread
    native_kernel = "add_sub"
    max_batch_size = 8
    inputs = [TensorSpec("INPUT0", "INT8", [16]), TensorSpec("INPUT1", "INT8", [16])]
    outputs = [TensorSpec("OUTPUT0", "INT8", [16]), TensorSpec("OUTPUT1", "INT8", [16])]

    def execute(self, inputs, ctx):
        a, b = inputs["INPUT0"], inputs["INPUT1"]
        return {"OUTPUT0": (a + b).astype(np.int8), "OUTPUT1": (a - b).astype(np.int8)}


@register
class VersionedAddSub(Model):
    """Three versions; 2 and 3 return (sub, add) instead of (add, sub)
    (reference ``src/c++/tests/cc_client_test.cc:436-440``)."""

    name = "onnx_int32_int32_int32"
    platform = "onnxruntime_onnx"
    versions = ("1", "2", "3")
    max_batch_size = 8
    inputs = [TensorSpec("INPUT0", "INT32", [16]), TensorSpec("INPUT1", "INT32", [16])]
    outputs = [TensorSpec("OUTPUT0", "INT32", [16]), TensorSpec("OUTPUT1", "INT32", [16])]

    def execute(self, inputs, ctx):
        a, b = inputs["INPUT0"], inputs["INPUT1"]
        if ctx.version in ("2", "3"):
            return {"OUTPUT0": a - b, "OUTPUT1": a + b}
        return {"OUTPUT0": a + b, "OUTPUT1": a - b}


class _DeviceBytesPath:
    """BYTES models that can serve a request without its tensors leaving the GPU.

    When every input of a request is a :class:`~triton_client_amd.ops.bytes_pipe.DeviceBytes`
    (a range of a device shared-memory region) and every output has a device destination, the
    request joins a queue that one thread drains in batches: prefix walk, the model's operation
    and the packing into the callers' regions run as three kernel launches per batch
    (``csrc/hip/bytes_models.hip``).  Any other request takes the host path of ``execute``."""

    device_bytes = True
    device_op = None          # bytes_pipe.OP_*
    device_outputs = ()       # output names in kernel order

    def _device_job(self, inputs, ctx):
        """(count, input tensors, destinations) when the request qualifies for the device path."""
        dests = ctx.output_destinations
        if not dests:
            return None
        ins = [inputs.get(t.name) for t in self.inputs]
        first = ins[0] if ins else None
        if first is None:
            return None
        DeviceBytes, limit = _device_bytes_types()
        count = math.prod(first.shape) if type(first) is DeviceBytes else 0
        if count <= 0 or count > 4096:
            return None
        for t in ins:
            if (type(t) is not DeviceBytes or t.nbytes > limit or t.device != first.device
                    or math.prod(t.shape) != count):
                return None
        wanted = ctx.requested_outputs
        if wanted is not None and (len(wanted) != len(self.device_outputs) or set(wanted) != set(self.device_outputs)):
            return None
        try:
            return count, ins, [dests[n] for n in self.device_outputs]
        except KeyError:
            return None

    def execute_async(self, inputs, ctx, callback):
        """Queue the request on the device path; ``callback(result, error)`` runs on the batch thread.
        Returns False (nothing queued) when the request does not qualify."""
        job = self._device_job(inputs, ctx)
        if job is None:
            return False
        self._ensure_worker(job[1][0].device)
        self._jobs.put((job, callback))
        return True

    def _ensure_worker(self, device):
        import queue
        import threading

        if getattr(self, "_jobs", None) is None:
            with _WORKER_LOCK:
                if getattr(self, "_jobs", None) is None:
                    self._pipes = {}
                    jobs = queue.Queue()
                    target = lambda: self._drain(jobs)  # noqa: E731
                    if os.environ.get("TCA_PROFILE_DIR"):  # developer aid: per-request Python cost of the batch thread
                        from ..server.batcher import _profiled

                        target = _profiled(target, f"bytes-{self.name}")
                    threading.Thread(target=target, name=f"tca-bytes-{self.name}", daemon=True).start()
                    self._jobs = jobs

    def _drain(self, jobs):
        from ..ops import _device
        from ..ops.bytes_pipe import STATUS_TEXT, BytesPipe, DeviceBytesResult

        max_batch = 64
        streams = {}
        while True:
            first = jobs.get()
            if first is None:
                return
            batch = [first]
            while len(batch) < max_batch:
                try:
                    nxt = jobs.get_nowait()
                except Exception:
                    break
                if nxt is None:
                    jobs.put(None)
                    break
                batch.append(nxt)
            groups = {}
            for item in batch:
                (count, ins, _), _ = item
                groups.setdefault((ins[0].device, count), []).append(item)
            for (device, count), items in groups.items():
                try:
                    pipe = self._pipes.get(device)
                    if pipe is None:
                        pipe = self._pipes[device] = BytesPipe(device, max_batch, 4096)
                        streams[device] = _device.Stream(device)
                    in_ptrs, in_bytes, out_ptrs, out_caps = [], [], [], []
                    for (_, ins, dests), _ in items:
                        for t in ins:
                            in_ptrs.append(t.address)
                            in_bytes.append(t.nbytes)
                        for d in dests:
                            out_ptrs.append(d.address)
                            out_caps.append(d.nbytes)
                    codes, produced = pipe.run(self.device_op, count, in_ptrs, in_bytes, out_ptrs, out_caps,
                                               streams[device].handle)
                    for ((_, ins, _), callback), code, sizes in zip(items, codes, produced):
                        if code:
                            callback(None,
                                     ModelError(f"{self.name}: {STATUS_TEXT.get(code, 'device BYTES path failed')}"))
                        else:
                            callback({name: DeviceBytesResult(ins[0].shape, size)
                                      for name, size in zip(self.device_outputs, sizes)}, None)
                except Exception as e:  # noqa: BLE001 - every caller of the group gets the failure
                    for _, callback in items:
                        callback(None, ModelError(f"{self.name}: device BYTES path failed: {e}"))

    def _host_inputs(self, inputs):
        """Inputs for the host path: device-resident BYTES tensors are fetched and parsed."""
        from ..ops.bytes_pipe import DeviceBytes

        return {k: (v.to_host() if isinstance(v, DeviceBytes) else v) for k, v in inputs.items()}

    def unload(self):
        jobs = getattr(self, "_jobs", None)
        if jobs is not None:
            jobs.put(None)
            self._jobs = None


import threading as _threading  # noqa: E402

_WORKER_LOCK = _threading.Lock()
_DEVICE_BYTES_TYPES = None


def _device_bytes_types():
    """(DeviceBytes class, largest stream the kernels take), resolved once."""
    global _DEVICE_BYTES_TYPES
    if _DEVICE_BYTES_TYPES is None:
        from ..ops.bytes_pipe import DeviceBytes, max_stream_bytes

        _DEVICE_BYTES_TYPES = (DeviceBytes, max_stream_bytes())
    return _DEVICE_BYTES_TYPES


@register
class SimpleString(_DeviceBytesPath, Model):
    name = "simple_string"
    inline = True  # microseconds of host work: transports may run it on their own thread
    max_batch_size = 8
    inputs = [TensorSpec("INPUT0", "BYTES", [16]), TensorSpec("INPUT1", "BYTES", [16])]
    outputs = [TensorSpec("OUTPUT0", "BYTES", [16]), TensorSpec("OUTPUT1", "BYTES", [16])]
    device_op = 0
    device_outputs = ("OUTPUT0", "OUTPUT1")
    native_kernel = "bytes_add_sub"  # device shared memory only; anything else runs below

    def execute(self, inputs, ctx):
        inputs = self._host_inputs(inputs)
        try:
            a = np.array([int(x) for x in inputs["INPUT0"].reshape(-1)], dtype=np.int64)
            b = np.array([int(x) for x in inputs["INPUT1"].reshape(-1)], dtype=np.int64)
        except ValueError as e:
            raise ModelError(f"simple_string expects decimal integer strings: {e}") from None
        shape = inputs["INPUT0"].shape
        return {
            "OUTPUT0": _np_bytes([str(v).encode() for v in (a + b)], shape),
            "OUTPUT1": _np_bytes([str(v).encode() for v in (a - b)], shape),
        }


@register
class SimpleIdentity(_DeviceBytesPath, Model):
    name = "simple_identity"
    inline = True  # microseconds of host work: transports may run it on their own thread
    max_batch_size = 8
    inputs = [TensorSpec("INPUT0", "BYTES", [-1])]
    outputs = [TensorSpec("OUTPUT0", "BYTES", [-1])]
    device_op = 1
    device_outputs = ("OUTPUT0",)
    native_kernel = "bytes_identity"  # device shared memory only; anything else runs below

    def execute(self, inputs, ctx):
        return {"OUTPUT0": self._host_inputs(inputs)["INPUT0"]}


@register
class CustomIdentityInt32(Model):
    """Identity with an optional server side delay (``execute_delay_ms``
    parameter) – what the timeout tests rely on
    (``src/c++/tests/client_timeout_test.cc:421``)."""

    name = "custom_identity_int32"
    max_batch_size = 1024
    native_kernel = "identity"  # requests without parameters (no delay asked for) are answered in C++
    inputs = [TensorSpec("INPUT0", "INT32", [-1])]
    outputs = [TensorSpec("OUTPUT0", "INT32", [-1])]

    def execute(self, inputs, ctx):
        delay = float(self.parameters.get("execute_delay_ms", 0) or 0)
        delay = float(ctx.parameters.get("execute_delay_ms", delay))
        if delay > 0:
            time.sleep(delay / 1000.0)
        return {"OUTPUT0": inputs["INPUT0"]}


class _SequenceBase(Model):
    sequence = True
    max_batch_size = 1
    inputs = [TensorSpec("INPUT", "INT32", [1])]
    outputs = [TensorSpec("OUTPUT", "INT32", [1])]


@register
class SimpleSequence(_SequenceBase):
    """``OUTPUT = INPUT`` (+1 on the first request of a sequence)."""

    name = "simple_sequence"
    native_kernel = "sequence"  # numeric correlation ids: answered in C++, which also keeps the table of open sequences

    def execute(self, inputs, ctx):
        value = inputs["INPUT"].astype(np.int32)
        if ctx.sequence is not None and ctx.sequence.start:
            value = value + 1
        return {"OUTPUT": value}


def _corrid_as_int(sequence_id):
    try:
        return int(sequence_id)
    except (TypeError, ValueError):
        raise ModelError("dyna sequence models require a correlation id that decodes to an integer") from None


@register
class SimpleDynaSequence(_SequenceBase):
    """Like ``simple_sequence`` and adds the correlation id on sequence end."""

    name = "simple_dyna_sequence"
    native_kernel = "dyna_sequence"

    def execute(self, inputs, ctx):
        value = inputs["INPUT"].astype(np.int32)
        seq = ctx.sequence
        if seq is not None:
            if seq.start:
                value = value + 1
            if seq.end:
                value = value + np.int32(_corrid_as_int(seq.sequence_id))
        return {"OUTPUT": value}


@register
class SimpleStringDynaSequence(_SequenceBase):
    """String correlation ids; accumulates inputs over the sequence and adds
    the correlation id on sequence end."""

    name = "simple_string_dyna_sequence"

    def execute(self, inputs, ctx):
        value = inputs["INPUT"].astype(np.int32)
        seq = ctx.sequence
        if seq is not None:
            acc = seq.state.get("acc")
            if acc is not None and not seq.start:
                value = value + acc
            seq.state["acc"] = value.copy()
            if seq.end:
                value = value + np.int32(_corrid_as_int(seq.sequence_id))
        return {"OUTPUT": value}


@register
class RepeatInt32(Model):
    """Decoupled: for ``IN[n]`` emit n responses ``OUT=[IN[i]]``, ``IDX=[i]``,
    sleeping ``DELAY[i]`` ms before each and ``WAIT`` ms after the last."""

    name = "repeat_int32"
    decoupled = True
    native_kernel = "repeat"  # stream requests without delays are answered in C++ (tca/native_executor.h)
    inputs = [TensorSpec("IN", "INT32", [-1]), TensorSpec("DELAY", "UINT32", [-1]), TensorSpec("WAIT", "UINT32", [1])]
    outputs = [TensorSpec("OUT", "INT32", [1]), TensorSpec("IDX", "UINT32", [1])]

    def execute(self, inputs, ctx):
        values = inputs["IN"].reshape(-1)
        delays = inputs["DELAY"].reshape(-1)
        if len(values) != len(delays):
            raise ModelError("IN and DELAY must have the same number of elements")
        wait = int(inputs["WAIT"].reshape(-1)[0]) if "WAIT" in inputs else 0

        def responses():
            for i, v in enumerate(values):
                if delays[i]:
                    time.sleep(float(delays[i]) / 1000.0)
                yield {"OUT": np.array([v], dtype=np.int32), "IDX": np.array([i], dtype=np.uint32)}
            if wait:
                time.sleep(wait / 1000.0)

        return responses()


def _tiny_logits(x_nchw):
    """Deterministic 10-class "classifier": class scores are weighted channel
    means, so results are easy to verify from the client side."""
    means = x_nchw.reshape(x_nchw.shape[0], x_nchw.shape[1], -1).mean(axis=2).astype(np.float32)  # [B, C]
    c = means.shape[1]
    weights = (np.arange(10, dtype=np.float32)[:, None] + 1.0) * (np.arange(c, dtype=np.float32)[None, :] - 0.5)
    return means @ weights.T  # [B, 10]


_TINY_LABELS = ["tench", "goldfish", "shark", "ray", "rooster", "hen", "ostrich", "finch", "robin", "owl"]


@register
class TinyClassifier(Model):
    """CPU stand-in for an image classifier (NCHW FP32 in, FP32 vector out) so
    that ``image_client`` runs without a GPU.  Contract as the reference's
    ``image_client.py:59-150`` expects: one 3-dim input + batch, one FP32
    vector output, labels for the classification extension."""

    name = "tiny_classifier"
    max_batch_size = 16
    inputs = [TensorSpec("input", "FP32", [3, 224, 224], format="FORMAT_NCHW")]
    outputs = [TensorSpec("output", "FP32", [10], label_filename="labels.txt")]
    labels = _TINY_LABELS

    def execute(self, inputs, ctx):
        return {"output": _tiny_logits(inputs["input"])}


@register
class TinyClassifierNhwc(Model):
    name = "tiny_classifier_nhwc"
    max_batch_size = 16
    inputs = [TensorSpec("input", "FP32", [224, 224, 3], format="FORMAT_NHWC")]
    outputs = [TensorSpec("output", "FP32", [10], label_filename="labels.txt")]
    labels = _TINY_LABELS

    def execute(self, inputs, ctx):
        return {"output": _tiny_logits(np.transpose(inputs["input"], (0, 3, 1, 2)))}


@register
class PreprocessEnsemble(Model):
    """Ensemble taking encoded image files (JPEG/PNG/PPM bytes) as a BYTES
    tensor: decode -> resize 224 + INCEPTION scaling -> ``tiny_classifier``.
    The reference's ``ensemble_image_client`` talks to such a model
    (``preprocess_inception_ensemble``, server side DALI preprocessing)."""

    name = "preprocess_inception_ensemble"
    platform = "ensemble"
    max_batch_size = 16
    inputs = [TensorSpec("INPUT", "BYTES", [1])]
    outputs = [TensorSpec("OUTPUT", "FP32", [10], label_filename="labels.txt")]
    labels = _TINY_LABELS

    def execute(self, inputs, ctx):
        import io

        from PIL import Image

        from ..ops.preprocess_ref import reference_preprocess

        blobs = inputs["INPUT"].reshape(-1)
        batch = []
        for blob in blobs:
            try:
                img = Image.open(io.BytesIO(blob))
                img.load()
            except Exception as e:
                raise ModelError(f"unable to decode image: {e}") from None
            batch.append(reference_preprocess(np.asarray(img.convert("RGB")), 224, 224, 3, "NCHW", np.float32,
                                              "INCEPTION", "pil"))
        return {"OUTPUT": _tiny_logits(np.stack(batch))}

    def config(self):
        cfg = super().config()
        cfg["ensemble_scheduling"] = {"step": [
            {"model_name": "image_preprocess", "model_version": -1, "input_map": {"INPUT": "INPUT"},
             "output_map": {"OUTPUT": "preprocessed"}},
            {"model_name": "tiny_classifier", "model_version": -1, "input_map": {"input": "preprocessed"},
             "output_map": {"output": "OUTPUT"}}]}
        return cfg

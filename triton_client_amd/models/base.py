"""Model abstraction served by the in-repo v2 server.

A model declares its tensors (:class:`TensorSpec`) and implements
``execute``.  Three execution styles exist:

* plain           ``execute(inputs, ctx) -> {name: array}``
* sequence        same, ``ctx.sequence`` carries id/start/end and a per
                  sequence ``state`` dict that persists until ``end``
* decoupled       ``execute`` is a generator yielding ``{name: array}``; each
                  item becomes one streamed response

``device == "gpu"`` models receive and return ``torch`` tensors living on
their GPU; ``"cpu"`` models work on numpy arrays.  The contracts of the named
models come from the reference's examples and tests (e.g. ``simple``:
``src/python/examples/simple_http_infer_client.py:207-262``).
"""

from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional

_DT_CONFIG = {
    "BOOL": "TYPE_BOOL", "UINT8": "TYPE_UINT8", "UINT16": "TYPE_UINT16", "UINT32": "TYPE_UINT32",
    "UINT64": "TYPE_UINT64", "INT8": "TYPE_INT8", "INT16": "TYPE_INT16", "INT32": "TYPE_INT32",
    "INT64": "TYPE_INT64", "FP16": "TYPE_FP16", "FP32": "TYPE_FP32", "FP64": "TYPE_FP64",
    "BYTES": "TYPE_STRING", "BF16": "TYPE_BF16",
}


@dataclass
class TensorSpec:
    name: str
    datatype: str
    dims: List[int]  # without the batch dimension when max_batch_size > 0
    format: str = "FORMAT_NONE"  # inputs only
    label_filename: str = ""  # outputs only
    optional: bool = False


@dataclass
class SequenceInfo:
    sequence_id: object = 0
    start: bool = False
    end: bool = False
    state: Optional[dict] = None


@dataclass
class ExecContext:
    request_id: str = ""
    parameters: Dict[str, object] = field(default_factory=dict)
    sequence: Optional[SequenceInfo] = None
    version: str = "1"
    requested_outputs: Optional[List[str]] = None
    stable_inputs: set = field(default_factory=set)  # inputs aliasing registered device shm
    # BYTES outputs bound for device shared memory: name -> ops.bytes_pipe.DeviceBytes of the destination range
    output_destinations: Dict[str, object] = field(default_factory=dict)


class ModelError(Exception):
    """Raised by a model for a request it cannot serve (maps to HTTP 400 /
    gRPC INVALID_ARGUMENT)."""


class Model:
    name = ""
    platform = "amd_python"
    backend = "python"
    versions = ("1",)
    max_batch_size = 0
    inputs: List[TensorSpec] = []
    outputs: List[TensorSpec] = []
    decoupled = False
    sequence = False
    device = "cpu"
    dynamic_batching = False
    labels: Optional[List[str]] = None

    def __init__(self):
        # per-instance copy so config overrides do not leak between loads
        self.max_batch_size = type(self).max_batch_size
        self.parameters = {}

    # -- lifecycle --
    def load(self, device_index=None):
        """Allocate weights / warm up.  ``device_index`` is the GPU ordinal for
        ``device == "gpu"`` models."""

    def unload(self):
        pass

    def apply_config_override(self, override: dict):
        """Apply the JSON config sent with ``load_model(config=...)``."""
        if "max_batch_size" in override:
            self.max_batch_size = int(override["max_batch_size"])
        for key, value in (override.get("parameters") or {}).items():
            self.parameters[key] = value.get("string_value") if isinstance(value, dict) else value

    # -- execution --
    def execute(self, inputs, ctx: ExecContext):
        raise NotImplementedError

    # -- description --
    def _shape_with_batch(self, dims):
        return ([-1] if self.max_batch_size > 0 else []) + list(dims)

    def metadata(self):
        return {
            "name": self.name,
            "versions": list(self.versions),
            "platform": self.platform,
            "inputs": [{"name": t.name, "datatype": t.datatype, "shape": self._shape_with_batch(t.dims)}
                       for t in self.inputs],
            "outputs": [{"name": t.name, "datatype": t.datatype, "shape": self._shape_with_batch(t.dims)}
                        for t in self.outputs],
        }

    def config(self):
        cfg = {
            "name": self.name,
            "platform": self.platform,
            "backend": self.backend,
            "version_policy": {"latest": {"num_versions": 1}},
            "max_batch_size": self.max_batch_size,
            "input": [
                {"name": t.name, "data_type": _DT_CONFIG[t.datatype], "format": t.format, "dims": list(t.dims),
                 "is_shape_tensor": False, "allow_ragged_batch": False, "optional": t.optional}
                for t in self.inputs
            ],
            "output": [
                {"name": t.name, "data_type": _DT_CONFIG[t.datatype], "dims": list(t.dims),
                 "label_filename": t.label_filename, "is_shape_tensor": False}
                for t in self.outputs
            ],
            "instance_group": [{"name": self.name + "_0", "kind": "KIND_GPU" if self.device == "gpu" else "KIND_CPU",
                                "count": int(getattr(self, "instance_count", 1))}],
            "model_transaction_policy": {"decoupled": self.decoupled},
            "parameters": {k: {"string_value": str(v)} for k, v in self.parameters.items()},
        }
        if self.sequence:
            cfg["sequence_batching"] = {"max_sequence_idle_microseconds": 5000000}
        elif self.dynamic_batching:
            cfg["dynamic_batching"] = {"preferred_batch_size": [self.max_batch_size],
                                       "max_queue_delay_microseconds": int(getattr(self, "max_queue_delay_us", 0))}
        return copy.deepcopy(cfg)


MODEL_ZOO = {}


def register(cls):
    """Class decorator adding a model to the zoo under ``cls.name``."""
    MODEL_ZOO[cls.name] = cls
    return cls

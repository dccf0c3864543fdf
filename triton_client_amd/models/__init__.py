"""Model zoo of the in-repo v2 server.

``MODEL_ZOO`` maps model name -> class.  CPU plumbing models are always
available; GPU models (``resnet50`` ...) import torch lazily when loaded.
"""

from .base import MODEL_ZOO, ExecContext, Model, ModelError, SequenceInfo, TensorSpec, register  # noqa: F401
from . import simple  # noqa: F401  (registers the CPU models)
from . import vision  # noqa: F401  (registers resnet50 & friends; torch is imported on load)
from . import sharded  # noqa: F401  (resnet50_sharded: batches split over several GPUs)

DEFAULT_CPU_MODELS = (
    "simple", "simple_int8", "simple_string", "simple_identity", "custom_identity_int32", "onnx_int32_int32_int32",
    "simple_sequence", "simple_dyna_sequence", "simple_string_dyna_sequence", "repeat_int32",
    "tiny_classifier", "tiny_classifier_nhwc", "preprocess_inception_ensemble",
)

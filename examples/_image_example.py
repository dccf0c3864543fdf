"""Shared pieces of the image classification examples: model introspection,
host- or GPU-side preprocessing, batching by wrap-around and result printing.

Two input paths:

``--preprocess host``  resize / scale on the CPU (numpy + Pillow) and send the
                       tensor on the wire – what the reference's
                       ``image_client.py:153-192`` does.
``--preprocess hip``   decode on the host, then ONE fused HIP kernel (resize +
                       scale + layout + cast) writes the batch straight into a
                       device shared-memory region the server reads in place;
                       no tensor bytes cross the wire.
"""
import os
from dataclasses import dataclass

import numpy as np
from _common import fail
from PIL import Image

from tritonclient.utils import triton_to_np_dtype


@dataclass
class ModelSpec:
    max_batch_size: int
    input_name: str
    output_name: str
    c: int
    h: int
    w: int
    layout: str  # "NCHW" | "NHWC"
    dtype: str


def _field(obj, name):
    return obj[name] if isinstance(obj, dict) else getattr(obj, name)


def parse_model(metadata, config):
    """Checks the model is an image classifier (one image input, one vector
    output) and extracts its geometry.  Accepts the JSON dicts of the HTTP
    client or the protobuf messages of the gRPC client.
    (Reference contract: ``src/python/examples/image_client.py:59-150``.)"""
    if hasattr(config, "config"):
        config = config.config
    inputs, outputs = _field(metadata, "inputs"), _field(metadata, "outputs")
    if len(inputs) != 1:
        fail(f"expecting 1 input, got {len(inputs)}")
    if len(outputs) != 1:
        fail(f"expecting 1 output, got {len(outputs)}")
    cfg_inputs = _field(config, "input")
    if len(cfg_inputs) != 1:
        fail(f"expecting 1 input in model configuration, got {len(cfg_inputs)}")
    inp, out, cfg_in = inputs[0], outputs[0], cfg_inputs[0]
    name = _field(metadata, "name")
    if _field(out, "datatype") != "FP32":
        fail(f"expecting output datatype to be FP32, model '{name}' output type is {_field(out, 'datatype')}")
    max_batch = int(_field(config, "max_batch_size") if not isinstance(config,
                                                                       dict) else config.get("max_batch_size", 0))
    batched = max_batch > 0
    out_shape = [int(d) for d in _field(out, "shape")]
    if sum(1 for d in out_shape[(1 if batched else 0):] if d > 1) > 1:
        fail("expecting model output to be a vector")
    in_shape = [int(d) for d in _field(inp, "shape")]
    if len(in_shape) != 3 + (1 if batched else 0):
        fail(f"expecting input to have {3 + (1 if batched else 0)} dimensions, model '{name}' input has "
             f"{len(in_shape)}")
    fmt = _field(cfg_in, "format")
    if not isinstance(fmt, str):  # protobuf enum value: 1 = NHWC, 2 = NCHW
        fmt = {1: "FORMAT_NHWC", 2: "FORMAT_NCHW"}.get(int(fmt), "FORMAT_NONE")
    if fmt not in ("FORMAT_NCHW", "FORMAT_NHWC"):
        fail(f"unexpected input format {fmt}, expecting FORMAT_NCHW or FORMAT_NHWC")
    dims = in_shape[1:] if batched else in_shape
    if fmt == "FORMAT_NHWC":
        h, w, c = dims
    else:
        c, h, w = dims
    return ModelSpec(max_batch, _field(inp, "name"), _field(out, "name"), c, h, w, fmt[len("FORMAT_"):],
                     _field(inp, "datatype"))


def list_images(path):
    if os.path.isdir(path):
        names = sorted(os.path.join(path, f) for f in os.listdir(path) if os.path.isfile(os.path.join(path, f)))
    else:
        names = [path]
    if not names:
        fail("no input images")
    return names


def decode(filename, channels):
    img = Image.open(filename)
    return np.asarray(img.convert("L" if channels == 1 else "RGB"))


def host_preprocess(pixels, spec, scaling):
    from triton_client_amd.ops.preprocess_ref import reference_preprocess

    return reference_preprocess(pixels, spec.h, spec.w, spec.c, spec.layout, triton_to_np_dtype(spec.dtype), scaling,
                                "pil")


def batches(items, batch_size):
    """Batches of ``batch_size`` indices; wraps to the first item to fill the
    last batch (reference ``image_client.py:455-464``)."""
    out, idx, last = [], 0, False
    while not last:
        batch = []
        for _ in range(batch_size):
            batch.append(idx)
            idx = (idx + 1) % len(items)
            if idx == 0:
                last = True
        out.append(batch)
    return out


class HipInputStage:
    """Owns a HIP preprocessor and ONE device shared-memory region cut into
    ``depth`` batch-sized slots.  ``fill`` runs the fused kernel with a free slot
    as destination, so batch k+1 is preprocessed while the server still reads
    batch k; the caller hands a slot back with ``release`` once the response
    of the request that used it has arrived."""

    def __init__(self, client, spec, scaling, batch_size, device=0, name="image_input", depth=4):
        import tritonclient.utils.cuda_shared_memory as hipshm
        from triton_client_amd.ops.preprocess import Preprocessor

        self._hipshm, self._client, self._name = hipshm, client, name
        self.pre = Preprocessor(spec.h, spec.w, spec.c, spec.layout, spec.dtype, scaling, "pil", device=device,
                                max_batch=batch_size)
        self.depth = max(1, int(depth))
        self.slot_bytes = self.pre.image_bytes * batch_size
        self.nbytes = self.slot_bytes * self.depth
        self._free = list(range(self.depth))
        self.region = hipshm.create_shared_memory_region(name, self.nbytes, device)
        client.register_cuda_shared_memory(name, hipshm.get_raw_handle(self.region), device, self.nbytes)

    def has_free_slot(self):
        return bool(self._free)

    def fill(self, images):
        """Preprocesses ``images`` into a free slot; returns (region name, byte size, offset, slot)."""
        if not self._free:
            raise RuntimeError("every slot of the input ring is still in flight")
        slot = self._free.pop(0)
        offset = slot * self.slot_bytes
        self.pre(images, self.region, out_offset=offset)  # synchronises: the server reads from another process
        return self._name, self.pre.image_bytes * len(images), offset, slot

    def release(self, slot):
        self._free.append(slot)

    def close(self):
        self._client.unregister_cuda_shared_memory(self._name)
        self._hipshm.destroy_shared_memory_region(self.region)
        self.pre.close()


def print_classes(result, output_name, batch_size, batched, request_id, names):
    array = result.as_numpy(output_name)
    if batched and len(array) != batch_size:
        fail(f"expected {batch_size} results, got {len(array)}")
    print(f"Request {request_id}, batch size {batch_size}")
    rows = array if batched else [array]
    for row, filename in zip(rows, names):
        print(f"Image '{filename}':")
        for entry in np.asarray(row).reshape(-1):
            text = entry.decode("utf-8") if isinstance(entry, bytes) else str(entry)
            parts = text.split(":")
            label = parts[2] if len(parts) > 2 else ""
            print(f"    {float(parts[0]):f} ({parts[1]}) = {label}")

"""Fused GPU image preprocessing (resize + colour convert + normalise + layout
+ cast) – the MI355X replacement for the CPU loop of the reference's
``image_client`` (``src/python/examples/image_client.py:153-192``).

``Preprocessor`` owns the device-side state (coefficient tables per source
size, descriptor and staging buffers, a HIP stream) and writes its output
directly into the destination you hand it: a torch tensor, a raw device
address or a hipshm region at ``batch_index * image_bytes`` – so the batch
"stack" of the reference (``np.stack``, ``image_client.py:459-469``) is just an
offset.

Resize flavours
---------------
``"pil"``  Pillow's ``Image.resize(size, BILINEAR)``: support-scaled triangle
           filter, two passes with 8-bit intermediate, 22-bit fixed-point
           weights.  The tables are built exactly like Pillow's
           ``precompute_coeffs`` so the result is bit-identical before scaling.
``"cv"``   OpenCV ``cv::resize`` default (INTER_LINEAR, half-pixel centres, 2
           taps), the flavour of the C++ ``image_client``.

Scaling modes (``image_client.py:173-181``): ``NONE``, ``INCEPTION``
(``x/127.5 - 1``), ``VGG`` (``x - (123,117,104)``, or ``x - 128`` for one
channel) plus a general ``(mean, std)`` per channel form.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

from ..utils import trace as _trace
from . import _bind
from ._device import DeviceBuffer, PinnedBuffer, Stream, as_pointer, check, lib, stream_handle

from .preprocess_ref import (  # noqa: F401  (re-exported)
    PRECISION_BITS,
    _tables,
    cv_bilinear_coeffs,
    pil_bilinear_coeffs,
    reference_preprocess,
    scaling_to_affine,
)

TILE_W = 64
_LDS_BUDGET = 48 * 1024
_LDS_LIMIT = 64 * 1024  # dynamic LDS a block may ask for without opting in to more
_STAGE_SOURCE = os.environ.get("TCA_PREPROCESS_STAGE", "1") != "0"


# ---------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------
class Preprocessor:
    """Batched fused preprocessing onto GPU ``device``.

    Parameters
    ----------
    out_h, out_w : int
    out_c : 1 or 3
    layout : "NCHW" | "NHWC"
    dtype : protocol datatype of the output ("FP16", "FP32", "INT8", ...)
    scaling : "NONE" | "INCEPTION" | "VGG" | (mean, std)
    resize : "pil" | "cv"
    device : GPU ordinal
    max_batch, max_image_bytes : sizes of the persistent staging buffers
    """

    def __init__(self, out_h, out_w, out_c=3, layout="NCHW", dtype="FP16", scaling="NONE", resize="pil", device=0,
                 max_batch=64, max_image_bytes=64 << 20):
        if out_c not in (1, 3):
            raise ValueError("out_c must be 1 or 3")
        if dtype not in _bind.DTYPE_CODE:
            raise ValueError(f"unsupported output datatype {dtype}")
        self.out_h, self.out_w, self.out_c = int(out_h), int(out_w), int(out_c)
        self.nhwc = layout == "NHWC"
        self.dtype = dtype
        self.resize = resize
        self.device = device
        self.scale, self.bias = scaling_to_affine(scaling, out_c)
        self.max_batch = max_batch
        self.image_elems = self.out_h * self.out_w * self.out_c
        self.image_bytes = self.image_elems * _bind.DTYPE_SIZE[dtype]
        self._stream = Stream(device)
        self._pinned = PinnedBuffer(max_image_bytes)
        self._dev_src = DeviceBuffer(max_image_bytes, device)
        self._desc_host = (_bind.PreprocImage * max_batch)()
        self._desc_dev = DeviceBuffer(ctypes.sizeof(self._desc_host), device)
        self._table_cache = {}  # (h, w) -> (xb_off, yb_off, xk_off, yk_off, kx, ky, tile_h, max_rows, span_px)
        self._bounds_host = np.zeros((0, 2), dtype=np.int32)
        self._coeffs_host = np.zeros((0,), dtype=np.int32)
        self._bounds_dev = None
        self._coeffs_dev = None
        self._tables_dirty = False

    # -- tables --
    def _tile_plan(self, yb):
        """Pick the tallest tile whose source-row span fits the LDS budget."""
        for tile_h in (16, 8, 4, 2, 1):
            worst = 0
            for oy0 in range(0, self.out_h, tile_h):
                last = min(oy0 + tile_h, self.out_h) - 1
                worst = max(worst, int(yb[last, 0] + yb[last, 1] - yb[oy0, 0]))
            if worst * (self.out_c * TILE_W + 4) <= _LDS_BUDGET:
                return tile_h, worst
        raise ValueError("down-scaling factor too large for the LDS tile (source rows per output row exceed budget)")

    def _entry(self, h, w):
        key = (h, w)
        entry = self._table_cache.get(key)
        if entry is not None:
            return entry
        xb, xk, kx = _tables(self.resize, w, self.out_w)
        yb, yk, ky = _tables(self.resize, h, self.out_h)
        tile_h, max_rows = self._tile_plan(yb)
        # widest run of source columns any 64-output tile draws from (sizes the staged source rows in LDS)
        span_px = 0
        for ox0 in range(0, self.out_w, TILE_W):
            last = min(ox0 + TILE_W, self.out_w) - 1
            span_px = max(span_px, int(xb[last, 0] + xb[last, 1] - xb[ox0, 0]))
        xb_off = len(self._bounds_host)
        yb_off = xb_off + len(xb)
        xk_off = len(self._coeffs_host)
        yk_off = xk_off + xk.size
        self._bounds_host = np.concatenate([self._bounds_host, xb, yb]).astype(np.int32)
        self._coeffs_host = np.concatenate([self._coeffs_host, xk.reshape(-1), yk.reshape(-1)]).astype(np.int32)
        entry = (xb_off, yb_off, xk_off, yk_off, kx, ky, tile_h, max_rows, span_px)
        self._table_cache[key] = entry
        self._tables_dirty = True
        return entry

    def _sync_tables(self):
        if not self._tables_dirty:
            return
        self._stream.synchronize()  # nothing in flight may still read the old tables
        self._bounds_dev = DeviceBuffer(max(8, self._bounds_host.nbytes), self.device)
        self._coeffs_dev = DeviceBuffer(max(4, self._coeffs_host.nbytes), self.device)
        keep1 = self._bounds_dev.upload(self._bounds_host, self._stream)
        keep2 = self._coeffs_dev.upload(self._coeffs_host, self._stream)
        self._stream.synchronize()
        del keep1, keep2
        self._tables_dirty = False

    # -- run --
    def __call__(self, images, out, out_offset=0, bgr=False, stream=None, synchronize=True):
        with _trace.range(f"hip preprocess n={len(images)}"):
            return self._run(images, out, out_offset, bgr, stream, synchronize)

    def _run(self, images, out, out_offset=0, bgr=False, stream=None, synchronize=True):
        """Preprocess ``images`` into ``out``.

        images : list of HWC / HW uint8 numpy arrays (host), or of CUDA uint8
                 torch tensors already on this GPU
        out    : torch tensor, hipshm region, ``DeviceBuffer`` or int address;
                 entry ``i`` lands at ``out_offset + i * image_bytes``
        """
        n = len(images)
        if n == 0:
            return
        if n > self.max_batch:
            raise ValueError(f"batch of {n} exceeds max_batch={self.max_batch}")
        use_stream = self._stream if stream is None else stream
        sh = stream_handle(use_stream)
        L = lib()
        cursor = 0
        tile_h, span_bytes = 16, 0
        host_chunks = []
        for i, im in enumerate(images):
            on_device = hasattr(im, "data_ptr")
            shape = tuple(im.shape)
            h, w = int(shape[0]), int(shape[1])
            c = 1 if len(shape) == 2 else int(shape[2])
            if c not in (1, 3, 4):
                raise ValueError(f"image {i}: unsupported channel count {c}")
            xb_off, yb_off, xk_off, yk_off, kx, ky, th, mr, span_px = self._entry(h, w)
            tile_h = min(tile_h, th)
            span_bytes = max(span_bytes, span_px * c)
            d = self._desc_host[i]
            if on_device:
                if not im.is_contiguous():
                    raise ValueError("device images must be contiguous HWC uint8 tensors")
                d.src = im.data_ptr()
            else:
                arr = np.ascontiguousarray(im, dtype=np.uint8)
                nbytes = arr.nbytes
                aligned = (cursor + 15) & ~15
                if aligned + nbytes > self._pinned.nbytes:
                    raise ValueError("batch does not fit the staging buffer; raise max_image_bytes")
                self._pinned.array[aligned: aligned + nbytes] = arr.reshape(-1)
                d.src = self._dev_src.ptr + aligned
                host_chunks.append((aligned, nbytes))
                cursor = aligned + nbytes
            d.height, d.width, d.stride, d.channels = h, w, w * c, c
            d.bgr = 1 if (bgr and c >= 3) else 0
            d.kx, d.ky = kx, ky
            d.xb_off, d.yb_off, d.xk_off, d.yk_off = xb_off, yb_off, xk_off, yk_off
        # tile height must suit every image of the batch: recompute rows for the chosen height
        max_rows = self._rows_for([(int(im.shape[0]), int(im.shape[1])) for im in images], tile_h)
        self._sync_tables()
        if cursor:
            check(L.tca_hip_memcpy_async(self._dev_src.ptr, self._pinned.ptr, cursor, 1, sh), "image upload")
        check(L.tca_hip_memcpy_async(self._desc_dev.ptr, ctypes.addressof(self._desc_host),
                                     ctypes.sizeof(_bind.PreprocImage) * n, 1, sh), "descriptor upload")
        prm = self._params(tile_h, max_rows, span_bytes)
        rc = L.tca_preprocess_launch(self._desc_dev.ptr, self._bounds_dev.ptr, self._coeffs_dev.ptr,
                                     as_pointer(out, out_offset), ctypes.byref(prm), n,
                                     _bind.DTYPE_CODE[self.dtype], sh)
        check(rc, "tca_preprocess_launch")
        if synchronize:
            check(L.tca_hip_stream_sync(sh), "hipStreamSynchronize")

    def _params(self, tile_h, max_rows, span_bytes):
        prm = _bind.PreprocParams()
        prm.out_h, prm.out_w, prm.out_c = self.out_h, self.out_w, self.out_c
        prm.nhwc = 1 if self.nhwc else 0
        prm.tile_h, prm.max_rows = tile_h, max_rows
        # staged source rows: widest tile's bytes in whole dwords; an odd dword pitch keeps the rows of one
        # byte column on different LDS banks
        pitch = ((span_bytes + 3) // 4) | 1
        if _STAGE_SOURCE and max_rows * (self.out_c * TILE_W + 4 + pitch * 4) <= _LDS_LIMIT:
            prm.src_pitch = pitch * 4
        else:
            prm.src_pitch = 0
        prm.image_stride = self.image_elems
        for c in range(4):
            prm.scale[c] = self.scale[c]
            prm.bias[c] = self.bias[c]
        return prm

    def _rows_for(self, shapes, tile_h):
        """Largest source-row span of a ``tile_h`` tall tile over images of ``shapes`` [(h, w), ...]."""
        max_rows = 1
        for h, w in shapes:
            _, yb_off, _, _, _, _, th, mr, _ = self._table_cache[(h, w)]
            if th == tile_h:
                max_rows = max(max_rows, mr)
            else:
                yb = self._bounds_host[yb_off: yb_off + self.out_h]
                for oy0 in range(0, self.out_h, tile_h):
                    last = min(oy0 + tile_h, self.out_h) - 1
                    max_rows = max(max_rows, int(yb[last, 0] + yb[last, 1] - yb[oy0, 0]))
        return max_rows

    def pin(self, images, bgr=False):
        """Copies decoded ``images`` (HWC / HW uint8 numpy arrays) once into page-locked memory and
        prepares their descriptors: running the returned ``PinnedBatch`` costs one host-to-device
        copy and one launch, with no per-image work on the host."""
        return PinnedBatch(self, images, bgr)

    def run_pinned(self, batch, out, out_offset=0, stream=None, synchronize=True):
        """Preprocesses a ``PinnedBatch`` of this preprocessor into ``out`` (see ``__call__``)."""
        if batch.owner is not self:
            raise ValueError("the batch was pinned for another Preprocessor")
        sh = stream_handle(self._stream if stream is None else stream)
        L = lib()
        self._sync_tables()
        check(L.tca_hip_memcpy_async(self._dev_src.ptr, batch.pinned.ptr, batch.nbytes, 1, sh), "image upload")
        check(L.tca_hip_memcpy_async(self._desc_dev.ptr, ctypes.addressof(batch.desc), ctypes.sizeof(batch.desc), 1,
                                     sh),
              "descriptor upload")
        rc = L.tca_preprocess_launch(self._desc_dev.ptr, self._bounds_dev.ptr, self._coeffs_dev.ptr,
                                     as_pointer(out, out_offset), ctypes.byref(batch.params), batch.count,
                                     _bind.DTYPE_CODE[self.dtype], sh)
        check(rc, "tca_preprocess_launch")
        if synchronize:
            check(L.tca_hip_stream_sync(sh), "hipStreamSynchronize")

    def close(self):
        self._stream.synchronize()
        for buf in (self._pinned, self._dev_src, self._desc_dev, self._bounds_dev, self._coeffs_dev):
            if buf is not None:
                buf.close()
        self._stream.close()


class PinnedBatch:
    """Decoded images of one batch in page-locked memory plus everything a launch needs
    (descriptors with their staging addresses, tile plan).  Built by ``Preprocessor.pin``."""

    def __init__(self, owner, images, bgr=False):
        n = len(images)
        if n == 0 or n > owner.max_batch:
            raise ValueError(f"batch of {n} images (max_batch={owner.max_batch})")
        self.owner, self.count = owner, n
        self.desc = (_bind.PreprocImage * n)()
        offsets, cursor = [], 0
        for im in images:
            cursor = (cursor + 15) & ~15
            offsets.append(cursor)
            cursor += int(np.asarray(im).size)
        if cursor > owner._dev_src.nbytes:
            raise ValueError("batch does not fit the staging buffer; raise max_image_bytes")
        self.nbytes = cursor
        self.pinned = PinnedBuffer(cursor)
        tile_h, span_bytes, shapes = 16, 0, []
        for i, im in enumerate(images):
            arr = np.ascontiguousarray(im, dtype=np.uint8)
            h, w = int(arr.shape[0]), int(arr.shape[1])
            c = 1 if arr.ndim == 2 else int(arr.shape[2])
            if c not in (1, 3, 4):
                raise ValueError(f"image {i}: unsupported channel count {c}")
            xb_off, yb_off, xk_off, yk_off, kx, ky, th, _, span_px = owner._entry(h, w)
            tile_h = min(tile_h, th)
            span_bytes = max(span_bytes, span_px * c)
            shapes.append((h, w))
            self.pinned.array[offsets[i]: offsets[i] + arr.size] = arr.reshape(-1)
            d = self.desc[i]
            d.src = owner._dev_src.ptr + offsets[i]
            d.height, d.width, d.stride, d.channels = h, w, w * c, c
            d.bgr = 1 if (bgr and c >= 3) else 0
            d.kx, d.ky = kx, ky
            d.xb_off, d.yb_off, d.xk_off, d.yk_off = xb_off, yb_off, xk_off, yk_off
        self.params = owner._params(tile_h, owner._rows_for(shapes, tile_h), span_bytes)

    def close(self):
        self.pinned.close()

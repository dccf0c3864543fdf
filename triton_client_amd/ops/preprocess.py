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
import math
from functools import lru_cache

import numpy as np

from . import _bind
from ._device import DeviceBuffer, PinnedBuffer, Stream, as_pointer, check, lib, stream_handle

PRECISION_BITS = 22
TILE_W = 64
_LDS_BUDGET = 48 * 1024


# ---------------------------------------------------------------------------
# coefficient tables (host)
# ---------------------------------------------------------------------------
@lru_cache(maxsize=256)
def pil_bilinear_coeffs(in_size, out_size):
    """Windows and fixed-point weights of Pillow's BILINEAR resampling.

    Returns ``(bounds[out,2] int32 (first, count), coeffs[out,ksize] int32, ksize)``.
    """
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.empty(xmax, dtype=np.float64)
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0 else t
            k[x] = 1.0 - t if t < 1.0 else 0.0
        ww = k.sum()
        if ww != 0.0:
            k = k / ww
        bounds[xx] = (xmin, xmax)
        # Pillow: (int)(k * (1 << PRECISION_BITS) + 0.5) for k >= 0 (triangle weights are never negative)
        coeffs[xx, :xmax] = (k * (1 << PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
    return bounds, coeffs, ksize


@lru_cache(maxsize=256)
def cv_bilinear_coeffs(in_size, out_size):
    """2-tap half-pixel-centre bilinear windows (OpenCV INTER_LINEAR geometry)."""
    scale = in_size / out_size
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, 2), dtype=np.int32)
    one = 1 << PRECISION_BITS
    for xx in range(out_size):
        fx = (xx + 0.5) * scale - 0.5
        sx = int(math.floor(fx))
        fx -= sx
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= in_size - 1:
            sx, fx = max(in_size - 2, 0), 1.0 if in_size > 1 else 0.0
        if in_size == 1:
            bounds[xx] = (0, 1)
            coeffs[xx, 0] = one
            continue
        w1 = int(fx * one + 0.5)
        bounds[xx] = (sx, 2)
        coeffs[xx] = (one - w1, w1)
    return bounds, coeffs, 2


def _tables(mode, in_size, out_size):
    return pil_bilinear_coeffs(in_size, out_size) if mode == "pil" else cv_bilinear_coeffs(in_size, out_size)


def scaling_to_affine(scaling, channels):
    """``(scale[4], bias[4])`` such that ``out = pixel * scale + bias``."""
    scale = [1.0] * 4
    bias = [0.0] * 4
    if scaling is None or scaling == "NONE":
        pass
    elif scaling == "INCEPTION":
        scale = [1.0 / 127.5] * 4
        bias = [-1.0] * 4
    elif scaling == "VGG":
        means = (128.0,) if channels == 1 else (123.0, 117.0, 104.0)
        for c, m in enumerate(means):
            bias[c] = -m
    elif isinstance(scaling, (tuple, list)) and len(scaling) == 2:
        mean, std = scaling  # in [0,1] units (torchvision convention)
        for c in range(channels):
            m = mean[c] if hasattr(mean, "__len__") else mean
            s = std[c] if hasattr(std, "__len__") else std
            scale[c] = 1.0 / (255.0 * s)
            bias[c] = -m / s
    else:
        raise ValueError(f"unknown scaling mode {scaling!r}")
    return scale, bias


def reference_preprocess(image, out_h, out_w, out_c=3, layout="NCHW", np_dtype=np.float32, scaling="NONE",
                         resize="pil", bgr=False):
    """Pure numpy/Pillow rendition of the same pipeline (the test oracle).

    ``resize="pil"`` uses Pillow itself; ``"cv"`` evaluates the 2-tap tables in
    numpy.
    """
    from PIL import Image

    arr = np.asarray(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.shape[2] >= 3 and bgr:
        arr = arr[:, :, [2, 1, 0] + list(range(3, arr.shape[2]))]
    if arr.shape[2] == 1:
        pil = Image.fromarray(arr[:, :, 0], "L")
    else:
        pil = Image.fromarray(np.ascontiguousarray(arr[:, :, :3]), "RGB")
    pil = pil.convert("L" if out_c == 1 else "RGB")
    if resize == "pil":
        resized = np.asarray(pil.resize((out_w, out_h), Image.BILINEAR))
    else:
        src = np.asarray(pil).astype(np.int64)
        if src.ndim == 2:
            src = src[:, :, None]
        xb, xk, _ = cv_bilinear_coeffs(src.shape[1], out_w)
        yb, yk, _ = cv_bilinear_coeffs(src.shape[0], out_h)
        half = 1 << (PRECISION_BITS - 1)
        tmp = np.zeros((src.shape[0], out_w, src.shape[2]), dtype=np.int64)
        for x in range(out_w):
            acc = half
            for t in range(xb[x, 1]):
                acc = acc + src[:, xb[x, 0] + t, :] * int(xk[x, t])
            tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        resized = np.zeros((out_h, out_w, src.shape[2]), dtype=np.int64)
        for y in range(out_h):
            acc = half
            for t in range(yb[y, 1]):
                acc = acc + tmp[yb[y, 0] + t, :, :] * int(yk[y, t])
            resized[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
        resized = resized.astype(np.uint8)
    if resized.ndim == 2:
        resized = resized[:, :, None]
    scale, bias = scaling_to_affine(scaling, out_c)
    val = resized.astype(np.float32) * np.array(scale[:out_c], dtype=np.float32) + \
        np.array(bias[:out_c], dtype=np.float32)
    if np.issubdtype(np.dtype(np_dtype), np.integer):
        info = np.iinfo(np_dtype)
        val = np.clip(np.rint(val), info.min, info.max)
    out = val.astype(np_dtype)
    if layout == "NCHW":
        out = np.transpose(out, (2, 0, 1))
    return out


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
        self._table_cache = {}  # (h, w) -> (xb_off, yb_off, xk_off, yk_off, kx, ky, tile_h, max_rows)
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
            if worst * self.out_c * TILE_W <= _LDS_BUDGET:
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
        xb_off = len(self._bounds_host)
        yb_off = xb_off + len(xb)
        xk_off = len(self._coeffs_host)
        yk_off = xk_off + xk.size
        self._bounds_host = np.concatenate([self._bounds_host, xb, yb]).astype(np.int32)
        self._coeffs_host = np.concatenate([self._coeffs_host, xk.reshape(-1), yk.reshape(-1)]).astype(np.int32)
        entry = (xb_off, yb_off, xk_off, yk_off, kx, ky, tile_h, max_rows)
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
        tile_h, max_rows = 16, 1
        host_chunks = []
        for i, im in enumerate(images):
            on_device = hasattr(im, "data_ptr")
            shape = tuple(im.shape)
            h, w = int(shape[0]), int(shape[1])
            c = 1 if len(shape) == 2 else int(shape[2])
            if c not in (1, 3, 4):
                raise ValueError(f"image {i}: unsupported channel count {c}")
            xb_off, yb_off, xk_off, yk_off, kx, ky, th, mr = self._entry(h, w)
            tile_h = min(tile_h, th)
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
        for im in images:
            h = int(im.shape[0])
            _, yb_off, _, _, _, _, th, mr = self._table_cache[(h, int(im.shape[1]))]
            if th == tile_h:
                max_rows = max(max_rows, mr)
            else:
                yb = self._bounds_host[yb_off: yb_off + self.out_h]
                for oy0 in range(0, self.out_h, tile_h):
                    last = min(oy0 + tile_h, self.out_h) - 1
                    max_rows = max(max_rows, int(yb[last, 0] + yb[last, 1] - yb[oy0, 0]))
        self._sync_tables()
        if cursor:
            check(L.tca_hip_memcpy_async(self._dev_src.ptr, self._pinned.ptr, cursor, 1, sh), "image upload")
        check(L.tca_hip_memcpy_async(self._desc_dev.ptr, ctypes.addressof(self._desc_host),
                                     ctypes.sizeof(_bind.PreprocImage) * n, 1, sh), "descriptor upload")
        prm = _bind.PreprocParams()
        prm.out_h, prm.out_w, prm.out_c = self.out_h, self.out_w, self.out_c
        prm.nhwc = 1 if self.nhwc else 0
        prm.tile_h, prm.max_rows = tile_h, max_rows
        prm.image_stride = self.image_elems
        for c in range(4):
            prm.scale[c] = self.scale[c]
            prm.bias[c] = self.bias[c]
        rc = L.tca_preprocess_launch(self._desc_dev.ptr, self._bounds_dev.ptr, self._coeffs_dev.ptr,
                                     as_pointer(out, out_offset), ctypes.byref(prm), n,
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

"""Encoded image files -> uint8 HWC arrays on native threads.

The reference's server-side preprocessing decodes inside a DALI pipeline
(``src/python/examples/ensemble_image_client.py:60-110`` feeds it raw file bytes); here the
decoders of the C++ library (PNG on zlib, sequential and progressive JPEG with libjpeg's
arithmetic, binary PNM - ``csrc/core/image_io.cc``, ``csrc/core/jpeg.cc``) run a whole batch side by side in one
call that holds no interpreter lock, so concurrent requests of a Python model decode in
parallel.  Files they do not read (GIF, BMP, CMYK JPEG, ...) go to Pillow one by one."""

from __future__ import annotations

import ctypes
import io

import numpy as np


class _Image(ctypes.Structure):
    _fields_ = [("height", ctypes.c_int32), ("width", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("pixels", ctypes.c_void_p), ("error", ctypes.c_char * 128)]


def decode_images(blobs, channels=3, threads=8):
    """``blobs``: bytes-like objects (one image file each) -> list of uint8 arrays ``[H, W, channels]``
    (``[H, W]`` for ``channels=1``).  Raises ``ValueError`` naming the first file nobody could decode."""
    from .._native import core

    n = len(blobs)
    if n == 0:
        return []
    lib = core()
    keep = [bytes(b) if not isinstance(b, bytes) else b for b in blobs]
    ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in keep])
    sizes = (ctypes.c_uint64 * n)(*[len(b) for b in keep])
    out = (_Image * n)()
    rc = lib.tca_decode_images(n, ptrs, sizes, int(channels), int(threads), ctypes.byref(out))
    if rc < 0:
        raise ValueError("tca_decode_images: bad arguments")
    images = []
    try:
        for i in range(n):
            item = out[i]
            if item.pixels:
                shape = (item.height, item.width) if item.channels == 1 else (item.height, item.width, item.channels)
                count = item.height * item.width * item.channels
                view = np.ctypeslib.as_array(ctypes.cast(item.pixels, ctypes.POINTER(ctypes.c_uint8)), shape=(count,))
                images.append(view.reshape(shape).copy())
                continue
            try:  # formats the native readers refuse by name
                from PIL import Image

                img = Image.open(io.BytesIO(keep[i]))
                img.load()
                images.append(np.asarray(img.convert("RGB" if channels == 3 else "L") if channels else img))
            except Exception as e:  # noqa: BLE001
                raise ValueError(f"image {i}: {item.error.decode(errors='replace')}; Pillow: {e}") from None
    finally:
        lib.tca_free_images(n, ctypes.byref(out))
    return images

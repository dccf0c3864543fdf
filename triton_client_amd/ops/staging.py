"""Wire-tensor staging on the serving path: host arrays that arrived inside request
messages -> byte ranges of a device buffer, through a page-locked ring, a copy stream
and double-buffered device staging (``csrc/hip/wire_stager.cc``).  One call per batch,
executed without the interpreter lock."""

from __future__ import annotations

import ctypes

import numpy as np

from ._device import lib


class WireStager:
    def __init__(self, device, entry_bytes, dev_bytes, entries=4):
        self._lib = lib()
        if not hasattr(self._lib, "tca_stager_create"):
            raise RuntimeError("libtca_hip.so was built without the wire stager; run "
                               "`python -m triton_client_amd._build hip`")
        handle = ctypes.c_void_p()
        rc = self._lib.tca_stager_create(int(device), int(entry_bytes), int(entries), int(dev_bytes),
                                         ctypes.byref(handle))
        if rc != 0:
            raise RuntimeError(f"tca_stager_create failed ({rc}): {self._lib.tca_stager_error().decode()}")
        self._handle = handle
        self._cap = 0
        self._srcs = self._sizes = self._offsets = None

    def stage(self, arrays, offsets, dst_ptr, compute_stream):
        """``arrays[i]`` (C-contiguous numpy array) lands at ``dst_ptr + offsets[i]``, ordered
        before later work on ``compute_stream`` (a ``torch.cuda.Stream`` or a raw handle)."""
        n = len(arrays)
        if n == 0:
            return
        if n > self._cap:
            self._cap = max(n, 2 * self._cap, 16)
            self._srcs = (ctypes.c_void_p * self._cap)()
            self._sizes = (ctypes.c_uint64 * self._cap)()
            self._offsets = (ctypes.c_uint64 * self._cap)()
        srcs, sizes, offs = self._srcs, self._sizes, self._offsets
        for i, (a, off) in enumerate(zip(arrays, offsets)):
            srcs[i] = a.ctypes.data
            sizes[i] = a.nbytes
            offs[i] = off
        stream = getattr(compute_stream, "cuda_stream", compute_stream)
        rc = self._lib.tca_stager_stage(self._handle, n, srcs, sizes, offs, ctypes.c_void_p(dst_ptr),
                                        ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"tca_stager_stage failed ({rc}): {self._lib.tca_stager_error().decode()}")

    def stats(self):
        b, t, w = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._lib.tca_stager_stats(self._handle, ctypes.byref(b), ctypes.byref(t), ctypes.byref(w))
        return {"bytes": b.value, "tensors": t.value, "ring_waits": w.value}

    def close(self):
        if self._handle:
            self._lib.tca_stager_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def as_staging_source(array):
    """C-contiguous view (or copy) of ``array`` whose bytes can be handed to the stager."""
    return array if isinstance(array, np.ndarray) and array.flags["C_CONTIGUOUS"] else np.ascontiguousarray(array)

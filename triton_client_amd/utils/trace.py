"""ROCTx ranges and markers for rocprofv3 timelines.

``rocprofv3 --marker-trace --kernel-trace -- python ...`` shows these ranges
next to the kernels they enclose (batch execution, graph replay, HIP
preprocessing, IPC handle opening).  The reference has no client-side tracing
(it only *configures* server tracing, ``src/python/library/tritonclient/http/_client.py:767-865``);
this is the MI355X-side addition.

Off by default: every call is a cheap no-op until ``enable()`` is called or the
process starts with ``TCA_ROCTX=1``.  A missing ROCTx library also leaves
everything a no-op.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import threading

_lock = threading.Lock()
_lib = None
_enabled = os.environ.get("TCA_ROCTX", "") not in ("", "0")
_load_failed = False


def _load():
    global _lib, _load_failed
    if _lib is not None or _load_failed:
        return _lib
    with _lock:
        if _lib is not None or _load_failed:
            return _lib
        # rocprofv3 (rocprofiler-sdk) records markers from its own ROCTx library; the older
        # libroctx64 (roctracer) has the same entry points and serves older profilers
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                     "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.argtypes = []
                lib.roctxRangePop.restype = ctypes.c_int
                lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                lib.roctxMarkA.restype = None
                _lib = lib
                return _lib
            except (OSError, AttributeError):
                continue
        _load_failed = True
        return None


def enable(on=True):
    """Switch tracing on or off at run time; returns whether markers will be emitted."""
    global _enabled
    _enabled = bool(on)
    return _enabled and _load() is not None


def enabled():
    return _enabled and _load() is not None


def mark(name):
    if _enabled:
        lib = _load()
        if lib is not None:
            lib.roctxMarkA(str(name).encode())


def push(name):
    """Open a range on the calling thread; returns True when ``pop()`` is owed."""
    if _enabled:
        lib = _load()
        if lib is not None:
            lib.roctxRangePushA(str(name).encode())
            return True
    return False


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def range(name):  # noqa: A001 - mirrors roctx's own vocabulary
    opened = push(name)
    try:
        yield
    finally:
        if opened:
            pop()

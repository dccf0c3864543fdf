#!/usr/bin/env python3
"""Resident set size of a process that does nothing but hipMemcpyAsync (page-locked host memory -> device) on one
stream, waiting for each round of 8 copies through an event - the native executor's staging loop in isolation -
and the same with a hipStreamSynchronize per round or every 64 rounds, with device-to-device and with
device-to-host copies."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import psutil  # noqa: E402

from triton_client_amd.ops import _device  # noqa: E402


def main():
    L = _device.lib()
    me = psutil.Process()
    vp = ctypes.c_void_p
    host, dev, dev2, stream, event = vp(), vp(), vp(), vp(), vp()
    nbytes = 256 << 20
    _device.check(L.tca_hip_host_alloc(ctypes.c_uint64(nbytes), ctypes.byref(host)), "host alloc")
    _device.check(L.tca_hip_malloc(0, ctypes.c_uint64(nbytes), ctypes.byref(dev)), "malloc")
    _device.check(L.tca_hip_malloc(0, ctypes.c_uint64(nbytes), ctypes.byref(dev2)), "malloc")
    _device.check(L.tca_hip_stream_create(0, ctypes.byref(stream)), "stream")
    _device.check(L.tca_hip_event_create(ctypes.byref(event)), "event")
    piece = 2408448  # one batch-8 request
    slots = nbytes // piece
    # "d2h" / "d2d" as the first argument: only that kind, in a fresh process (nothing freed earlier to grow into)
    only = {"d2h": ("D2H, event sync only", 2, dev2, 0), "d2d": ("D2D, event sync only", 3, dev2, 0)}.get(
        sys.argv[1] if len(sys.argv) > 1 else "")
    for label, kind, src, sync_stream in ((only,) * 2) if only else (("H2D, event sync only", 1, host, 0), ("H2D, + stream QUERY per round", 1, host, -1),
                                          ("H2D, event sync only (again)", 1, host, 0),
                                          ("H2D, + stream sync every 64 rounds", 1, host, 64),
                                          ("H2D, event sync only (again)", 1, host, 0),
                                          ("H2D, + stream sync per round", 1, host, 1), ("D2D, event sync only", 3, dev2, 0),
                                          ("D2H, event sync only", 2, dev2, 0), ("D2H, + stream sync every 64 rounds", 2, dev2, 64)):
        t0 = time.time()
        calls = 0
        print(f"{label}: RSS {me.memory_info().rss / 2 ** 20:.0f} MB at start", flush=True)
        for second in range(3):
            t1 = time.time()
            while time.time() - t1 < 2.0:
                for _ in range(20):
                    for i in range(8):
                        off = ((calls + i) % slots) * piece
                        dst = host if kind == 2 else dev
                        L.tca_hip_memcpy_async(vp(dst.value + off), vp(src.value + off), ctypes.c_uint64(piece), kind, stream)
                    L.tca_hip_event_record(event, stream)
                    L.tca_hip_event_sync(event)
                    calls += 8
                    if sync_stream < 0:
                        L.tca_hip_stream_query(stream)
                    elif sync_stream and (calls // 8) % sync_stream == 0:
                        L.tca_hip_stream_sync(stream)
            print(f"  {time.time() - t0:5.1f} s  {calls:8d} copies  RSS {me.memory_info().rss / 2 ** 20:.0f} MB", flush=True)


if __name__ == "__main__":
    main()

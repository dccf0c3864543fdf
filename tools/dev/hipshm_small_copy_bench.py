#!/usr/bin/env python3
"""Latency of small host<->region copies (what a BYTES tensor in a device region
costs the server), single-threaded and from 8 threads."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from triton_client_amd.utils import hip_shared_memory as hipshm

h = hipshm.create_shared_memory_region("bench", 4096, 0)
blob = np.arange(80, dtype=np.uint8)
hipshm.set_shared_memory_region(h, [blob])
m = h  # the owner handle exposes the same native object as a mapping


def loop(n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        hipshm.read_mapping(m, 0, 80)
    t1 = time.perf_counter()
    for _ in range(n):
        hipshm.write_mapping(m, 0, blob.tobytes())
    t2 = time.perf_counter()
    out.append(((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))


res = []
loop(200, res)
res.clear()
loop(2000, res)
print("1 thread: read %.1f us, write %.1f us" % res[0])
res.clear()
ts = [threading.Thread(target=loop, args=(2000, res)) for _ in range(8)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
dt = time.perf_counter() - t0
print("8 threads: %.0f copies/s aggregate (per-thread read %.1f us, write %.1f us)"
      % (8 * 4000 / dt, np.mean([r[0] for r in res]), np.mean([r[1] for r in res])))
hipshm.destroy_shared_memory_region(h)

#!/usr/bin/env python3
"""Stand-alone throughput of the wire stager: 16 tensors of 1.2 MB per call (one ResNet-50 batch
of 16 wire requests), host -> pinned ring -> device staging -> destination."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from triton_client_amd.ops.staging import WireStager

row = 8 * 3 * 224 * 224 * 2
n = 16
arrays = [np.random.default_rng(i).integers(0, 255, size=row, dtype=np.uint8) for i in range(n)]
dst = torch.empty(n * row, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.Stream()
for entries in (4, 8):
    st = WireStager(0, entry_bytes=row, dev_bytes=n * row, entries=entries)
    offsets = [i * row for i in range(n)]
    for _ in range(5):
        st.stage(arrays, offsets, dst.data_ptr(), stream)
    stream.synchronize()
    t0 = time.perf_counter()
    iters = 50
    host = 0.0
    for _ in range(iters):
        h0 = time.perf_counter()
        st.stage(arrays, offsets, dst.data_ptr(), stream)
        host += time.perf_counter() - h0
    stream.synchronize()
    dt = time.perf_counter() - t0
    print(f"threads={os.environ.get('TCA_STAGER_THREADS', 'auto')} ring={entries}: {dt / iters * 1e3:.2f} ms per batch "
          f"(host call {host / iters * 1e3:.2f} ms), {n * row * iters / dt / 1e9:.1f} GB/s, ring waits {st.stats()['ring_waits']}")
    st.close()

#!/usr/bin/env python3
"""Latency of ONE graph replay on an idle GPU (launch call, launch -> completion) per batch size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from triton_client_amd.models import MODEL_ZOO

m = MODEL_ZOO["resnet50"]()
m.load(0)
for n in (8, 16, 32):
    x = torch.randn(n, 3, 224, 224, device="cuda").half()
    o = torch.empty(n, 1000, device="cuda")
    for _ in range(5):
        m.run(x, o, 0, stable=True)
    torch.cuda.synchronize()
    launch, total = [], []
    for _ in range(200):
        time.sleep(0.0005)
        t0 = time.perf_counter()
        m.run(x, o, 0, stable=True)
        t1 = time.perf_counter()
        m.stream_of(0).synchronize()
        t2 = time.perf_counter()
        launch.append(t1 - t0)
        total.append(t2 - t0)
    print(f"batch {n}: launch call p50 {np.percentile(launch, 50) * 1e6:.0f} us, launch -> done p50 {np.percentile(total, 50) * 1e6:.0f} us "
          f"p99 {np.percentile(total, 99) * 1e6:.0f} us", flush=True)

#!/usr/bin/env python3
"""Background host-to-device DMA traffic at a given rate (GB/s) for a number of seconds: an experiment on whether the
GPU's kernels run faster while the DMA engines / the bus are active."""
import sys
import time

import torch

rate_gbs, seconds = float(sys.argv[1]), float(sys.argv[2])
chunk = 2408448
src = torch.empty(chunk, dtype=torch.uint8).pin_memory()
dst = torch.empty(chunk, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
period = chunk / (rate_gbs * 1e9)
t_end = time.perf_counter() + seconds
n = 0
t0 = time.perf_counter()
with torch.cuda.stream(stream):
    while time.perf_counter() < t_end:
        dst.copy_(src, non_blocking=True)
        n += 1
        if n % 8 == 0:
            stream.synchronize()
        target = t0 + n * period
        while time.perf_counter() < target:
            pass
print(f"background copies: {n * chunk / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)

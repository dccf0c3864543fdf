#!/usr/bin/env python3
"""Cost of the pieces of the wire staging path for one 8x3x224x224 fp16 request (1.2 MB)."""
import time

import numpy as np
import torch

n = 16
srcs = [np.random.default_rng(i).integers(0, 255, size=(8, 3, 224, 224),
                                          dtype=np.uint16).view(np.float16) for i in range(n)]
pinned = [torch.empty(srcs[0].nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
dev = torch.empty((128, 3, 224, 224), dtype=torch.float16, device="cuda")
stream = torch.cuda.Stream()


def timeit(label, fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t0) / reps / n * 1e6:8.1f} us per request", flush=True)


def memcpy_only():
    for i, a in enumerate(srcs):
        t = torch.from_numpy(a)
        pinned[i % 4].view(torch.float16).view(t.shape).copy_(t)


def np_memcpy_only():
    for i, a in enumerate(srcs):
        np.copyto(pinned[i % 4].numpy().view(np.float16).reshape(a.shape), a)


def h2d_only():
    with torch.cuda.stream(stream):
        for i in range(n):
            dev[(i % 16) * 8:(i % 16) * 8 + 8].copy_(pinned[i % 4].view(torch.float16).view(8, 3, 224, 224),
                                                     non_blocking=True)


def both():
    with torch.cuda.stream(stream):
        for i, a in enumerate(srcs):
            t = torch.from_numpy(a)
            h = pinned[i % 4].view(torch.float16).view(t.shape)
            h.copy_(t)
            dev[(i % 16) * 8:(i % 16) * 8 + 8].copy_(h, non_blocking=True)


def pageable_direct():
    with torch.cuda.stream(stream):
        for i, a in enumerate(srcs):
            dev[(i % 16) * 8:(i % 16) * 8 + 8].copy_(torch.from_numpy(a), non_blocking=True)


timeit("torch copy_ pageable -> pinned", memcpy_only)
timeit("numpy copyto pageable -> pinned", np_memcpy_only)
timeit("H2D pinned -> device (async, 16 in a row)", h2d_only)
timeit("memcpy + H2D interleaved", both)
timeit("H2D straight from pageable memory", pageable_direct)

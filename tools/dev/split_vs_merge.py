#!/usr/bin/env python3
"""One batch of 2n images on one execution instance vs two batches of n side by side on two."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.models import MODEL_ZOO

m = MODEL_ZOO["resnet50"]()
m.load(0)


def bench(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for n in (8, 16, 32, 64):
    xa = torch.randn(n, 3, 224, 224, device="cuda").half()
    xb = torch.randn(n, 3, 224, 224, device="cuda").half()
    xab = torch.randn(2 * n, 3, 224, 224, device="cuda").half()
    oa = torch.empty(n, 1000, device="cuda")
    ob = torch.empty(n, 1000, device="cuda")
    oab = torch.empty(2 * n, 1000, device="cuda")

    def merged():
        m.run(xab, oab, 0, stable=True)
        m.stream_of(0).synchronize()

    def split():
        m.run(xa, oa, 0, stable=True)
        m.run(xb, ob, 1, stable=True)
        m.stream_of(0).synchronize()
        m.stream_of(1).synchronize()

    print(f"{2 * n:4d} images: one batch {bench(merged):.3f} ms, two batches of {n} side by side {bench(split):.3f} ms", flush=True)

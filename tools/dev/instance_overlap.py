#!/usr/bin/env python3
"""How well two execution instances of ResNet-50 overlap on one GPU: time per forward with one instance
replaying its graph back to back, and per PAIR of forwards with both instances replaying at once.
pair / alone = 1.0 would be perfect overlap, 2.0 none (one hardware queue, or a GPU that is already full)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.models import MODEL_ZOO

# streams created first: shifts the stream -> hardware queue mapping
pre = int(os.environ.get("TCA_EXTRA_STREAMS", "0"))
extra = [torch.cuda.Stream() for _ in range(pre)]
m = MODEL_ZOO["resnet50"]()
m.load(0)
iters = 100
for b in (8, 16, 32, 64, 128):
    xs = [torch.randn(b, 3, 224, 224, device="cuda").half() for _ in range(2)]
    outs = [torch.empty(b, 1000, device="cuda") for _ in range(2)]
    for inst in range(2):
        for _ in range(3):
            m.run(xs[inst], outs[inst], inst, stable=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m.run(xs[0], outs[0], 0, stable=True)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        m.run(xs[0], outs[0], 0, stable=True)
        m.run(xs[1], outs[1], 1, stable=True)
    torch.cuda.synchronize()
    pair = (time.perf_counter() - t0) / iters
    print(f"batch {b:3d}: alone {alone * 1e3:.3f} ms, pair {pair * 1e3:.3f} ms, pair/alone {pair / alone:.2f}, "
          f"{2 * b / pair:.0f} img/s with two instances vs {b / alone:.0f} with one", flush=True)

# the same at batch 128 with the second instance half a forward behind the first (a batch-64 forward first)
b = 128
xs = [torch.randn(b, 3, 224, 224, device="cuda").half() for _ in range(2)]
outs = [torch.empty(b, 1000, device="cuda") for _ in range(2)]
xh, oh = torch.randn(64, 3, 224, 224, device="cuda").half(), torch.empty(64, 1000, device="cuda")
for inst in range(2):
    for _ in range(3):
        m.run(xs[inst], outs[inst], inst, stable=True)
m.run(xh, oh, 1, stable=True)
torch.cuda.synchronize()
for label, offset in (("lockstep", False), ("half a forward apart", True)) * 2:
    t0 = time.perf_counter()
    if offset:
        m.run(xh, oh, 1, stable=True)
    for _ in range(iters):
        m.run(xs[0], outs[0], 0, stable=True)
        m.run(xs[1], outs[1], 1, stable=True)
    torch.cuda.synchronize()
    pair = (time.perf_counter() - t0) / iters
    print(f"batch 128, {label}: pair {pair * 1e3:.3f} ms, {2 * b / pair:.0f} img/s", flush=True)

# three instances at once (TCA_RESNET_INSTANCES=3), batches the closed loop could keep full with six in flight
if m.instance_count >= 3:
    for b3 in (80, 88, 128):
        xs3 = [torch.randn(b3, 3, 224, 224, device="cuda").half() for _ in range(3)]
        outs3 = [torch.empty(b3, 1000, device="cuda") for _ in range(3)]
        for inst in range(3):
            for _ in range(3):
                m.run(xs3[inst], outs3[inst], inst, stable=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            for inst in range(3):
                m.run(xs3[inst], outs3[inst], inst, stable=True)
        torch.cuda.synchronize()
        triple = (time.perf_counter() - t0) / iters
        print(f"batch {b3} x three instances: {triple * 1e3:.3f} ms per round, {3 * b3 / triple:.0f} img/s", flush=True)

# sustained: the same pair loop for several seconds, throughput per second (clock / power effects of a long run)
if os.environ.get("TCA_OVERLAP_SUSTAINED", "1") != "0":
    for second in range(6):
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.0:
            for _ in range(20):
                m.run(xs[0], outs[0], 0, stable=True)
                m.run(xs[1], outs[1], 1, stable=True)
            torch.cuda.synchronize()
            n += 20
        dt = time.perf_counter() - t0
        print(f"batch 128 pairs, second {second}: {2 * b * n / dt:.0f} img/s", flush=True)

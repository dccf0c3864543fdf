#!/usr/bin/env python3
"""Clock and power the GPU holds while ONE kernel runs back to back for a couple of seconds (rocm-smi sampled
from a thread): tells apart "the kernel waits" from "the chip slows the kernel down"."""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.ops import conv

dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                               timeout=5)
            card = next(iter(json.loads(r.stdout).values()))
            out.append({k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower()
                        or "mclk" in k.lower() or "fclk" in k.lower()})
        except Exception as e:  # noqa: BLE001
            out.append({"error": str(e)[:100]})
        time.sleep(0.05)


def run(name, fn, seconds=2.5):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(20):
                fn()
        g.replay()
        stream.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, samples))
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                g.replay()
            stream.synchronize()
            n += 400
        el = time.perf_counter() - t0
        stop.set()
        th.join()
    mid = samples[len(samples) // 3:] or samples
    print(f"{name}: {el / n * 1e6:8.2f} us per call; samples {len(samples)}; last: {mid[-1] if mid else None}", flush=True)
    return mid


h = w = 56
x = nhwc(torch.randn(batch, 64, h, w, device=dev).half())
w1 = nhwc((torch.randn(64, 64, 3, 3, device=dev) / 24.0).half())
b1 = torch.randn(64, device=dev).half()
w2 = (torch.randn(256, 64, device=dev) / 8.0).half()
w2p = conv.permute_mid_k(w2)
b2 = torch.randn(256, device=dev).half()
w3 = (torch.randn(64, 256, device=dev) / 16.0).half()
b3 = torch.randn(64, device=dev).half()
res = nhwc(torch.randn(batch, 256, h, w, device=dev).half())
other = nhwc(torch.randn(batch, 256, h, w, device=dev).half())
out = nhwc(torch.empty(batch, 256, h, w, device=dev, dtype=torch.float16))
x14 = nhwc(torch.randn(batch, 256, 14, 14, device=dev).half())
w14 = nhwc((torch.randn(256, 256, 3, 3, device=dev) / 48.0).half())
b14 = torch.randn(256, device=dev).half()
time.sleep(1.0)
run("idle-ish (tiny kernel)", lambda: b1.add_(0))
run("elementwise add (3 x 205 MB)", lambda: torch.add(res, other, out=out))
run("rows kernel", lambda: conv.bottleneck_rows(x, w1, b1, w2p, b2, res, w3, b3, out=out))
run("chunked kernel", lambda: conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, padding=1, out=out, next_weight=w3,
                                                      next_bias=b3, tile_m=128))
run("3x3 14x14 256->256 (compute-bound)", lambda: conv.conv2d(x14, w14, b14, None, relu=True, padding=1))

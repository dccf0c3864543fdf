#!/usr/bin/env python3
"""Stage-1 bottleneck tails of ResNet-50 (56x56, 64 -> 64 -> 256, next block's first 1x1 attached): the chunked
fused kernel (three workgroup barriers per 64-channel chunk) against the row-organised kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.ops import conv

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()


def graph_time(fn, reps=10, rounds=5):
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                fn()
        g.replay()
        stream.synchronize()
        best = None
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            g.replay()
            b.record(stream)
            b.synchronize()
            t = a.elapsed_time(b) * 1e3 / reps
            best = t if best is None else min(best, t)
    return best


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


batches = [int(v) for v in sys.argv[1:]] or [8, 32, 64, 128]
for batch in batches:
    h = w = 56
    for n3, proj in ((64, True), (64, False), (128, False)):
        x = nhwc(torch.randn(batch, 64, h, w, device=dev).half())
        w1 = nhwc((torch.randn(64, 64, 3, 3, device=dev) / 24.0).half())
        b1 = torch.randn(64, device=dev).half()
        w2 = (torch.randn(256, 64, device=dev) / 8.0).half()
        b2 = torch.randn(256, device=dev).half()
        w3 = (torch.randn(n3, 256, device=dev) / 16.0).half()
        b3 = torch.randn(n3, device=dev).half()
        x0 = nhwc(torch.randn(batch, 64, h, w, device=dev).half()) if proj else None
        res = None if proj else nhwc(torch.randn(batch, 256, h, w, device=dev).half())
        wcat = torch.cat([w2, (torch.randn(256, 64, device=dev) / 8.0).half()], dim=1).contiguous() if proj else w2
        wperm = conv.permute_mid_k(wcat)
        out = nhwc(torch.empty(batch, 256, h, w, device=dev, dtype=torch.float16))

        def chunked(tile):
            return lambda: conv.conv2d_pointwise(x, w1, b1, wcat, b2, res, relu=True, padding=1, out=out, next_weight=w3,
                                                 next_bias=b3, proj_input=x0, tile_m=tile)

        def rows():
            conv.bottleneck_rows(x, w1, b1, wperm, b2, res, w3, b3, proj_input=x0, out=out)

        mb = batch * h * w * (64 + 256 * (1 if proj else 2) + n3 + (64 if proj else 0)) * 2 / 1e6
        t64, t128, tr = graph_time(chunked(64)), graph_time(chunked(128)), graph_time(rows)
        print(f"batch {batch:3d} n3={n3:3d} proj={int(proj)}: chunked 64-px {t64:7.2f} us, 128-px {t128:7.2f} us, "
              f"rows {tr:7.2f} us ({mb / tr:.2f} TB/s)", flush=True)

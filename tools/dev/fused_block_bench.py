#!/usr/bin/env python3
"""3x3 convolution + 1x1 expansion of a bottleneck block: two kernels vs the fused kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.ops import conv

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()


def graph_time(fn, reps=10):
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                fn()
        g.replay()
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(5):
            g.replay()
        b.record(stream)
        b.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


# (H, W, Cin, Mid, Cout, stride)
SHAPES = [(56, 56, 64, 64, 256, 1), (56, 56, 128, 128, 512, 2), (28, 28, 128, 128, 512, 1)]
for batch in (8, 32, 128):
    for (h, w, cin, mid, cout, stride) in SHAPES:
        x = nhwc(torch.randn(batch, cin, h, w, device=dev).half())
        w1 = nhwc((torch.randn(mid, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).half())
        b1 = torch.randn(mid, device=dev).half()
        w2 = (torch.randn(cout, mid, device=dev) / mid ** 0.5).half()
        b2 = torch.randn(cout, device=dev).half()
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        res = nhwc(torch.randn(batch, cout, oh, ow, device=dev).half())
        t = nhwc(torch.empty(batch, mid, oh, ow, device=dev, dtype=torch.float16))
        out = nhwc(torch.empty(batch, cout, oh, ow, device=dev, dtype=torch.float16))

        def two():
            conv.conv2d(x, w1, b1, None, relu=True, stride=stride, padding=1, out=t)
            conv.conv1x1(t, w2, b2, res, relu=True, out=out)

        def fused():
            conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1, out=out)

        line = (f"batch {batch:3d} {h}x{w} {cin}->{mid}->{cout} s{stride}: two kernels {graph_time(two):7.2f} us, "
                f"fused {graph_time(fused):7.2f} us")
        if mid == 64:
            # with the next block's first 1x1 convolution (cout -> 64) attached, vs running it separately
            w3 = (torch.randn(64, cout, device=dev) / cout ** 0.5).half()
            b3 = torch.randn(64, device=dev).half()
            t3 = nhwc(torch.empty(batch, 64, oh, ow, device=dev, dtype=torch.float16))

            def fused_then_next():
                fused()
                conv.conv1x1(out, w3, b3, None, relu=True, out=t3)

            def three():
                conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1, out=out,
                                      next_weight=w3, next_bias=b3)

            line += f" | + next 1x1: separate {graph_time(fused_then_next):7.2f} us, attached {graph_time(three):7.2f} us"
            if stride == 1 and cin == 64:
                # first block of the stage: projection shortcut (64 -> cout on the block input) as its own
                # kernel + residual, vs inside the kernel
                x0 = nhwc(torch.randn(batch, 64, oh, ow, device=dev).half())
                wd = (torch.randn(cout, 64, device=dev) / 8.0).half()
                bd = torch.randn(cout, device=dev).half()
                wcat = torch.cat([w2, wd], dim=1).contiguous()

                def with_down_kernel():
                    conv.conv1x1(x0, wd, bd, None, relu=False, out=res)
                    three()

                def in_kernel():
                    conv.conv2d_pointwise(x, w1, b1, wcat, b2, None, relu=True, stride=1, padding=1, out=out,
                                          next_weight=w3, next_bias=b3, proj_input=x0)

                line += f" | projection: own kernel {graph_time(with_down_kernel):7.2f} us, in kernel {graph_time(in_kernel):7.2f} us"
        print(line, flush=True)

#!/usr/bin/env python3
"""Classifier tail: library (mean + linear + cast) vs the in-tree pool + linear kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.ops import conv

stream = torch.cuda.Stream()


def graph_time(fn, reps=20):
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


fc = torch.nn.Linear(2048, 1000).cuda().half()
for batch in (8, 32, 128):
    y = torch.randn(batch, 2048, 7, 7, device="cuda").half().contiguous(memory_format=torch.channels_last)
    out = torch.empty(batch, 1000, device="cuda")
    feats = torch.empty(batch, 2048, device="cuda", dtype=torch.float16)
    lib = graph_time(lambda: out.copy_(fc(y.mean(dim=(2, 3)))))
    own = graph_time(lambda: conv.avgpool_linear(y, fc.weight, fc.bias, out))
    from triton_client_amd.ops._device import lib as L
    s = lambda: torch.cuda.current_stream().cuda_stream
    pool = graph_time(lambda: L().tca_avgpool_launch(y.data_ptr(), feats.data_ptr(), batch, 49, 2048, s()))
    lin = graph_time(lambda: L().tca_linear_launch(feats.data_ptr(), fc.weight.data_ptr(), fc.bias.data_ptr(),
                                                   out.data_ptr(), batch, 1000, 2048, s()))
    print(f"batch {batch:3d}: library {lib:6.2f} us, in-tree {own:6.2f} us (pool {pool:5.2f} + linear {lin:5.2f})",
          flush=True)

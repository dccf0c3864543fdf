#!/usr/bin/env python3
"""Where the time of the row-organised bottleneck kernel goes: the same launch with parts of its memory traffic
switched off (TCA_ROWS_ABLATE is read once per process, so every setting runs in a child), next to a plain
elementwise pass with a comparable read / write mix."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch

    from triton_client_amd.ops import conv

    batch = int(sys.argv[2])
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()

    def nhwc(t):
        return t.contiguous(memory_format=torch.channels_last)

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

    h = w = 56
    x = nhwc(torch.randn(batch, 64, h, w, device=dev).half())
    w1 = nhwc((torch.randn(64, 64, 3, 3, device=dev) / 24.0).half())
    b1 = torch.randn(64, device=dev).half()
    w2 = conv.permute_mid_k((torch.randn(256, 64, device=dev) / 8.0).half())
    b2 = torch.randn(256, device=dev).half()
    w3 = (torch.randn(64, 256, device=dev) / 16.0).half()
    b3 = torch.randn(64, device=dev).half()
    res = nhwc(torch.randn(batch, 256, h, w, device=dev).half())
    out = nhwc(torch.empty(batch, 256, h, w, device=dev, dtype=torch.float16))
    t = graph_time(lambda: conv.bottleneck_rows(x, w1, b1, w2, b2, res, w3, b3, out=out))
    line = f"ablate={os.environ.get('TCA_ROWS_ABLATE', '0')}: rows {t:7.2f} us"
    if os.environ.get("TCA_ROWS_ABLATE", "0") == "0":
        other = nhwc(torch.randn(batch, 256, h, w, device=dev).half())
        te = graph_time(lambda: torch.add(res, other, out=out))
        mb = 3 * res.numel() * 2 / 1e6
        tc = graph_time(lambda: out.copy_(res))
        line += f" | elementwise add of two such tensors {te:7.2f} us = {mb / te:.2f} TB/s | copy {tc:7.2f} us = {2 * res.numel() * 2 / 1e6 / tc:.2f} TB/s"
    print(line, flush=True)
    sys.exit(0)

batch = sys.argv[1] if len(sys.argv) > 1 else "128"
for flags in (sys.argv[2:] or ["0", "1", "2", "3", "4", "7"]):
    env = dict(os.environ, TCA_ROWS_ABLATE=flags)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", batch], env=env, check=True, timeout=120)

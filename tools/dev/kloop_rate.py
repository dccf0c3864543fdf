#!/usr/bin/env python3
"""Steady-state rate of the convolution GEMM's K loop: a 1x1 convolution over 14x14 pixels at batch 128 with
1024, 4096 and 8192 input channels (the same tiles, prologue and epilogue; only the number of K steps differs),
per tile variant.  The difference between two lengths, per K step and tile round, is what the loop itself costs;
the rest is prologue, epilogue and launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch

    from triton_client_amd.ops import conv

    variants = (sys.argv[1] if len(sys.argv) > 1 else "128x128w8,128x256w16,256x128w16,256x256w16,128x64w8").split(",")
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    batch, h, w = 128, 14, 14
    m = batch * h * w

    def timed(fn, reps=10):
        with torch.cuda.stream(st):
            fn()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    fn()
            g.replay()
            st.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                g.replay()
            st.synchronize()
        return (time.perf_counter() - t0) * 1e6 / (4 * reps)

    for variant in variants:
        bm, bn = (int(v) for v in variant.split("w")[0].split("x"))
        tiles = ((m + bm - 1) // bm) * (cout // bn)
        row = []
        times = {}
        for cin in (1024, 4096, 8192):
            x = torch.randn(batch, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
            wt = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).half().contiguous(memory_format=torch.channels_last)
            bias = torch.randn(cout, device=dev).half()
            out = torch.empty(batch, cout, h, w, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
            t = timed(lambda: conv.conv2d(x, wt, bias, None, relu=True, stride=1, padding=0, variant=variant, out=out))
            times[cin] = t
            row.append(f"cin {cin}: {t:7.1f} us {2.0 * m * cout * cin / t / 1e6:6.0f} TF/s")
            del x, wt
        steps = (8192 - 4096) // 64
        per_step_us = (times[8192] - times[4096]) / steps
        flop_step = 2.0 * m * cout * 64
        print(f"{variant:11s} cout {cout} tiles {tiles:4d}: " + " | ".join(row) +
              f" | loop alone: {per_step_us * 1e3:6.1f} ns per K step over all tiles = {flop_step / per_step_us / 1e6:5.0f} TF/s, "
              f"fixed part {times[4096] - per_step_us * 64:6.1f} us", flush=True)


if __name__ == "__main__":
    main()

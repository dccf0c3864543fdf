#!/usr/bin/env python3
"""Every tile variant of the convolution GEMM on ResNet-50's stage-3 / stage-4 layers (and any other layer of
``tools/conv_bench.py``): us per call alone and with the same work on two streams at once, and the largest
difference from the 128x128 tile's result.

    python tools/dev/tile_compare.py [--batch 128] [--variants a,b,...] [--min-cin 256]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--min-cin", type=int, default=256)
    ap.add_argument("--variants", default="128x128w8,128x256w16,256x256w16,256x128w16,128x128w8/reg")
    args = ap.parse_args()
    import torch

    from conv_bench import LAYERS, LAYERS_3X3
    from triton_client_amd.ops import conv

    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream() for _ in range(2)]

    def timed(fn, n_streams):
        graphs = []
        for st in streams[:n_streams]:
            with torch.cuda.stream(st):
                fn()
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for _ in range(args.reps):
                        fn()
                g.replay()
                st.synchronize()
                graphs.append(g)
        rounds = 6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / (rounds * args.reps * n_streams)

    def nhwc(t):
        return t.contiguous(memory_format=torch.channels_last)

    layers = [(1, s, h, w, cin, cout, r, relu) for (h, w, cin, cout, s, r, relu) in LAYERS] + \
             [(3, s, h, w, c, c, False, True) for (h, w, c, s) in LAYERS_3X3]
    batch = args.batch
    for (k, stride, h, w, cin, cout, with_res, relu) in layers:
        if cin < args.min_cin:
            continue
        pad = k // 2
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = nhwc(torch.randn(batch, cin, h, w, device=dev).half())
        wt = nhwc((torch.randn(cout, cin, k, k, device=dev) / (k * k * cin) ** 0.5).half())
        bias = torch.randn(cout, device=dev).half()
        res = nhwc(torch.randn(batch, cout, oh, ow, device=dev).half()) if with_res else None
        out = nhwc(torch.empty(batch, cout, oh, ow, device=dev, dtype=torch.float16))
        ref = None
        cells = []
        for variant in args.variants.split(","):
            variant, _, staging = variant.partition("/")  # "256x256w4/reg": register staging instead of the LDS DMA
            tile_n = int(variant.split("x")[1].split("w")[0].replace("halo", ""))
            if cout % tile_n:
                continue

            def call(v=variant, sg=staging or None):
                return conv.conv2d(x, wt, bias, res, relu=relu, stride=stride, padding=pad, variant=v, out=out, staging=sg)

            call()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
                err = 0.0
            else:
                err = float((out.float() - ref.float()).abs().max())
            cells.append(f"{variant}{'/' + staging if staging else ''} {timed(call, 1):6.1f}/{timed(call, 2):6.1f} (d {err:.3g})")
        print(f"b{batch} k{k} s{stride} {h}x{w} {cin}->{cout}{'+r' if with_res else ''}: " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Split-K on the ResNet-50 layers with few output tiles: per layer and batch, the MFMA kernel
without split-K, with the automatic choice, with fixed split counts, and library + epilogue."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,16,32")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F

    from conv_bench import LAYERS, LAYERS_3X3
    from triton_client_amd.models.vision import install_miopen_db
    from triton_client_amd.ops import conv

    install_miopen_db()
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()

    def graph_time(fn):
        with torch.cuda.stream(stream):
            fn()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(args.reps):
                    fn()
            g.replay()
            stream.synchronize()
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(stream)
            for _ in range(5):
                g.replay()
            end.record(stream)
            end.synchronize()
        return start.elapsed_time(end) * 1e3 / (5 * args.reps)

    cols = [("auto", 1), ("auto", 0), ("64x64w8", 2), ("64x64w8", 4), ("64x64w8", 8), ("64x64", 4), ("128x64w8", 4)]
    print(f"{'batch':>5} {'layer':>26} {'lib+ep':>8} " + " ".join(f"{v + '/' + str(k):>11}" for v, k in cols),
          flush=True)
    sums = {}
    for b in [int(v) for v in args.batches.split(",")]:
        layers = [(h, w, cin, cout, 1, s, r, relu) for (h, w, cin, cout, s, r, relu) in LAYERS] + \
                 [(h, w, c, c, 3, s, False, True) for (h, w, c, s) in LAYERS_3X3]
        for (h, w, cin, cout, ks, stride, with_res, relu) in layers:
            pad = ks // 2
            oh, ow = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
            tiles = ((b * oh * ow + 63) // 64) * (cout // 64)
            if tiles > 128:
                continue
            x = torch.randn(b, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
            wt = (torch.randn(cout, cin, ks, ks, device=dev) / (ks * ks * cin) ** 0.5).half().contiguous(
                memory_format=torch.channels_last)
            bias = torch.randn(cout, device=dev).half()
            res = torch.randn(b, cout, oh, ow, device=dev).half().contiguous(
                memory_format=torch.channels_last) if with_res else None
            out = torch.empty(b, cout, oh, ow, device=dev, dtype=torch.float16).contiguous(
                memory_format=torch.channels_last)
            t = [graph_time(lambda: conv.bias_act_(F.conv2d(x, wt, None, stride=stride, padding=pad), bias, res, relu))]
            for variant, k in cols:
                t.append(graph_time(lambda v=variant, k=k: conv.conv2d(x, wt, bias, res, relu=relu, stride=stride,
                                                                        padding=pad, variant=v, out=out, split_k=k)))
            name = f"{ks}x{ks} {h}x{w} {cin}->{cout} s{stride}{'+r' if with_res else ''}"
            print(f"{b:5d} {name:>26} " + " ".join(f"{v:8.2f}" if i == 0 else f"{v:11.2f}" for i, v in enumerate(t)),
                  flush=True)
            s = sums.setdefault(b, [0.0] * len(t))
            for i, v in enumerate(t):
                s[i] += v
    for b, s in sums.items():
        print(f"batch {b} sums: lib+ep {s[0]:.1f}, " + ", ".join(f"{v}/{k} {x:.1f}" for (v, k), x in zip(cols, s[1:])))


if __name__ == "__main__":
    main()

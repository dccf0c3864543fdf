#!/usr/bin/env python3
"""Per-layer timing of ResNet-50's 1x1 convolutions: library convolution with
separate bias / residual / ReLU passes vs library convolution + one fused
epilogue pass vs the MFMA GEMM with the epilogue inside (all tile variants).
Every candidate is captured in a HIP graph with ``--reps`` back-to-back calls so
the number is GPU time, not launch overhead."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (H, W, Cin, Cout, stride, residual, relu) of every distinct 1x1 convolution
LAYERS = [
    (56, 56, 64, 64, 1, False, True), (56, 56, 64, 256, 1, True, True), (56, 56, 64, 256, 1, False, False),
    (56, 56, 256, 64, 1, False, True), (56, 56, 256, 128, 1, False, True), (28, 28, 128, 512, 1, True, True),
    (56, 56, 256, 512, 2, False, False), (28, 28, 512, 128, 1, False, True), (28, 28, 512, 256, 1, False, True),
    (14, 14, 256, 1024, 1, True, True), (28, 28, 512, 1024, 2, False, False), (14, 14, 1024, 256, 1, False, True),
    (14, 14, 1024, 512, 1, False, True), (7, 7, 512, 2048, 1, True, True), (14, 14, 1024, 2048, 2, False, False),
    (7, 7, 2048, 512, 1, False, True),
]


# (H, W, C, stride) of every distinct 3x3 convolution (Cin == Cout)
LAYERS_3X3 = [(56, 56, 64, 1), (56, 56, 128, 2), (28, 28, 128, 1), (28, 28, 256, 2), (14, 14, 256, 1),
              (14, 14, 512, 2), (7, 7, 512, 1)]


MFMA_VARIANTS = tuple(os.environ["TCA_BENCH_VARIANTS"].split(",")) if os.environ.get("TCA_BENCH_VARIANTS") else (
    "128x64", "64x64", "128x128w8", "128x64w8", "64x64w8", "128x256w8", "256x128w16", "128x256w16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,32,128")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F

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
        return start.elapsed_time(end) * 1e3 / (5 * args.reps)  # us per call

    print(f"{'batch':>5} {'layer':>26} {'eager':>8} {'lib+ep':>8} " + " ".join(f"{v:>8}" for v in MFMA_VARIANTS)
          + "  best",
          flush=True)
    totals = {}
    for b in [int(v) for v in args.batches.split(",")]:
        for (h, w, cin, cout, stride, with_res, relu) in LAYERS:
            x = torch.randn(b, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
            wt4 = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).half().contiguous(
                memory_format=torch.channels_last)
            wt2 = wt4.reshape(cout, cin).contiguous()
            bias = torch.randn(cout, device=dev).half()
            oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
            res = torch.randn(b, cout, oh, ow, device=dev).half().contiguous(
                memory_format=torch.channels_last) if with_res else None

            def eager():
                y = F.conv2d(x, wt4, bias, stride=stride)
                if res is not None:
                    y += res
                if relu:
                    torch.relu_(y)
                return y

            def lib_ep():
                y = F.conv2d(x, wt4, None, stride=stride)
                return conv.bias_act_(y, bias, res, relu)

            out = torch.empty(b, cout, oh, ow, device=dev, dtype=torch.float16).contiguous(
                memory_format=torch.channels_last)
            times = {"eager": graph_time(eager), "lib+ep": graph_time(lib_ep)}
            for variant in MFMA_VARIANTS:
                if ("x128" in variant and cout % 128) or ("x256" in variant and cout % 256):
                    times[variant] = float("nan")
                    continue
                times[variant] = graph_time(
                    lambda v=variant: conv.conv1x1(x, wt2, bias, res, relu=relu, stride=stride, variant=v, out=out))
            best = min((k for k in times if times[k] == times[k]), key=lambda k: times[k])
            name = f"{h}x{w} {cin}->{cout} s{stride}{'+r' if with_res else ''}"
            print(f"{b:5d} {name:>26} " + " ".join(f"{times[k]:8.2f}" for k in
                                                  ("eager", "lib+ep") + MFMA_VARIANTS) + f"  {best}",
                  flush=True)
            t = totals.setdefault(b, {"eager": 0.0, "lib+ep": 0.0, "mfma_best": 0.0, "best": 0.0})
            t["eager"] += times["eager"]
            t["lib+ep"] += times["lib+ep"]
            mf = min((times[k] for k in MFMA_VARIANTS if times[k] == times[k]), default=times["lib+ep"])
            t["mfma_best"] += mf
            t["best"] += min(mf, times["lib+ep"])
        for (h, w, c, stride) in LAYERS_3X3:
            x = torch.randn(b, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
            wt = (torch.randn(c, c, 3, 3, device=dev) / (9 * c) ** 0.5).half().contiguous(
                memory_format=torch.channels_last)
            bias = torch.randn(c, device=dev).half()

            def eager3():
                return torch.relu_(F.conv2d(x, wt, bias, stride=stride, padding=1))

            def lib_ep3():
                return conv.bias_act_(F.conv2d(x, wt, None, stride=stride, padding=1), bias, None, True)

            times = {"eager": graph_time(eager3), "lib+ep": graph_time(lib_ep3)}
            for variant in MFMA_VARIANTS:
                if ("x128" in variant and c % 128) or ("x256" in variant and c % 256):
                    times[variant] = float("nan")
                    continue
                times[variant] = graph_time(
                    lambda v=variant: conv.conv2d(x, wt, bias, None, relu=True, stride=stride, padding=1, variant=v))
            best = min((k for k in times if times[k] == times[k]), key=lambda k: times[k])
            name = f"3x3 {h}x{w} {c}->{c} s{stride}"
            print(f"{b:5d} {name:>26} " + " ".join(f"{times[k]:8.2f}" for k in
                                                  ("eager", "lib+ep") + MFMA_VARIANTS) + f"  {best}",
                  flush=True)
    for b, t in totals.items():
        print(f"batch {b}: sum over distinct layers (us): " + ", ".join(f"{k}={v:.1f}" for k, v in t.items()),
              flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Stand-alone timing of the ResNet-50 instance: ms per forward for several
batch sizes, HIP graph replay vs eager, MIOpen benchmark on/off.  Used to pick
the batcher's bucket sizes and as the rocprofv3 target."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,16,32,64,128")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--benchmark", type=int, default=0)
    ap.add_argument("--no-fused", action="store_true", help="plain library forward instead of the fused kernels")
    args = ap.parse_args()
    import torch

    from triton_client_amd.models import MODEL_ZOO

    t0 = time.time()
    model = MODEL_ZOO["resnet50"]()
    model.load(0)
    torch.backends.cudnn.benchmark = bool(args.benchmark)
    model.use_graphs = not args.no_graph
    model.fused = not args.no_fused
    print(f"load {time.time() - t0:.1f}s graphs={model.use_graphs} fused={model.fused} benchmark={args.benchmark}",
          flush=True)
    for b in [int(x) for x in args.batches.split(",")]:
        x = torch.randn(b, 3, 224, 224, device="cuda", dtype=torch.float16)
        out = torch.empty(b, 1000, device="cuda", dtype=torch.float32)
        t1 = time.time()
        model.run(x, out, stable=True)
        model.stream.synchronize()
        first = time.time() - t1
        for _ in range(5):
            model.run(x, out, stable=True)
        model.stream.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.iters):
            model.run(x, out, stable=True)
        model.stream.synchronize()
        ms = (time.perf_counter() - t2) * 1e3 / args.iters
        print(f"batch {b:4d}: {ms:7.3f} ms/forward  {b / ms * 1e3:9.0f} img/s   (first call {first:.2f}s)", flush=True)


if __name__ == "__main__":
    main()

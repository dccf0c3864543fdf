#!/usr/bin/env python3
"""Single-process driver of every hand-written kernel plus the ResNet-50 graph,
meant to run under ``rocprofv3 --kernel-trace --stats`` (per-kernel time) or
``rocprofv3 --pmc ...`` (counters).  ``--only`` restricts the run to one group so
a counter pass stays short."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resnet-batch-only", action="store_true", help="ResNet-50: only --batch instead of 8/64/128")
    ap.add_argument("--only", default="all", choices=["all", "preprocess", "codec", "classify", "resnet"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    import torch

    from triton_client_amd.ops import classification, codec
    from triton_client_amd.ops.preprocess import Preprocessor

    rng = np.random.default_rng(0)
    dev = torch.device("cuda:0")
    report = {}

    def timed(name, fn, iters=args.iters):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        report[name] = (time.perf_counter() - t0) * 1e3 / iters

    if args.only in ("all", "preprocess"):
        images = [rng.integers(0, 255, size=(480, 640, 3), dtype=np.uint8) for _ in range(args.batch)]
        d_images = [torch.from_numpy(im).to(dev) for im in images]
        for layout, dtype in (("NCHW", "FP16"), ("NHWC", "FP32")):
            pre = Preprocessor(224, 224, 3, layout, dtype, scaling="INCEPTION", resize="pil", device=0,
                               max_batch=args.batch)
            out = torch.empty(args.batch * pre.image_bytes, dtype=torch.uint8, device=dev)
            timed(f"preprocess_{layout}_{dtype}_b{args.batch}_device_src", lambda: pre(d_images, out))
            timed(f"preprocess_{layout}_{dtype}_b{args.batch}_host_src", lambda: pre(images, out))
            pre.close()
    if args.only in ("all", "codec"):
        n = 1 << 16
        lens = rng.integers(1, 64, size=n).astype(np.int32)
        payload = torch.from_numpy(rng.integers(0, 255, size=int(lens.sum()), dtype=np.uint8)).to(dev)
        d_lens = torch.from_numpy(lens).to(dev)
        packed = codec.bytes_pack_from_lengths(payload, d_lens)
        timed("bytes_pack_64k_elems", lambda: codec.bytes_pack_from_lengths(payload, d_lens))
        timed("bytes_index_64k_elems", lambda: codec.bytes_index(packed, n))
        x = torch.randn(1 << 24, device=dev)
        timed("fp32_to_bf16_16M", lambda: codec.fp32_to_bf16(x))
        h = codec.fp32_to_bf16(x)
        timed("bf16_to_fp32_16M", lambda: codec.bf16_to_fp32(h))
    if args.only in ("all", "classify"):
        logits = torch.randn(128, 1000, device=dev)
        labels = [f"label_{i}" for i in range(1000)]
        table = classification.LabelTable(labels, dev)

        def classify():
            vals, idx = classification.topk(logits, 5)
            classification.classification_bytes(vals, idx, table)

        timed("topk5_strings_128x1000", classify)
    if args.only in ("all", "resnet"):
        from triton_client_amd.models import MODEL_ZOO

        model = MODEL_ZOO["resnet50"]()
        model.load(0)
        for b in ([args.batch] if args.resnet_batch_only else (8, 64, 128)):
            x = torch.randn(b, 3, 224, 224, device=dev, dtype=torch.float16)
            out = torch.empty(b, 1000, device=dev, dtype=torch.float32)

            def fwd():
                model.run(x, out, stable=True)
                model.stream.synchronize()

            timed(f"resnet50_fp16_graph_b{b}", fwd, iters=max(5, args.iters // 2))
    for k, v in report.items():
        print(f"{k:48s} {v:9.4f} ms", flush=True)


if __name__ == "__main__":
    main()

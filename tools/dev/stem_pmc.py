#!/usr/bin/env python3
"""Runs the fused network stem (7x7/2 convolution + bias + ReLU + 3x3/2 max pooling in one
kernel) a few times at one batch size: the program `rocprofv3 --pmc` wraps (tools/dev/stem_pmc.sh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import torch

    from triton_client_amd.ops import conv

    x = torch.randn(args.batch, 3, 224, 224, device="cuda", dtype=torch.float16)
    w = conv.pack_stem_weight(torch.randn(64, 3, 7, 7, device="cuda", dtype=torch.float16) * 0.05)
    b = torch.randn(64, device="cuda", dtype=torch.float16)
    out = torch.empty(args.batch, 64, 56, 56, device="cuda",
                      dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    for _ in range(args.iters):
        conv.stem_conv_pool(x, w, b, out=out)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.iters):
        conv.stem_conv_pool(x, w, b, out=out)
    t1.record()
    torch.cuda.synchronize()
    print(f"stem b{args.batch}: {t0.elapsed_time(t1) * 1e3 / args.iters:.1f} us")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Run ONE convolution layer through ONE tile variant of the MFMA kernel a few
times - the program `rocprofv3 --pmc` wraps to collect counters for that kernel
alone (see tools/conv_pmc.sh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--hw", type=int, default=14)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--ksize", type=int, default=3)
    ap.add_argument("--variant", default="128x64")
    ap.add_argument("--staging", default="lds")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--expand", type=int, default=0,
                    help="fused block kernel: 3x3 cin->cout, then 1x1 cout->EXPAND with residual (cout = 64 | 128)")
    ap.add_argument("--next", type=int, default=0, help="with --expand: attach the next block's 1x1 EXPAND->NEXT")
    args = ap.parse_args()
    import torch

    from triton_client_amd.ops import conv

    dev = torch.device("cuda:0")
    x = torch.randn(args.batch, args.cin, args.hw, args.hw, device=dev).half().contiguous(
        memory_format=torch.channels_last)
    k = args.ksize
    w = (torch.randn(args.cout, args.cin, k, k, device=dev) / (args.cin * k * k) ** 0.5).half()
    w2 = w.reshape(args.cout, -1).contiguous()
    w4 = w.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(args.cout, device=dev).half()
    if args.expand:
        we = (torch.randn(args.expand, args.cout, device=dev) / args.cout ** 0.5).half()
        be = torch.randn(args.expand, device=dev).half()
        res = torch.randn(args.batch, args.expand, args.hw, args.hw, device=dev).half().contiguous(
            memory_format=torch.channels_last)
        wn = (torch.randn(args.next, args.expand, device=dev) / args.expand ** 0.5).half() if args.next else None
        bn = torch.randn(args.next, device=dev).half() if args.next else None
        for _ in range(args.iters):
            conv.conv2d_pointwise(x, w4, bias, we, be, res, relu=True, stride=1, padding=k // 2, next_weight=wn,
                                  next_bias=bn)
        torch.cuda.synchronize()
        flops = 2.0 * args.batch * args.hw * args.hw * args.cout * (args.cin * k * k + args.expand) + \
            2.0 * args.batch * args.hw * args.hw * args.expand * args.next
        print(f"layer b{args.batch} {args.hw}x{args.hw} {args.cin}->{args.cout}->{args.expand}"
              f"{'->' + str(args.next) if args.next else ''} k{k} fused: {flops / 1e9:.2f} GFLOP per call")
        return
    for _ in range(args.iters):
        if k == 1:
            conv.conv1x1(x, w2, bias, relu=True, variant=args.variant, staging=args.staging)
        else:
            conv.conv2d(x, w4, bias, stride=1, padding=k // 2, relu=True, variant=args.variant, staging=args.staging)
    torch.cuda.synchronize()
    flops = 2.0 * args.batch * args.hw * args.hw * args.cout * args.cin * k * k
    print(f"layer b{args.batch} {args.hw}x{args.hw} {args.cin}->{args.cout} k{k} {args.variant}/{args.staging}: "
          f"{flops / 1e9:.2f} GFLOP per call")


if __name__ == "__main__":
    main()

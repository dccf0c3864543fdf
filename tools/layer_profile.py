#!/usr/bin/env python3
"""Per-layer timing of the fused ResNet-50 forward: every kernel call of the plan is timed on
its own (HIP events, warm caches), with its arithmetic and the bytes it must move, so the
layers furthest from the roofline are visible.

  python tools/layer_profile.py --batch 128

Columns: us, GFLOP, MB (inputs + weights + outputs + residual, each once), TFLOP/s, TB/s,
tiles (output tiles of 128x128), share of the summed time."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch

    from triton_client_amd.models import MODEL_ZOO
    from triton_client_amd.ops import conv

    model = MODEL_ZOO["resnet50"]()
    model.load(0)
    records = []

    def timed(name, fn, flops_bytes):
        def wrapper(*a, **k):
            out = fn(*a, **k)  # first call: shapes, result for the rest of the network
            for _ in range(3):
                fn(*a, **k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            desc, flops, nbytes, tiles = flops_bytes(out, *a, **k)
            records.append((name, desc, us, flops, nbytes, tiles))
            return out
        return wrapper

    def nb(*ts):
        return sum(t.numel() * t.element_size() for t in ts if t is not None)

    def d1(out, x, w, bias=None, residual=None, relu=False, stride=1, **k):
        px = out.shape[0] * out.shape[2] * out.shape[3]
        xin = px * w.shape[1] * 2 if stride > 1 else nb(x)
        return (f"1x1 {x.shape[2]}x{x.shape[3]} {w.shape[1]}->{w.shape[0]} s{stride}{'+r' if residual is not None else ''}",
                2 * px * w.shape[0] * w.shape[1], xin + nb(w, out, residual), -(-px // 128) * -(-w.shape[0] // 128))

    def d3(out, x, w, bias=None, residual=None, relu=False, stride=1, padding=0, **k):
        px = out.shape[0] * out.shape[2] * out.shape[3]
        kk = w.shape[1] * w.shape[2] * w.shape[3]
        return (f"3x3 {x.shape[2]}x{x.shape[3]} {w.shape[1]}->{w.shape[0]} s{stride}", 2 * px * w.shape[0] * kk,
                nb(x, w, out), -(-px // 128) * -(-w.shape[0] // 128))

    def dpw(out, x, w, bias, w2, bias2=None, residual=None, relu=True, stride=1, padding=0, **k):
        y = out[0] if isinstance(out, tuple) else out
        t3 = out[1] if isinstance(out, tuple) else None
        px = y.shape[0] * y.shape[2] * y.shape[3]
        kk = w.shape[1] * 9
        flops = 2 * px * (w.shape[0] * kk + w2.shape[0] * w2.shape[1])
        nw = k.get("next_weight")
        if nw is not None:
            flops += 2 * px * nw.shape[0] * nw.shape[1]
        return (f"fused 3x3 {x.shape[2]}x{x.shape[3]} {w.shape[1]}->{w.shape[0]}->{w2.shape[0]} s{stride}"
                f"{'+r' if residual is not None else ''}{'+proj' if k.get('proj_input') is not None else ''}"
                f"{'+next' + str(nw.shape[0]) if nw is not None else ''}", flops,
                nb(x, w, w2, y, residual, t3, k.get("proj_input"), nw), -(-px // 128))

    def dstem(out, x, *a, **k):
        px = x.shape[0] * 112 * 112
        return ("stem 7x7/2 + pool", 2 * px * 64 * 147, nb(x, out), -(-px // 128))

    def dtail(out, y, w, *a, **k):
        return ("avgpool + fc", 2 * y.shape[0] * w.shape[0] * w.shape[1], nb(y, w), 0)

    conv.conv1x1 = timed("conv1x1", conv.conv1x1, d1)
    conv.conv2d = timed("conv2d", conv.conv2d, d3)
    conv.conv2d_pointwise = timed("conv2d_pointwise", conv.conv2d_pointwise, dpw)
    conv.stem_conv_pool = timed("stem", conv.stem_conv_pool, dstem)
    conv.avgpool_linear = timed("tail", conv.avgpool_linear, dtail)
    x = torch.randn(args.batch, 3, 224, 224, device="cuda", dtype=torch.float16)
    out = torch.empty(args.batch, 1000, device="cuda", dtype=torch.float32)
    with torch.no_grad():
        model._forward_into(x, out)
    torch.cuda.synchronize()
    total = sum(r[2] for r in records)
    print(f"batch {args.batch}: {len(records)} kernels, {total:.1f} us summed")
    print(f"{'layer':58s} {'us':>7s} {'GFLOP':>7s} {'MB':>7s} {'TF/s':>6s} {'TB/s':>5s} {'tiles':>6s} {'share':>6s}")
    for name, desc, us, flops, nbytes, tiles in records:
        print(f"{desc:58s} {us:7.1f} {flops / 1e9:7.2f} {nbytes / 1e6:7.1f} {flops / us / 1e6:6.0f} "
              f"{nbytes / us / 1e6:5.2f} {tiles:6d} {100 * us / total:5.1f}%")


if __name__ == "__main__":
    main()

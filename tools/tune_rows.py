#!/usr/bin/env python3
"""Where the row-organised bottleneck kernel (``ops.conv.bottleneck_rows``) beats the chunked fused kernel on
ResNet-50's stage-1 blocks: both are timed on two streams at once (the server's two execution instances) for
every batch the dynamic batcher can form, and the winners are recorded in the tuned dispatch table under
``"rows"`` (key ``"n3,proj,batch"``; the row kernel must win by more than 2 %).

    python tools/tune_rows.py --table triton_client_amd/ops/conv_tuning_gfx950.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="triton_client_amd/ops/conv_tuning_gfx950.json")
    ap.add_argument("--out", default=None, help="where to write the updated table (default: gpurun_out/ copy)")
    ap.add_argument("--batches", default=",".join(str(b) for b in [1, 2, 4] + list(range(8, 129, 8))))
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch

    from triton_client_amd.ops import conv

    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream() for _ in range(args.streams)]

    def nhwc(t):
        return t.contiguous(memory_format=torch.channels_last)

    def timed(fn):
        graphs = []
        for st in streams:
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
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                for st, g in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        g.replay()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) * 1e6 / (3 * args.reps * len(streams))
            best = t if best is None else min(best, t)
        return best

    with open(args.table) as fh:
        table = json.load(fh)
    rows = {}
    h = w = 56
    for batch in [int(v) for v in args.batches.split(",")]:
        for n3, proj in ((64, 1), (64, 0), (128, 0)):
            x = nhwc(torch.randn(batch, 64, h, w, device=dev).half())
            w1 = nhwc((torch.randn(64, 64, 3, 3, device=dev) / 24.0).half())
            b1 = torch.randn(64, device=dev).half()
            b2 = torch.randn(256, device=dev).half()
            w3 = (torch.randn(n3, 256, device=dev) / 16.0).half()
            b3 = torch.randn(n3, device=dev).half()
            x0 = nhwc(torch.randn(batch, 64, h, w, device=dev).half()) if proj else None
            res = None if proj else nhwc(torch.randn(batch, 256, h, w, device=dev).half())
            wcat = (torch.randn(256, 128 if proj else 64, device=dev) / 8.0).half()
            wperm = conv.permute_mid_k(wcat)
            out = nhwc(torch.empty(batch, 256, h, w, device=dev, dtype=torch.float16))
            mode = conv.tuned_block_mode(h, w, 64, 64, 256, 1, n3, batch)
            tile = int(mode) if mode in ("64", "128") else 0
            t_old = timed(lambda: conv.conv2d_pointwise(x, w1, b1, wcat, b2, res, relu=True, padding=1, out=out,
                                                        next_weight=w3, next_bias=b3, proj_input=x0, tile_m=tile))
            t_new = timed(lambda: conv.bottleneck_rows(x, w1, b1, wperm, b2, res, w3, b3, proj_input=x0, out=out))
            win = t_new < 0.98 * t_old
            rows[f"{n3},{proj},{batch}"] = bool(win)
            print(f"batch {batch:3d} n3={n3:3d} proj={proj}: chunked {t_old:7.2f} us, rows {t_new:7.2f} us -> "
                  f"{'rows' if win else 'chunked'}", flush=True)
    table["rows"] = rows
    out = args.out or os.path.join("gpurun_out", os.path.basename(args.table))
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    print(f"wrote {out}: rows wins {sum(rows.values())} of {len(rows)}")


if __name__ == "__main__":
    main()

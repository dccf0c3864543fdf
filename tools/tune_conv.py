#!/usr/bin/env python3
"""Builds the tuned dispatch table of the MFMA convolution kernels for ResNet-50's layers on the
GPU it runs on: for every batch size the dynamic batcher can form (multiples of 8 up to 128) and
every distinct layer, the fastest tile variant; for every bottleneck block whose 3x3 convolution
and 1x1 expansion can run as one kernel, whether to fuse and with which tile height.  The kernels'
built-in rules pick tiles from thresholds fitted at batches 8 / 32 / 128; between those the number
of tiles regularly lands just past a multiple of the CU count (batch 88: 27 % slower per image
than batch 80) and a smaller tile is the better choice - this table records what was measured.

    python tools/tune_conv.py --out triton_client_amd/ops/conv_tuning_gfx950.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

VARIANTS = ("128x128w8", "128x64", "64x64", "128x64w8", "64x64w8")
# Wider tiles (more FLOP per operand byte a CU takes in).  Alone on the GPU they win single layers by 3-10 %, but a
# table that includes them measured 1-2.5 % slower over the whole forward pass and no different under serving load
# (their few, long blocks leave CUs idle that the 128x128 launch fills): opt-in with --wide.
WIDE_VARIANTS = ("128x256w8", "256x128w16", "128x256w16", "256x256w16")
HALO_VARIANTS = ("128x256halo", "128x128halo")  # plain 3x3 / stride 1 only; refused where the halo exceeds LDS
# (H, W, Cin, Mid, Cout, stride, next 1x1 outputs or 0) of the blocks the fused kernel covers
FUSED_BLOCKS = [(56, 56, 64, 64, 256, 1, 64), (56, 56, 64, 64, 256, 1, 128), (56, 56, 128, 128, 512, 2, 128),
                (28, 28, 128, 128, 512, 1, 128), (28, 28, 128, 128, 512, 1, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/conv_tuning_gfx950.json")
    ap.add_argument("--batches", default=",".join(str(b) for b in range(8, 129, 8)))
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--streams", type=int, default=1,
                    help="time every candidate on this many streams at once (2: the server's two execution instances)")
    ap.add_argument("--stagings", action="store_true",
                    help="also try every tile with two and with three LDS stages (measured: the table it gives was "
                         "1.5 %% slower under the headline load than the one without)")
    ap.add_argument("--wide", action="store_true", help="also try the 256-wide and 16-wavefront tiles")
    ap.add_argument("--sustain", type=float, default=0.0,
                    help="seconds every candidate is replayed for (with --streams > 1); only the second half is timed.  "
                         "Under a sustained load the GPU is power-limited (2.1 instead of 2.4 GHz under the serving load): "
                         "what a tile costs then includes what its energy does to the clock")
    args = ap.parse_args()
    variants = VARIANTS + (WIDE_VARIANTS if args.wide else ())
    import torch

    from conv_bench import LAYERS, LAYERS_3X3
    from triton_client_amd.ops import conv

    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()

    streams = [torch.cuda.Stream() for _ in range(max(1, args.streams))]

    def graph_time_concurrent(fn):
        """us per call with the same work replayed on every stream at once: what a layer costs while another
        execution instance keeps the rest of the GPU busy (the condition the server runs under)."""
        import time

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
        def replay_all():
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()

        if args.sustain > 0:
            # first half: the clock settles; second half: timed.  A synchronisation every 20 rounds keeps the
            # launch queue short without draining the GPU for long.
            for timed in (False, True):
                rounds = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < args.sustain / 2:
                    for _ in range(20):
                        replay_all()
                    rounds += 20
                    torch.cuda.synchronize()
                elapsed = time.perf_counter() - t0
            return elapsed * 1e6 / (rounds * args.reps * len(streams))
        rounds = 6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            replay_all()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / (rounds * args.reps * len(streams))

    def graph_time(fn):
        if len(streams) > 1:
            return graph_time_concurrent(fn)
        with torch.cuda.stream(stream):
            fn()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(args.reps):
                    fn()
            g.replay()
            stream.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(3):
                g.replay()
            b.record(stream)
            b.synchronize()
        return a.elapsed_time(b) * 1e3 / (3 * args.reps)

    def nhwc(t):
        return t.contiguous(memory_format=torch.channels_last)

    layers = [(1, s, h, w, cin, cout, r, relu) for (h, w, cin, cout, s, r, relu) in LAYERS] + \
             [(3, s, h, w, c, c, False, True) for (h, w, c, s) in LAYERS_3X3]
    table = {"device": torch.cuda.get_device_name(0), "conv": {}, "fused": {}}
    best_time = {}
    for batch in [int(v) for v in args.batches.split(",")]:
        for (k, stride, h, w, cin, cout, with_res, relu) in layers:
            pad = k // 2
            oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            x = nhwc(torch.randn(batch, cin, h, w, device=dev).half())
            wt = nhwc((torch.randn(cout, cin, k, k, device=dev) / (k * k * cin) ** 0.5).half())
            bias = torch.randn(cout, device=dev).half()
            res = nhwc(torch.randn(batch, cout, oh, ow, device=dev).half()) if with_res else None
            out = nhwc(torch.empty(batch, cout, oh, ow, device=dev, dtype=torch.float16))
            times = {}
            for variant in variants + (HALO_VARIANTS if args.wide and k == 3 and stride == 1 else ()):
                if ("x128" in variant and cout % 128) or ("x256" in variant and cout % 256):
                    continue
                if variant in HALO_VARIANTS and ((cin // 64) * (((128 + 2 * w + 2) // 64 + 1) * 64)
                                                 + 2 * int(variant[4:7])) * 128 > 160 * 1024:
                    continue  # halo rows + two weight stages do not fit the CU's LDS
                times[variant] = graph_time(lambda v=variant: conv.conv2d(x, wt, bias, res, relu=relu, stride=stride,
                                                                          padding=pad, variant=v, out=out))
                if args.stagings and variant not in HALO_VARIANTS:
                    # the number of LDS stages pinned (the kernel's rule picks it from the grid size)
                    for staging in ("lds2", "lds3"):
                        times[f"{variant}/{staging}"] = graph_time(
                            lambda v=variant, sg=staging: conv.conv2d(x, wt, bias, res, relu=relu, stride=stride,
                                                                      padding=pad, variant=v, out=out, staging=sg))
            auto = graph_time(lambda: conv.conv2d(x, wt, bias, res, relu=relu, stride=stride, padding=pad,
                                                  variant=0, out=out, tuned=False))
            best = min(times, key=times.get)
            key = f"{k},{stride},{h},{w},{cin},{cout},{int(with_res)},{batch}"
            # keep the built-in rule unless the measured winner is clearly (3 %) faster: no noise in the table
            if times[best] < 0.97 * auto:
                table["conv"][key] = best
                best_time[key] = times[best]
            else:
                best_time[key] = auto
            print(f"b{batch:3d} k{k} s{stride} {h}x{w} {cin}->{cout}{'+r' if with_res else ''}: rule {auto:7.2f} us, best {best} "
                  f"{times[best]:7.2f} us{'  *' if key in table['conv'] else ''}", flush=True)
        for (h, w, cin, mid, cout, stride, n3) in FUSED_BLOCKS:
            oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
            x = nhwc(torch.randn(batch, cin, h, w, device=dev).half())
            w1 = nhwc((torch.randn(mid, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).half())
            b1 = torch.randn(mid, device=dev).half()
            w2 = (torch.randn(cout, mid, device=dev) / mid ** 0.5).half()
            b2 = torch.randn(cout, device=dev).half()
            res = nhwc(torch.randn(batch, cout, oh, ow, device=dev).half())
            out = nhwc(torch.empty(batch, cout, oh, ow, device=dev, dtype=torch.float16))
            w3 = (torch.randn(n3, cout, device=dev) / cout ** 0.5).half() if n3 else None
            b3 = torch.randn(n3, device=dev).half() if n3 else None
            separate = best_time[f"3,{stride},{h},{w},{cin},{mid},0,{batch}"] + best_time[f"1,1,{oh},{ow},{mid},{cout},1,{batch}"]
            if n3:
                separate += best_time[f"1,1,{oh},{ow},{cout},{n3},0,{batch}"]
            times = {"none": separate}
            for bm in (64, 128):
                times[str(bm)] = graph_time(lambda bm=bm: conv.conv2d_pointwise(
                    x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1, out=out, next_weight=w3, next_bias=b3,
                    tile_m=bm))
            best = min(times, key=times.get)
            table["fused"][f"{h},{w},{cin},{mid},{cout},{stride},{n3},{batch}"] = best
            print(f"b{batch:3d} block {h}x{w} {cin}->{mid}->{cout}{'->' + str(n3) if n3 else ''} s{stride}: "
                  + ", ".join(f"{k} {v:7.2f}" for k, v in times.items()) + f"  -> {best}", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    print(f"{len(table['conv'])} convolution entries differ from the built-in rule; {len(table['fused'])} block entries -> {args.out}")


if __name__ == "__main__":
    main()

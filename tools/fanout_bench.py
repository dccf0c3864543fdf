#!/usr/bin/env python3
"""Transport cost of the sharded group: a batch of fp16 images leaves the ingest GPU for N-1
replica processes and fp32 logit rows come back, with replicas that compute nothing.

  python tools/fanout_bench.py --gpus 8 --batch 256            # one replica per GPU
  python tools/fanout_bench.py --devices 0,0 --batch 96        # rehearsal: three replicas on GPU 0

Prints one JSON line per transport ("direct": peer stores + device flags, "rccl": point-to-
point operations of torch.distributed) with the time per step, the bytes per second that
leave the ingest GPU and that figure per xGMI link (one link per peer); for "direct" also the bytes
per second of the copy kernel into each peer's inbox by itself ("peer_GBps", one entry per replica
process) and into all of them in one launch, and before the runs the peer-access matrix of the GPUs
involved (hipDeviceCanAccessPeer)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="replicas = GPUs 0..N-1")
    ap.add_argument("--devices", default=None, help="GPU ordinals of the replicas besides GPU 0")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--modes", default="direct,rccl")
    ap.add_argument("--gather", default=None, choices=[None, "p2p", "collective"], help="mode rccl: how output rows return")
    ap.add_argument("--flags", default=None, choices=[None, "kernel", "stream"])
    args = ap.parse_args()
    import torch

    from triton_client_amd.parallel.sharded import SLOTS, ShardRoot

    if args.devices is not None:
        devices = [int(v) for v in args.devices.split(",") if v.strip() != ""]
    else:
        devices = list(range(1, args.gpus or torch.cuda.device_count()))
    if not devices:
        raise SystemExit("needs at least one replica besides GPU 0 (--gpus N or --devices)")
    world = 1 + len(devices)
    same_gpu = len(set(devices + [0])) < world
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sample_bytes, out_row_bytes = 3 * 224 * 224 * 2, 1000 * 4
    x = torch.randn(args.batch, 3, 224, 224, device=dev).half()
    out = torch.zeros(args.batch, 1000, device=dev)
    gpus = sorted(set([0] + devices))
    print(json.dumps({"peer_access": {"gpus": gpus, "matrix": [[1 if a == b else int(torch.cuda.can_device_access_peer(a, b))
                                                                for b in gpus] for a in gpus]}}), flush=True)
    for mode in [m for m in args.modes.split(",") if m]:
        if mode == "rccl" and same_gpu:
            print(json.dumps({"mode": mode, "skipped": "RCCL needs one GPU per rank"}))
            continue
        streams = [torch.cuda.Stream(device=dev) for _ in range(SLOTS)]
        root = ShardRoot(lambda a, b, slot: None, world, 0, sample_bytes, out_row_bytes, args.batch, streams=streams,
                         mode=mode, flags=args.flags, timeout_ms=20000, gather=args.gather)
        try:
            root.spawn_workers(devices, model="test_noop", startup_timeout=300.0)

            def run(n):
                for i in range(n):
                    root.step(x, out, i % SLOTS)
                for s in streams:
                    s.synchronize()

            run(20)
            t0 = time.perf_counter()
            run(args.iters)
            dt = (time.perf_counter() - t0) / args.iters
            # latency of one isolated step (nothing else in flight)
            lat = []
            for _ in range(20):
                t1 = time.perf_counter()
                run(1)
                lat.append(time.perf_counter() - t1)
            root.check()
            peers = None
            if mode == "direct":
                shard = -(-args.batch // world) * sample_bytes
                peers = root.peer_bandwidth(x.data_ptr(), shard, iters=50)
        finally:
            root.close()
        sent = (args.batch - -(-args.batch // world)) * sample_bytes  # bytes that leave the ingest GPU per step
        print(json.dumps({"mode": mode, "replicas": world, "devices": [0] + devices, "same_gpu": same_gpu,
                          "batch": args.batch, "us_per_step_pipelined": round(dt * 1e6, 1),
                          "us_per_step_isolated_p50": round(sorted(lat)[len(lat) // 2] * 1e6, 1),
                          "GBps_out_of_root": round(sent / dt / 1e9, 1),
                          "GBps_per_link": round(sent / dt / 1e9 / len(devices), 1),
                          **({"peer_bytes": peers["bytes"],
                              "peer_GBps": [round(v / 1e9, 1) for v in peers["per_peer"]],
                              "all_peers_GBps": round(peers["all_peers"] / 1e9, 1)} if peers else {})}), flush=True)


if __name__ == "__main__":
    main()

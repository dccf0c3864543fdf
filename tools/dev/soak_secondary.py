#!/usr/bin/env python3
"""The secondary serving configurations (tools/secondary_bench.py) held for a while each against ONE server:
per window throughput and the server's resident set, file descriptors and threads.  Stateful models, streams,
BYTES tensors and shared-memory registration churn are where a serving process usually leaks.

    python tools/dev/soak_secondary.py [--gpu 0] [--windows 6] [--window-ms 5000]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=int, default=None)
    ap.add_argument("--windows", type=int, default=6)
    ap.add_argument("--window-ms", type=int, default=5000)
    args = ap.parse_args()
    import contextlib
    import io

    import psutil

    from triton_client_amd.perf import main as perf
    from triton_client_amd.server import spawn_server

    shm = "hip" if args.gpu is not None else "system"
    where = ["--devices", str(args.gpu)] if args.gpu is not None else []
    with spawn_server(models=["simple", "simple_string", "repeat_int32", "simple_sequence"], gpu=args.gpu) as srv:
        server = psutil.Process(srv.proc.pid)
        common = ["-p", str(args.window_ms), "--stability-windows", "1", "--warmup-ms", "200", "--native", "--concurrency-range", "16"]
        configs = [
            ("simple, HTTP + system shm", ["-m", "simple", "-u", srv.http_url, "-i", "http", "--shared-memory", "system"]),
            ("simple, gRPC on the wire", ["-m", "simple", "-u", srv.grpc_url, "-i", "grpc"]),
            (f"simple_string, gRPC + {shm} shm", ["-m", "simple_string", "-u", srv.grpc_url, "-i", "grpc", "--shared-memory", shm] + where),
            ("repeat_int32, decoupled streams", ["-m", "repeat_int32", "-u", srv.grpc_url, "-i", "grpc", "--streaming", "--shape", "IN:16",
                                                 "--shape", "DELAY:16", "--shape", "WAIT:1"]),
            ("simple_sequence, streams of 20", ["-m", "simple_sequence", "-u", srv.grpc_url, "-i", "grpc", "--streaming",
                                                "--sequence-length", "20"]),
        ]
        print(f"{'configuration':34s} {'window':>6s} {'per second':>11s} {'p99 us':>8s} {'server RSS MB':>14s} {'fds':>5s} {'threads':>7s}", flush=True)
        for label, extra in configs:
            for w in range(args.windows):
                with contextlib.redirect_stdout(io.StringIO()):
                    rows = perf(extra + common)
                r = rows[0]
                rate = r.get("responses_per_sec") if "decoupled" in label else r["infer_per_sec"]
                print(f"{label:34s} {w:6d} {rate:11.0f} {r['p99_us']:8.0f} {server.memory_info().rss / 2 ** 20:14.1f} "
                      f"{server.num_fds():5d} {server.num_threads():7d}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak run of the serving path: N measurement rounds of the native load generator against one
server process (gRPC, alternating HIP shared-memory and wire requests), printing the server's
resident set, its GPU memory and the throughput after every round.  A server that leaks per
request (host buffers, pinned ring entries, graphs, IPC mappings) shows up as a rising column."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def rss_mib(pid):
    with open(f"/proc/{pid}/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
    return float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--window-ms", type=int, default=4000)
    ap.add_argument("--concurrency", type=int, default=64)
    args = ap.parse_args()
    import torch

    from triton_client_amd.perf import main as perf
    from triton_client_amd.server import spawn_server

    with spawn_server(models=["resnet50"], gpu=0) as srv:
        common = ["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "-p", str(args.window_ms),
                  "--stability-windows", "2", "--warmup-ms", "300"]
        for r in range(args.rounds):
            mode = "hip" if r % 2 == 0 else "none"
            c = args.concurrency if mode == "hip" else max(1, args.concurrency // 2)
            rows = perf(common + ["--shared-memory", mode, "--concurrency-range", f"{c}:{c}"])
            free, total = torch.cuda.mem_get_info(0)
            print(json.dumps({"round": r, "shared_memory": mode, "concurrency": c,
                              "infer_per_sec": round(rows[-1]["infer_per_sec"], 1),
                              "server_rss_mib": round(rss_mib(srv.proc.pid), 1),
                              "gpu_used_mib": round((total - free) / 2**20, 1)}), flush=True)


if __name__ == "__main__":
    main()

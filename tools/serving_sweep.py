#!/usr/bin/env python3
"""Concurrency sweep of the ResNet-50 serving path on one GPU: gRPC with HIP
shared-memory I/O (1..256, x2) against the same requests carried on the wire
(1..64, x2), both driven by the C++ load generator through the analyzer's
``--native`` mode.  Writes one JSON table per configuration."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/profiles")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--hip-range", default="1:256:2")
    ap.add_argument("--wire-range", default="1:64:2")
    ap.add_argument("--window-ms", type=int, default=1500)
    args = ap.parse_args()
    from triton_client_amd.perf import main as perf
    from triton_client_amd.server import spawn_server

    os.makedirs(args.out, exist_ok=True)
    with spawn_server(models=["resnet50"], gpu=0) as srv:
        common = ["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", str(args.batch), "--native",
                  "--multiplicative-step", "-p", str(args.window_ms), "--stability-windows", "2", "--warmup-ms", "500"]
        hip = perf(common + ["--shared-memory", "hip", "--concurrency-range", args.hip_range,
                             "--json", os.path.join(args.out, "sweep_grpc_hipshm.json")])
        wire = perf(common + ["--shared-memory", "none", "--concurrency-range", args.wire_range,
                              "--json", os.path.join(args.out, "sweep_grpc_wire.json")])
    best_hip = max(r["infer_per_sec"] for r in hip)
    best_wire = max(r["infer_per_sec"] for r in wire)
    print(json.dumps({"best_hipshm_infer_per_sec": best_hip, "best_wire_infer_per_sec": best_wire,
                      "hipshm_over_wire": round(best_hip / best_wire, 2) if best_wire else None}))


if __name__ == "__main__":
    main()

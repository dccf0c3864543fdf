#!/usr/bin/env python3
"""ResNet-50 over gRPC + HIP shared memory at low concurrency (1 ... 32 requests of batch 8 in flight): how the
batch-forming policy (environment of the server: TCA_ADAPTIVE_CAP, TCA_FILL_WINDOW_US, ...) moves throughput and
latency where the GPU is not saturated."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf  # noqa: E402
from triton_client_amd.server import spawn_server  # noqa: E402

os.environ.setdefault("TCA_PERF_CONNECTIONS", "8")
levels = sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,6,8,12,16,24,32"
env = {k: v for k, v in os.environ.items() if k.startswith("TCA_")}
with spawn_server(models=["resnet50"], gpu=0, env=env) as srv:
    for c in levels.split(","):
        rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "--shared-memory", "hip",
                     "--concurrency-range", c, "-p", "1200", "--stability-windows", "2", "--warmup-ms", "300"])
        r = rows[0]
        print(json.dumps({"concurrency": int(c), "infer_per_sec": round(r["infer_per_sec"]), "p50_us": round(r["p50_us"]),
                          "p99_us": round(r["p99_us"])}), flush=True)

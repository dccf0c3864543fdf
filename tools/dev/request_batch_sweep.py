#!/usr/bin/env python3
"""ResNet-50 over gRPC + HIP shared memory at request batch sizes other than the headline's 8: batch 1
(perf_analyzer's default) at 256 requests in flight, batch 32 at 16, batch 8 at 1 - and the request rate
at batch 1 / 2 / 4 / 8 with 64 in flight.  One JSON line per run; the server's counters for the native
graph path (batches, padding, online captures) at the end."""
import json
import os
import sys
import urllib.request

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf  # noqa: E402
from triton_client_amd.server import spawn_server  # noqa: E402

os.environ.setdefault("TCA_PERF_CONNECTIONS", "8")
with spawn_server(models=["resnet50"], gpu=0) as srv:
    for b, c in ((1, 256), (32, 16), (8, 1), (1, 64), (2, 64), (4, 64), (8, 64)):
        rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", str(b), "--native", "--shared-memory", "hip",
                     "--concurrency-range", str(c), "-p", "2000", "--stability-windows", "2", "--warmup-ms", "500"])
        r = rows[0]
        print(json.dumps({"request_batch": b, "concurrency": c, "infer_per_sec": round(r["infer_per_sec"]),
                          "requests_per_sec": round(r["infer_per_sec"] / b), "p50_us": round(r["p50_us"]),
                          "p99_us": round(r["p99_us"])}), flush=True)
    with urllib.request.urlopen(f"http://{srv.http_url}/metrics", timeout=30) as resp:
        rows = [ln for ln in resp.read().decode().splitlines() if ln.startswith(("tca_native_graph_", "tca_graph_captures"))]
    print(json.dumps({"metrics": rows}), flush=True)

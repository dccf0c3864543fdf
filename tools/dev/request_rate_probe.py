#!/usr/bin/env python3
"""Requests per second the ResNet-50 serving path sustains when the GPU work per request shrinks (batch 8 -> 1 per
request, gRPC + hipshm, closed loop): if requests/s stays where it is at batch 8, the request path - not the GPU -
sets the headline."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

conc = sys.argv[1] if len(sys.argv) > 1 else "64"
with spawn_server(models=["resnet50"], gpu=0) as srv:
    for b in (8, 4, 2, 1):
        rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", str(b), "--native",
                     "--shared-memory", "hip", "--concurrency-range", conc, "-p", "1500", "--stability-windows", "2",
                     "--warmup-ms", "500"])
        r = rows[0]
        print(f"RESULT batch {b} per request, concurrency {conc}: {r['infer_per_sec']:.0f} img/s = "
              f"{r['infer_per_sec'] / b:.0f} requests/s, p50 {r['p50_us']:.0f} us", flush=True)

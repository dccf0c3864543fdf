#!/usr/bin/env python3
"""Is the wire path bound by the load generator or by the server?  The same total concurrency
driven by 1, 2 and 4 generator processes (the endpoint listed several times)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

with spawn_server(models=["resnet50"], gpu=0) as srv:
    for level in (16, 32):
        for procs in (1, 2, 4):
            rows = perf(["-m", "resnet50", "--urls", ",".join([srv.grpc_url] * procs), "-i", "grpc", "-b", "8",
                         "--native",
                         "--shared-memory", "none", "--concurrency-range", str(level), "-p", "1000",
                         "--stability-windows", "2", "--warmup-ms", "400"])
            r = rows[0]
            print("RESULT c=%d generators=%d: %.0f infer/s p50 %.0f us | server queue %.0f in %.0f infer %.0f" % (
                level, procs, r["infer_per_sec"], r["p50_us"], r["server_queue_us"], r["server_compute_input_us"],
                r["server_compute_infer_us"]), flush=True)

#!/usr/bin/env python3
"""Which merged batch sizes the dynamic batcher forms under closed-loop load (ResNet-50, gRPC + hipshm,
batch 8 per request), from the server's per-batch-size statistics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tritonclient.grpc as grpcclient

from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

conc = sys.argv[1] if len(sys.argv) > 1 else "64"
with spawn_server(models=["resnet50"], gpu=0) as srv:
    rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "--shared-memory", "hip",
                 "--concurrency-range", conc, "-p", "2000", "--stability-windows", "2", "--warmup-ms", "500"])
    with grpcclient.InferenceServerClient(srv.grpc_url) as c:
        st = c.get_inference_statistics("resnet50").model_stats[0]
    total = sum(b.compute_infer.count for b in st.batch_stats)
    print(f"RESULT c{conc}: {rows[0]['infer_per_sec']:.0f} img/s; executions {st.execution_count}, inferences {st.inference_count}")
    for b in sorted(st.batch_stats, key=lambda b: b.batch_size):
        n = b.compute_infer.count
        if n:
            print(f"RESULT   batch {b.batch_size:4d}: {n:6d} ({100.0 * n / total:5.1f} %)  infer {b.compute_infer.ns / n / 1e3:8.1f} us")

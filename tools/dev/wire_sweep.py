#!/usr/bin/env python3
"""ResNet-50 with tensors on the wire (gRPC, batch 8, concurrency 64) against the number of
client connections and server reactors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

for reactors, conns in ((2, 4), (2, 8), (4, 8), (4, 16)):
    os.environ["TCA_PERF_CONNECTIONS"] = str(conns)
    with spawn_server(models=["resnet50"], gpu=0, env={"TCA_GRPC_REACTORS": str(reactors)}) as srv:
        rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "--shared-memory",
                     "none",
                     "--concurrency-range", "64", "-p", "1500", "--stability-windows", "2", "--warmup-ms", "500"])
        print(f"RESULT reactors={reactors} connections={conns}: {rows[0]['infer_per_sec']:.0f} img/s p50 {rows[0]['p50_us']:.0f} us",
              flush=True)

#!/usr/bin/env python3
"""ResNet-50 gRPC + hipshm at low / mid concurrency against reactor count and client connections."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

for reactors, conns, inline in ((2, 4, 1), (2, 4, 0), (1, 4, 1), (2, 4, 1), (2, 4, 0), (1, 4, 1)):
    os.environ["TCA_PERF_CONNECTIONS"] = str(conns)
    with spawn_server(models=["resnet50"], gpu=0, env={"TCA_GRPC_REACTORS": str(reactors),
                                                        "TCA_FUNNEL_INLINE": str(inline)}) as srv:
        rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "--shared-memory",
                     "hip",
                     "--concurrency-range", "1:64:4", "--multiplicative-step", "-p", "1000", "--stability-windows", "2",
                     "--warmup-ms", "400"])
        print(f"RESULT reactors={reactors} connections={conns} inline={inline}: " +
              "  ".join(f"c{r['level']}={r['infer_per_sec']:.0f}/{r['p50_us']:.0f}us" for r in rows), flush=True)

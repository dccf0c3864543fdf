#!/usr/bin/env python3
"""ResNet-50 over REST (HTTP/1.1, binary tensor data inside the request bodies, batch 8) at a few concurrencies."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from triton_client_amd.server import spawn_server  # noqa: E402

LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")
levels = (sys.argv[1] if len(sys.argv) > 1 else "1,8,32,64,128").split(",")
conns = sys.argv[2] if len(sys.argv) > 2 else "8"
with spawn_server(models=["resnet50"], gpu=0, env={k: v for k, v in os.environ.items() if k.startswith("TCA_")}) as srv:
    for c in levels:
        cmd = [LOADGEN, "-u", srv.http_url, "-i", "http", "-m", "resnet50", "-c", c, "-b", "8", "--connections", conns,
               "-d", "2.5", "--warmup", "0.8", "--shape", "input:FP16:8,3,224,224"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("no result", out.stdout[-300:], out.stderr[-300:], flush=True)
            continue
        r = json.loads(line[-1])
        print(json.dumps({"concurrency": int(c), "connections": int(conns), "infer_per_sec": round(r["infer_per_sec"]),
                          "p50_us": round(r["p50_us"]), "p99_us": round(r["p99_us"]), "errors": r["errors"]}), flush=True)

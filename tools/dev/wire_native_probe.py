#!/usr/bin/env python3
"""ResNet-50 with the tensors inside the gRPC messages (batch 8) over a range of concurrencies, the load
generator in its timed mode: throughput, latency percentiles and the server's counters for the native
graph path per level.  ``TCA_NATIVE_WIRE_ARENA_MB=0`` in the environment shows the general path."""
import json
import os
import subprocess
import sys
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from triton_client_amd.server import spawn_server  # noqa: E402

LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")


def stats(srv):
    import tritonclient.grpc as grpcclient

    with grpcclient.InferenceServerClient(srv.grpc_url) as c:
        row = c.get_inference_statistics("resnet50", as_json=True)["model_stats"][0]["inference_stats"]
    out = {"n": int(row.get("success", {}).get("count", 0))}
    for k in ("queue", "compute_input", "compute_infer", "compute_output"):
        out[k] = int(row.get(k, {}).get("ns", 0))
    return out


def main():
    levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64,128,256").split(",")]
    connections = sys.argv[2] if len(sys.argv) > 2 else "4"
    reactors = sys.argv[3] if len(sys.argv) > 3 else "8"
    mode = sys.argv[4] if len(sys.argv) > 4 else "wire"
    with spawn_server(models=["resnet50"], gpu=0, env={k: v for k, v in os.environ.items() if k.startswith("TCA_")}, extra_args=["--grpc-workers", "64", "--grpc-reactors", reactors]) as srv:
        for c in levels:
            cmd = [LOADGEN, "-u", srv.grpc_url, "-i", "grpc", "-m", "resnet50", "-c", str(c), "-b", "8", "--connections",
                   connections, "-d", "2.5", "--warmup", "0.8"]
            regions = []
            if mode == "system":
                import tritonclient.grpc as grpcclient
                import tritonclient.utils.shared_memory as sysshm
                import numpy as np

                ctl = grpcclient.InferenceServerClient(srv.grpc_url)
                ctl.unregister_system_shared_memory()
                x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
                for i in range(c):
                    hin = sysshm.create_shared_memory_region(f"pin{i}", f"/tca_probe_in{i}", x.nbytes)
                    hout = sysshm.create_shared_memory_region(f"pout{i}", f"/tca_probe_out{i}", 32000)
                    sysshm.set_shared_memory_region(hin, [x])
                    ctl.register_system_shared_memory(f"pin{i}", f"/tca_probe_in{i}", x.nbytes)
                    ctl.register_system_shared_memory(f"pout{i}", f"/tca_probe_out{i}", 32000)
                    regions += [hin, hout]
                cmd += ["--shm-input", f"input:FP16:8,3,224,224:pin{{slot}}:{x.nbytes}", "--shm-output", "output:pout{slot}:32000"]
            else:
                cmd += ["--shape", "input:FP16:8,3,224,224"]
            before = stats(srv)
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            after = stats(srv)
            if mode == "system":
                ctl.unregister_system_shared_memory()
                ctl.close()
                for r in regions:
                    sysshm.destroy_shared_memory_region(r)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print("no result", out.stdout[-500:], out.stderr[-500:], flush=True)
                continue
            r = json.loads(line[-1])
            with urllib.request.urlopen(f"http://{srv.http_url}/metrics", timeout=30) as resp:
                metrics = {ln.split("{")[0]: ln.rsplit(" ", 1)[1] for ln in resp.read().decode().splitlines()
                           if ln.startswith("tca_native_graph_")}
            print(json.dumps({"concurrency": c, "connections": int(connections), "reactors": int(reactors), "infer_per_sec": round(r["infer_per_sec"]),
                              "p50_us": round(r["p50_us"]), "p99_us": round(r["p99_us"]), "errors": r["errors"],
                              "mode": mode, "per_request_us": {k: round((after[k] - before[k]) / max(1, after["n"] - before["n"]) / 1e3, 1)
                                                               for k in ("queue", "compute_input", "compute_infer", "compute_output")},
                              "images_per_batch": round(float(metrics.get("tca_native_graph_images_total", 0)) / max(1.0, float(
                                  metrics.get("tca_native_graph_batches_total", 1))), 1),
                              "counters_since_start": {k[len("tca_native_graph_"):]: float(v) for k, v in metrics.items()}}), flush=True)


if __name__ == "__main__":
    main()

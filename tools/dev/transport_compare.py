#!/usr/bin/env python3
"""One server, one box: ResNet-50 batch-8 requests with tensors in HIP shared memory (gRPC), inside gRPC messages and
inside REST bodies, at the same concurrencies, with the server's graph-path counters per run."""
import json
import os
import subprocess
import sys
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from triton_client_amd.server import spawn_server  # noqa: E402

LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")
levels = (sys.argv[1] if len(sys.argv) > 1 else "64,128").split(",")


def counters(srv):
    with urllib.request.urlopen(f"http://{srv.http_url}/metrics", timeout=30) as resp:
        rows = {ln.split("{")[0]: float(ln.rsplit(" ", 1)[1]) for ln in resp.read().decode().splitlines()
                if ln.startswith("tca_native_graph_")}
    return rows


with spawn_server(models=["resnet50"], gpu=0, env={k: v for k, v in os.environ.items() if k.startswith("TCA_")}) as srv:
    import numpy as np
    import tritonclient.grpc as grpcclient
    import tritonclient.utils.cuda_shared_memory as hipshm

    ctl = grpcclient.InferenceServerClient(srv.grpc_url)
    x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
    regions = []
    for i in range(max(int(c) for c in levels)):
        hin = hipshm.create_shared_memory_region(f"tin{i}", x.nbytes, 0)
        hout = hipshm.create_shared_memory_region(f"tout{i}", 32000, 0)
        hipshm.set_shared_memory_region(hin, [x])
        ctl.register_cuda_shared_memory(f"tin{i}", hipshm.get_raw_handle(hin), 0, x.nbytes)
        ctl.register_cuda_shared_memory(f"tout{i}", hipshm.get_raw_handle(hout), 0, 32000)
        regions += [hin, hout]
    n_slots = max(int(c) for c in levels)
    in_stride, out_stride = (x.nbytes + 255) // 256 * 256, 32000
    big_in = hipshm.create_shared_memory_region("tbig_in", n_slots * in_stride, 0)
    big_out = hipshm.create_shared_memory_region("tbig_out", n_slots * out_stride, 0)
    for i in range(n_slots):
        hipshm.set_shared_memory_region(big_in, [x], offset=i * in_stride)
    ctl.register_cuda_shared_memory("tbig_in", hipshm.get_raw_handle(big_in), 0, n_slots * in_stride)
    ctl.register_cuda_shared_memory("tbig_out", hipshm.get_raw_handle(big_out), 0, n_slots * out_stride)
    regions += [big_in, big_out]
    for rep in range(int(os.environ.get('REPS', '1'))):
        for c in levels:
            for name, proto, url, extra in (
                    ("hipshm grpc", "grpc", srv.grpc_url, ["--shm-input", f"input:FP16:8,3,224,224:tin{{slot}}:{x.nbytes}",
                                                           "--shm-output", "output:tout{slot}:32000"]),
                    ("hipshm grpc, ONE region per direction", "grpc", srv.grpc_url,
                     ["--shm-input", f"input:FP16:8,3,224,224:tbig_in:{x.nbytes}:{in_stride}*slot",
                      "--shm-output", f"output:tbig_out:32000:{out_stride}*slot"]),
                    ("wire grpc", "grpc", srv.grpc_url, ["--shape", "input:FP16:8,3,224,224"]),
                    ("wire http", "http", srv.http_url, ["--shape", "input:FP16:8,3,224,224"]),
                    ("in hipshm / out wire http", "http", srv.http_url, ["--shm-input", f"input:FP16:8,3,224,224:tin{{slot}}:{x.nbytes}"]),
                    ("in wire / out hipshm http", "http", srv.http_url, ["--shape", "input:FP16:8,3,224,224", "--shm-output",
                                                                         "output:tout{slot}:32000"]),
                    ("hipshm http", "http", srv.http_url, ["--shm-input", f"input:FP16:8,3,224,224:tin{{slot}}:{x.nbytes}",
                                                           "--shm-output", "output:tout{slot}:32000"])):
                before = counters(srv)
                t_before = time.perf_counter()
                cmd = [LOADGEN, "-u", url, "-i", proto, "-m", "resnet50", "-c", c, "-b", "8", "--connections", "8", "-d", "2.5",
                       "--warmup", "0.8"] + extra
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
                r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                after = counters(srv)
                wall = time.perf_counter() - t_before
                d = {k.replace("tca_native_graph_", "").replace("_total", ""): after[k] - before.get(k, 0) for k in after}
                print(json.dumps({"transport": name, "concurrency": int(c), "infer_per_sec": round(r["infer_per_sec"]),
                                  "p50_us": round(r["p50_us"]), "p99_us": round(r["p99_us"]), "errors": r["errors"],
                                  "server_images_per_wall_second_incl_startup": round(d["images"] / wall) if False else None,
                                  "server_images": int(after["tca_native_graph_images_total"] - before.get("tca_native_graph_images_total", 0)),
                                  "generator_wall_s": round(wall, 2),
                                  "images_per_batch": round(d["images"] / max(1, d["batches"]), 1),
                                  "padded_share": round(d["padded_images"] / max(1, d["images"]), 3)}), flush=True)
    ctl.unregister_cuda_shared_memory()
    ctl.close()
    for r in regions:
        hipshm.destroy_shared_memory_region(r)

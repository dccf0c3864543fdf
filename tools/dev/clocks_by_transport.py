#!/usr/bin/env python3
"""GPU clocks while the same ResNet-50 load arrives through HIP shared memory (random normal inputs) and inside REST
bodies with pseudo-random and with zero-filled inputs (128 requests of batch 8 in flight, one server).  A network
fed zeros switches fewer bits: the GPU, power-limited under this load, then clocks 14 % higher."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from triton_client_amd.server import spawn_server  # noqa: E402

LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")


def sample(stop, rows):
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks"], capture_output=True, text=True).stdout
        row = {}
        for ln in out.splitlines():
            for name in ("sclk", "mclk", "fclk", "socclk"):
                if f"{name} clock level" in ln and "(" in ln:
                    row[name] = ln.split("(")[-1].split(")")[0]
        rows.append(row)
        time.sleep(0.7)


def main():
    import numpy as np
    import tritonclient.grpc as grpcclient
    import tritonclient.utils.cuda_shared_memory as hipshm

    c = 128
    with spawn_server(models=["resnet50"], gpu=0) as srv:
        ctl = grpcclient.InferenceServerClient(srv.grpc_url)
        x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
        regions = []
        for i in range(c):
            hin = hipshm.create_shared_memory_region(f"tin{i}", x.nbytes, 0)
            hout = hipshm.create_shared_memory_region(f"tout{i}", 32000, 0)
            hipshm.set_shared_memory_region(hin, [x])
            ctl.register_cuda_shared_memory(f"tin{i}", hipshm.get_raw_handle(hin), 0, x.nbytes)
            ctl.register_cuda_shared_memory(f"tout{i}", hipshm.get_raw_handle(hout), 0, 32000)
            regions += [hin, hout]
        for rep in range(2):
            for name, proto, url, extra in (
                    ("hipshm grpc", "grpc", srv.grpc_url, ["--shm-input", f"input:FP16:8,3,224,224:tin{{slot}}:{x.nbytes}",
                                                           "--shm-output", "output:tout{slot}:32000"]),
                    ("wire http, pseudo-random values", "http", srv.http_url, ["--shape", "input:FP16:8,3,224,224"]),
                    ("wire http, zeros", "http", srv.http_url, ["--shape", "input:FP16:8,3,224,224", "--wire-data", "zero"])):
                stop, rows = threading.Event(), []
                t = threading.Thread(target=sample, args=(stop, rows))
                t.start()
                out = subprocess.run([LOADGEN, "-u", url, "-i", proto, "-m", "resnet50", "-c", str(c), "-b", "8", "--connections", "8",
                                      "-d", "6", "--warmup", "1.0"] + extra, capture_output=True, text=True, timeout=120)
                stop.set()
                t.join()
                r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                print(f"{name:32s} {r['infer_per_sec']:8.0f} infer/s  clocks: " +
                      " | ".join(",".join(f"{k}={v}" for k, v in row.items()) for row in rows[2:8]), flush=True)
        ctl.unregister_cuda_shared_memory()
        ctl.close()
        for r in regions:
            hipshm.destroy_shared_memory_region(r)


if __name__ == "__main__":
    main()

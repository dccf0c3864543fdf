#!/usr/bin/env python3
"""Churn instead of load: clients that connect, send a few requests and go away (HTTP and gRPC), shared-memory
regions registered and unregistered around every few requests, and models loaded and unloaded (explicit model
control) - with the server's resident set, file descriptors and threads after every round.

    python tools/dev/churn.py [--gpu 0] [--rounds 6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=6)
    args = ap.parse_args()
    import numpy as np
    import psutil
    import tritonclient.grpc as grpcclient
    import tritonclient.http as httpclient
    import tritonclient.utils.shared_memory as sysshm

    from triton_client_amd.server import spawn_server

    models = ["simple", "simple_string", "simple_sequence"] + (["resnet50"] if args.gpu is not None else [])
    with spawn_server(models=models, gpu=args.gpu, extra_args=["--model-control-mode", "explicit"]) as srv:
        server = psutil.Process(srv.proc.pid)
        ctl = grpcclient.InferenceServerClient(srv.grpc_url)
        for m in models:
            ctl.load_model(m)
        a = np.arange(16, dtype=np.int32).reshape(1, 16)
        b = np.ones((1, 16), dtype=np.int32)

        def report(label, r):
            gpu = ""
            if args.gpu is not None:
                import torch

                free, total = torch.cuda.mem_get_info(args.gpu)
                gpu = f"  GPU memory in use {(total - free) / 2 ** 20:8.0f} MB"
            print(f"{label:44s} round {r}: RSS {server.memory_info().rss / 2 ** 20:8.1f} MB  fds {server.num_fds():4d}  "
                  f"threads {server.num_threads():3d}{gpu}", flush=True)

        for r in range(args.rounds):
            for _ in range(300):  # gRPC clients that come and go
                c = grpcclient.InferenceServerClient(srv.grpc_url)
                i0, i1 = grpcclient.InferInput("INPUT0", [1, 16], "INT32"), grpcclient.InferInput("INPUT1", [1, 16], "INT32")
                i0.set_data_from_numpy(a)
                i1.set_data_from_numpy(b)
                for _ in range(3):
                    c.infer("simple", [i0, i1])
                c.close()
            report("300 gRPC clients x 3 requests", r)
        for r in range(args.rounds):
            for _ in range(300):
                c = httpclient.InferenceServerClient(srv.http_url)
                i0, i1 = httpclient.InferInput("INPUT0", [1, 16], "INT32"), httpclient.InferInput("INPUT1", [1, 16], "INT32")
                i0.set_data_from_numpy(a)
                i1.set_data_from_numpy(b)
                for _ in range(3):
                    c.infer("simple", [i0, i1])
                c.close()
            report("300 HTTP clients x 3 requests", r)
        for r in range(args.rounds):
            for k in range(200):
                name, key = f"churn{k % 7}", f"/tca_churn_{os.getpid()}_{k % 7}"
                h = sysshm.create_shared_memory_region(name, key, 256)
                sysshm.set_shared_memory_region(h, [a, b])
                ctl.register_system_shared_memory(name, key, 256)
                i0, i1 = grpcclient.InferInput("INPUT0", [1, 16], "INT32"), grpcclient.InferInput("INPUT1", [1, 16], "INT32")
                i0.set_shared_memory(name, 64, 0)
                i1.set_shared_memory(name, 64, 64)
                o0 = grpcclient.InferRequestedOutput("OUTPUT0")
                o0.set_shared_memory(name, 64, 128)
                ctl.infer("simple", [i0, i1], outputs=[o0])
                ctl.unregister_system_shared_memory(name)
                sysshm.destroy_shared_memory_region(h)
            report("200 x (create, register, infer, unregister)", r)
        for r in range(args.rounds):
            # sequences that are started and never ended (the server's idle-sequence bookkeeping), then model reloads
            for k in range(300):
                i0 = grpcclient.InferInput("INPUT", [1, 1], "INT32")
                i0.set_data_from_numpy(np.array([[k]], dtype=np.int32))
                ctl.infer("simple_sequence", [i0], sequence_id=100000 + r * 1000 + k, sequence_start=True)
            for m in models:
                ctl.unload_model(m)
                ctl.load_model(m)
            report("300 abandoned sequences + every model reloaded", r)
        if args.gpu is not None:
            # a load generator killed in mid-flight, four times over, then a clean run on the same regions
            import json
            import signal
            import subprocess
            import time

            import tritonclient.utils.cuda_shared_memory as hipshm

            loadgen = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                   "triton_client_amd", "_lib", "tca_loadgen")
            x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
            regions = []
            for i in range(64):
                hin = hipshm.create_shared_memory_region(f"kin{i}", x.nbytes, args.gpu)
                hout = hipshm.create_shared_memory_region(f"kout{i}", 32000, args.gpu)
                hipshm.set_shared_memory_region(hin, [x])
                ctl.register_cuda_shared_memory(f"kin{i}", hipshm.get_raw_handle(hin), args.gpu, x.nbytes)
                ctl.register_cuda_shared_memory(f"kout{i}", hipshm.get_raw_handle(hout), args.gpu, 32000)
                regions += [hin, hout]
            cmd = [loadgen, "-u", srv.grpc_url, "-i", "grpc", "-m", "resnet50", "-c", "64", "-b", "8", "--connections", "8",
                   "--shm-input", f"input:FP16:8,3,224,224:kin{{slot}}:{x.nbytes}", "--shm-output", "output:kout{slot}:32000"]
            for r in range(4):
                p = subprocess.Popen(cmd + ["-d", "30", "--warmup", "0.2"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                time.sleep(2.5)
                p.send_signal(signal.SIGKILL)
                p.wait()
                time.sleep(0.5)
                report("load generator killed with 64 requests in flight", r)
            out = subprocess.run(cmd + ["-d", "3", "--warmup", "0.5"], capture_output=True, text=True, timeout=120)
            res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
            print(f"clean run afterwards: {res['infer_per_sec']:.0f} infer/s, {res['errors']} errors", flush=True)
            ctl.unregister_cuda_shared_memory()
            for h in regions:
                hipshm.destroy_shared_memory_region(h)
            report("regions unregistered", 0)
        ctl.close()


if __name__ == "__main__":
    main()

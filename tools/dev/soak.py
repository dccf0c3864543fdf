#!/usr/bin/env python3
"""The headline load held for minutes instead of seconds: ResNet-50, batch-8 requests over gRPC with HIP shared
memory, 64 in flight, in consecutive windows; per window throughput, p50 / p99, errors, the server's resident set
size, its open file descriptors and the GPU memory in use.  A server that leaks, fragments its pools or slows
down over time shows here.

    python tools/dev/soak.py [windows=12] [seconds per window=10] [mode=hipshm|wire|system|ensemble]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from triton_client_amd.server import spawn_server  # noqa: E402

LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")


def gpu_used_mb():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
    try:
        card = next(iter(json.loads(out).values()))
        used = next(v for k, v in card.items() if "used" in k.lower())
        return round(int(used) / 2 ** 20)
    except Exception:  # noqa: BLE001 - the tool's output format is not ours
        return None


def ensemble(windows, seconds):
    """``preprocess_resnet50_ensemble``: eight 224x224 PNG files per request over gRPC from 24 client threads."""
    import io
    import threading
    import time

    import numpy as np
    import psutil
    import tritonclient.grpc as grpcclient
    from PIL import Image

    rng = np.random.default_rng(0)
    blobs = []
    for _ in range(8):
        buf = io.BytesIO()
        Image.fromarray((rng.random((224, 224, 3)) * 255).astype(np.uint8)).save(buf, "PNG", compress_level=3)
        blobs.append(buf.getvalue())
    with spawn_server(models=["resnet50", "preprocess_resnet50_ensemble"], gpu=0,
                      env={k: v for k, v in os.environ.items() if k.startswith("TCA_")}) as srv:
        server = psutil.Process(srv.proc.pid)
        print(f"ensemble (8 PNG files per request), 24 client threads, {windows} windows of {seconds:.0f} s", flush=True)
        print(f"{'window':>6s} {'img/s':>9s} {'errors':>6s} {'server RSS MB':>14s} {'fds':>5s} {'threads':>7s}", flush=True)
        stop = threading.Event()
        counts = [0] * 24
        errors = [0] * 24

        def worker(i):
            c = grpcclient.InferenceServerClient(srv.grpc_url)
            inp = grpcclient.InferInput("INPUT", [8, 1], "BYTES")
            inp.set_data_from_numpy(np.array(blobs, dtype=np.object_).reshape(8, 1))
            out = [grpcclient.InferRequestedOutput("OUTPUT", class_count=3)]
            while not stop.is_set():
                try:
                    c.infer("preprocess_resnet50_ensemble", [inp], outputs=out)
                    counts[i] += 1
                except Exception:  # noqa: BLE001 - counted
                    errors[i] += 1
            c.close()

        threads = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(24)]
        for t in threads:
            t.start()
        last = 0
        for w in range(windows):
            time.sleep(seconds)
            now = sum(counts)
            print(f"{w:6d} {8 * (now - last) / seconds:9.0f} {sum(errors):6d} {server.memory_info().rss / 2 ** 20:14.0f} "
                  f"{server.num_fds():5d} {server.num_threads():7d}", flush=True)
            last = now
        stop.set()
        for t in threads:
            t.join(timeout=30)


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "ensemble":
        return ensemble(int(sys.argv[1]), float(sys.argv[2]))
    import numpy as np
    import psutil
    import tritonclient.grpc as grpcclient
    import tritonclient.utils.cuda_shared_memory as hipshm

    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    mode = sys.argv[3] if len(sys.argv) > 3 else "hipshm"
    c = 64
    with spawn_server(models=["resnet50"], gpu=0, env={k: v for k, v in os.environ.items() if k.startswith("TCA_")}) as srv:
        server = psutil.Process(srv.proc.pid)
        ctl = grpcclient.InferenceServerClient(srv.grpc_url)
        x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
        regions = []
        cmd = [LOADGEN, "-u", srv.grpc_url, "-i", "grpc", "-m", "resnet50", "-c", str(c), "-b", "8", "--connections", "8",
               "-d", str(seconds), "--warmup", "0.5"]
        if mode == "hipshm":
            for i in range(c):
                hin = hipshm.create_shared_memory_region(f"sin{i}", x.nbytes, 0)
                hout = hipshm.create_shared_memory_region(f"sout{i}", 32000, 0)
                hipshm.set_shared_memory_region(hin, [x])
                ctl.register_cuda_shared_memory(f"sin{i}", hipshm.get_raw_handle(hin), 0, x.nbytes)
                ctl.register_cuda_shared_memory(f"sout{i}", hipshm.get_raw_handle(hout), 0, 32000)
                regions += [hin, hout]
            cmd += ["--shm-input", f"input:FP16:8,3,224,224:sin{{slot}}:{x.nbytes}", "--shm-output", "output:sout{slot}:32000"]
        elif mode == "system":
            import tritonclient.utils.shared_memory as sysshm

            for i in range(c):
                hin = sysshm.create_shared_memory_region(f"pin{i}", f"/tca_soak_in{i}", x.nbytes)
                hout = sysshm.create_shared_memory_region(f"pout{i}", f"/tca_soak_out{i}", 32000)
                sysshm.set_shared_memory_region(hin, [x])
                ctl.register_system_shared_memory(f"pin{i}", f"/tca_soak_in{i}", x.nbytes)
                ctl.register_system_shared_memory(f"pout{i}", f"/tca_soak_out{i}", 32000)
                regions += [hin, hout]
            cmd += ["--shm-input", f"input:FP16:8,3,224,224:pin{{slot}}:{x.nbytes}", "--shm-output", "output:pout{slot}:32000"]
        else:
            cmd += ["--shape", "input:FP16:8,3,224,224"]
        print(f"{mode}, {c} in flight, {windows} windows of {seconds:.0f} s", flush=True)
        print(f"{'window':>6s} {'infer/s':>9s} {'p50 us':>8s} {'p99 us':>8s} {'errors':>6s} {'server RSS MB':>14s} {'fds':>5s} "
              f"{'threads':>7s} {'GPU MB':>8s}", flush=True)
        for w in range(windows):
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=seconds + 120)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print("no result", out.stdout[-300:], out.stderr[-300:], flush=True)
                break
            r = json.loads(line[-1])
            print(f"{w:6d} {r['infer_per_sec']:9.0f} {r['p50_us']:8.0f} {r['p99_us']:8.0f} {r['errors']:6d} "
                  f"{server.memory_info().rss / 2 ** 20:14.0f} {server.num_fds():5d} {server.num_threads():7d} {gpu_used_mb()!s:>8s}",
                  flush=True)
        if regions and mode == "system":
            ctl.unregister_system_shared_memory()
            for r in regions:
                sysshm.destroy_shared_memory_region(r)
        elif regions:
            ctl.unregister_cuda_shared_memory()
            for r in regions:
                hipshm.destroy_shared_memory_region(r)
        ctl.close()


if __name__ == "__main__":
    main()

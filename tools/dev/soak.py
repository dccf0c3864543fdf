#!/usr/bin/env python3
"""The headline load held for minutes instead of seconds: ResNet-50, batch-8 requests over gRPC with HIP shared
memory, 64 in flight, in consecutive windows; per window throughput, p50 / p99, errors, the server's resident set
size, its open file descriptors and the GPU memory in use.  A server that leaks, fragments its pools or slows
down over time shows here.

    python tools/dev/soak.py [windows=12] [seconds per window=10] [mode=hipshm|wire|system]
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


def main():
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

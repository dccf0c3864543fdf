#!/usr/bin/env python3
"""The secondary serving configurations, one JSON line each:

  simple_http_sysshm      `simple` (add/sub, INT32[1,16]) over HTTP with system shared memory
  simple_grpc_wire        the same model over gRPC with tensors on the wire
  string_grpc_hipshm      `simple_string` (BYTES tensors) over gRPC with HIP shared memory
  decoupled_stream        `repeat_int32`: responses/sec of one bidirectional stream
"""
import argparse
import json
import os
import queue
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=int, default=None, help="GPU ordinal (needed for the hipshm configuration)")
    ap.add_argument("--window-ms", type=int, default=1000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np

    from triton_client_amd.perf import main as perf
    from triton_client_amd.server import spawn_server

    lines = []

    def emit(name, **kv):
        lines.append({"config": name, **kv})
        print(json.dumps(lines[-1]), flush=True)

    models = ["simple", "simple_string", "repeat_int32"]
    with spawn_server(models=models, gpu=args.gpu) as srv:
        common = ["-p", str(args.window_ms), "--stability-windows", "2", "--warmup-ms", "300"]
        rows = perf(["-m", "simple", "-u", srv.http_url, "-i", "http", "--shared-memory", "system", "--native",
                     "--concurrency-range", "16"] + common)
        emit("simple_http_sysshm", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
             p99_us=rows[0]["p99_us"])
        rows = perf(["-m", "simple", "-u", srv.grpc_url, "-i", "grpc", "--native", "--concurrency-range", "16"] + common)
        emit("simple_grpc_wire", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
             p99_us=rows[0]["p99_us"])
        shm = "hip" if args.gpu is not None else "system"
        rows = perf(["-m", "simple_string", "-u", srv.grpc_url, "-i", "grpc", "--shared-memory", shm, "--native",
                     "--concurrency-range", "16"] + common)
        emit(f"string_grpc_{shm}shm", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
             p99_us=rows[0]["p99_us"])

        import tritonclient.grpc as g

        got = queue.Queue()
        with g.InferenceServerClient(srv.grpc_url) as c:
            c.start_stream(callback=lambda result, error: got.put((result, error)))
            n_req, per_req = 200, 16
            values = np.arange(per_req, dtype=np.int32)
            inputs = [g.InferInput("IN", [per_req], "INT32"), g.InferInput("DELAY", [per_req], "UINT32"),
                      g.InferInput("WAIT", [1], "UINT32")]
            inputs[0].set_data_from_numpy(values)
            inputs[1].set_data_from_numpy(np.zeros(per_req, dtype=np.uint32))
            inputs[2].set_data_from_numpy(np.zeros(1, dtype=np.uint32))
            t0 = time.perf_counter()
            for i in range(n_req):
                c.async_stream_infer("repeat_int32", inputs, request_id=str(i))
            received = 0
            while received < n_req * per_req:
                result, error = got.get(timeout=60)
                if error is not None:
                    raise SystemExit(f"stream error: {error}")
                received += 1
            dt = time.perf_counter() - t0
            c.stop_stream()
        emit("decoupled_stream", requests=n_req, responses=received, responses_per_sec=round(received / dt, 1))
    if args.out:
        with open(args.out, "w") as fh:
            for line in lines:
                fh.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The secondary serving configurations, one JSON line each:

  simple_http_sysshm      `simple` (add/sub, INT32[1,16]) over HTTP with system shared memory
  simple_grpc_wire        the same model over gRPC with tensors on the wire
  string_grpc_hipshm      `simple_string` (BYTES tensors) over gRPC with HIP shared memory
  decoupled_stream        `repeat_int32` (16 responses per request) over bidirectional streams: responses/sec
  sequence_stream         `simple_sequence` (stateful: sequences of 20 requests, one correlation id each) over
                          bidirectional streams
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_configs(gpu=None, window_ms=1000, emit=None):
    """Runs the configurations against one freshly spawned server; returns a list of dicts."""
    from triton_client_amd.perf import main as perf
    from triton_client_amd.server import spawn_server

    lines = []

    def add(name, **kv):
        lines.append({"config": name, **kv})
        if emit is not None:
            emit(lines[-1])

    models = ["simple", "simple_string", "repeat_int32", "simple_sequence"]
    with spawn_server(models=models, gpu=gpu) as srv:
        common = ["-p", str(window_ms), "--stability-windows", "2", "--warmup-ms", "300"]
        rows = perf(["-m", "simple", "-u", srv.http_url, "-i", "http", "--shared-memory", "system", "--native",
                     "--concurrency-range", "16"] + common)
        add("simple_http_sysshm", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
            p99_us=rows[0]["p99_us"])
        rows = perf(["-m", "simple", "-u", srv.grpc_url, "-i", "grpc", "--native", "--concurrency-range",
                     "16"] + common)
        add("simple_grpc_wire", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
            p99_us=rows[0]["p99_us"])
        shm = "hip" if gpu is not None else "system"
        where = ["--devices", str(gpu)] if gpu is not None else []  # regions on the GPU the server runs on
        rows = perf(["-m", "simple_string", "-u", srv.grpc_url, "-i", "grpc", "--shared-memory", shm, "--native",
                     "--concurrency-range", "16"] + where + common)
        add(f"string_grpc_{shm}shm", concurrency=16, infer_per_sec=rows[0]["infer_per_sec"], p50_us=rows[0]["p50_us"],
            p99_us=rows[0]["p99_us"])
        rows = perf(["-m", "repeat_int32", "-u", srv.grpc_url, "-i", "grpc", "--native", "--streaming", "--shape",
                     "IN:16",
                     "--shape", "DELAY:16", "--shape", "WAIT:1", "--concurrency-range", "16"] + common)
        add("decoupled_stream", concurrency=16, responses_per_request=16, requests_per_sec=rows[0]["requests_per_sec"],
            responses_per_sec=rows[0]["responses_per_sec"], p50_us=rows[0]["p50_us"], p99_us=rows[0]["p99_us"])
        rows = perf(["-m", "simple_sequence", "-u", srv.grpc_url, "-i", "grpc", "--native", "--streaming",
                     "--sequence-length", "20", "--concurrency-range", "16"] + common)
        add("sequence_stream", concurrency=16, sequence_length=20, infer_per_sec=rows[0]["infer_per_sec"],
            p50_us=rows[0]["p50_us"], p99_us=rows[0]["p99_us"])
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=int, default=None, help="GPU ordinal (needed for the hipshm configuration)")
    ap.add_argument("--window-ms", type=int, default=1000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    lines = run_configs(args.gpu, args.window_ms, emit=lambda line: print(json.dumps(line), flush=True))
    if args.out:
        with open(args.out, "w") as fh:
            for line in lines:
                fh.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()

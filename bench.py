#!/usr/bin/env python3
"""Headline benchmark: perf_analyzer-style inferences/sec (+ p50/p99 latency)
of ResNet-50 fp16 batch 8 served by one PyTorch-ROCm instance per MI355X, with
inputs and outputs in HIP shared memory (hipshm) and requests over gRPC.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU through
``python -m torch.distributed.run``; every rank starts the v2 server for its
GPU (a child process: HIP IPC handles cannot be opened by their exporter),
drives it with ``--concurrency`` outstanding requests and the ranks meet on
RCCL barriers.  A *step* is ``--requests-per-step`` completed requests per GPU
(each request = one batch of 8 images).  Rank 0 prints one JSON line; ``value``
is the aggregate inferences/sec over all GPUs.

Data: synthetic images of the named shape, run through the HIP preprocessing
kernel into the hipshm input regions before timing; weights: random-init
ResNet-50.  Compute dtype fp16.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--concurrency", type=int, default=int(os.environ.get("TCA_BENCH_CONCURRENCY", "64")))
    ap.add_argument("--requests-per-step", type=int, default=256,
                    help="requests per GPU that make up one timed step")
    ap.add_argument("--protocol", choices=["grpc", "http"], default="grpc")
    ap.add_argument("--shared-memory", choices=["hip", "system", "none"], default="hip")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--extra", action="store_true", help="also measure the wire path and the fan-out path")
    ap.add_argument("--client", choices=["native", "python"], default=os.environ.get("TCA_BENCH_CLIENT", "native"),
                    help="native: C++ load generator (tca_loadgen) over the in-tree gRPC/HTTP client; "
                         "python: tritonclient-API threads in this process")
    ap.add_argument("--connections", type=int, default=4, help="gRPC channels the native load generator spreads over")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default=os.environ.get("TCA_BENCH_DIST_BACKEND", "nccl"),
                    help="nccl (= RCCL) for one rank per GPU; gloo only to rehearse several ranks on ONE GPU, "
                         "which RCCL refuses")
    return ap.parse_args()


LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")
if not os.path.exists(LOADGEN):
    LOADGEN = os.path.join(ROOT, "build", "bin", "tca_loadgen")


def measure_native(args, url, device, dist, label):
    """Closed loop driven by the C++ load generator.  This process only creates
    the HIP shared-memory regions (filled by the HIP preprocessing kernel),
    registers them and brackets the measured phase: the generator warms up,
    prints READY, waits for the go line, then issues exactly steps x
    requests-per-step requests with ``--concurrency`` outstanding."""
    import subprocess

    import numpy as np
    import torch

    from triton_client_amd.ops.preprocess import Preprocessor

    if args.shared_memory != "hip":
        raise SystemExit("--client native drives the hip shared-memory configuration; use --client python otherwise")
    if not os.path.exists(LOADGEN):
        raise SystemExit(f"{LOADGEN} is missing: run `python -m triton_client_amd._build cpp`")
    import tritonclient.grpc as grpcclient
    import tritonclient.utils.cuda_shared_memory as hipshm

    b = args.batch
    in_bytes, out_bytes = b * 3 * 224 * 224 * 2, b * 1000 * 4
    rng = np.random.default_rng(1234 + device)
    images = [rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8) for _ in range(b)]
    pre = Preprocessor(224, 224, 3, "NCHW", "FP16", scaling="INCEPTION", device=device)
    control = grpcclient.InferenceServerClient(url if args.protocol == "grpc" else args.control_url)
    regions = []
    per_step = args.requests_per_step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    proc = None
    try:
        for i in range(args.concurrency):
            hin = hipshm.create_shared_memory_region(f"in_{device}_{i}", in_bytes, device)
            hout = hipshm.create_shared_memory_region(f"out_{device}_{i}", out_bytes, device)
            regions += [hin, hout]
            pre(images, hin)  # HIP kernel writes the batch straight into the region
            control.register_cuda_shared_memory(f"in_{device}_{i}", hipshm.get_raw_handle(hin), device, in_bytes)
            control.register_cuda_shared_memory(f"out_{device}_{i}", hipshm.get_raw_handle(hout), device, out_bytes)
        cmd = [LOADGEN, "-u", url, "-i", args.protocol, "-m", args.model, "-c", str(args.concurrency), "-b", str(b),
               "--connections", str(args.connections), "--count", str(args.steps * per_step), "--warmup-count",
               str(max(1, args.warmup) * per_step), "--wait-go",
               "--shm-input", f"input:FP16:{b},3,224,224:in_{device}_{{slot}}:{in_bytes}",
               "--shm-output", f"output:out_{device}_{{slot}}:{out_bytes}"]
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        ready = proc.stdout.readline().strip()
        if ready != "READY":
            raise RuntimeError(f"load generator did not get ready: {ready!r}")
        barrier()
        t0 = time.perf_counter()
        proc.stdin.write("GO\n")
        proc.stdin.flush()
        result = json.loads(proc.stdout.readline())
        barrier()
        elapsed = time.perf_counter() - t0
        proc.wait(timeout=30)
        if result["errors"] or result["requests"] != args.steps * per_step:
            raise RuntimeError(f"load generator reported {result['errors']} errors "
                               f"({result['requests']} of {args.steps * per_step} requests): {result['first_error']}")
    finally:
        if proc is not None and proc.poll() is None:
            proc.kill()
        for i in range(len(regions) // 2):
            try:
                control.unregister_cuda_shared_memory(f"in_{device}_{i}")
                control.unregister_cuda_shared_memory(f"out_{device}_{i}")
            except Exception:
                pass
        for r in regions:
            hipshm.destroy_shared_memory_region(r)
        control.close()
        pre.close()
    pcts = [result["p50_us"], result["p90_us"], result["p95_us"], result["p99_us"]]
    if dist is not None:
        t = torch.tensor([elapsed] + pcts, device=args.reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # slowest rank bounds the job; percentiles: worst rank's
        elapsed, pcts = float(t[0].item()), [float(v) for v in t[1:].tolist()]
    world = dist.get_world_size() if dist is not None else 1
    return {"label": label, "infer_per_sec": world * args.steps * per_step * b / elapsed, "elapsed_s": elapsed,
            "p50_us": pcts[0], "p90_us": pcts[1], "p95_us": pcts[2], "p99_us": pcts[3],
            "generator_infer_per_sec": result["infer_per_sec"]}


class Worker:
    """One outstanding-request slot: own client, own pair of shm regions."""

    def __init__(self, idx, args, url, device, images_u8, pre):
        import numpy as np

        self.idx = idx
        self.args = args
        self.device = device
        b = args.batch
        self.in_bytes = b * 3 * 224 * 224 * 2
        self.out_bytes = b * 1000 * 4
        if args.protocol == "grpc":
            import tritonclient.grpc as mod
            self.client = mod.InferenceServerClient(url)
        else:
            import tritonclient.http as mod
            self.client = mod.InferenceServerClient(url)
        self.mod = mod
        self.regions = []
        inp = mod.InferInput("input", [b, 3, 224, 224], "FP16")
        out = mod.InferRequestedOutput("output")
        mode = args.shared_memory
        if mode == "hip":
            import tritonclient.utils.cuda_shared_memory as hipshm

            self.hin = hipshm.create_shared_memory_region(f"in_{device}_{idx}", self.in_bytes, device)
            self.hout = hipshm.create_shared_memory_region(f"out_{device}_{idx}", self.out_bytes, device)
            pre(images_u8, self.hin)  # HIP kernel writes the batch straight into the region
            self.client.register_cuda_shared_memory(f"in_{device}_{idx}", hipshm.get_raw_handle(self.hin), device,
                                                    self.in_bytes)
            self.client.register_cuda_shared_memory(f"out_{device}_{idx}", hipshm.get_raw_handle(self.hout), device,
                                                    self.out_bytes)
            inp.set_shared_memory(f"in_{device}_{idx}", self.in_bytes)
            out.set_shared_memory(f"out_{device}_{idx}", self.out_bytes)
            self._shm = hipshm
        else:
            import torch

            x = torch.empty(b, 3, 224, 224, device=f"cuda:{device}", dtype=torch.float16)
            pre(images_u8, x)
            host = x.cpu().numpy()
            if mode == "system":
                import tritonclient.utils.shared_memory as sysshm

                self.hin = sysshm.create_shared_memory_region(f"in_{device}_{idx}", f"/tca_b_in_{os.getpid()}_{idx}",
                                                              self.in_bytes)
                self.hout = sysshm.create_shared_memory_region(f"out_{device}_{idx}",
                                                               f"/tca_b_out_{os.getpid()}_{idx}", self.out_bytes)
                sysshm.set_shared_memory_region(self.hin, [host])
                self.client.register_system_shared_memory(f"in_{device}_{idx}", f"/tca_b_in_{os.getpid()}_{idx}",
                                                          self.in_bytes)
                self.client.register_system_shared_memory(f"out_{device}_{idx}", f"/tca_b_out_{os.getpid()}_{idx}",
                                                          self.out_bytes)
                inp.set_shared_memory(f"in_{device}_{idx}", self.in_bytes)
                out.set_shared_memory(f"out_{device}_{idx}", self.out_bytes)
                self._shm = sysshm
            else:
                inp.set_data_from_numpy(host)
                self._shm = None
        self.inputs = [inp]
        self.outputs = [out]
        self.latencies = []

    def one(self):
        t0 = time.perf_counter_ns()
        self.client.infer(self.args.model, self.inputs, outputs=self.outputs)
        self.latencies.append(time.perf_counter_ns() - t0)

    def close(self):
        try:
            if self.args.shared_memory == "hip":
                self.client.unregister_cuda_shared_memory(f"in_{self.device}_{self.idx}")
                self.client.unregister_cuda_shared_memory(f"out_{self.device}_{self.idx}")
            elif self.args.shared_memory == "system":
                self.client.unregister_system_shared_memory(f"in_{self.device}_{self.idx}")
                self.client.unregister_system_shared_memory(f"out_{self.device}_{self.idx}")
            if self._shm is not None:
                self._shm.destroy_shared_memory_region(self.hin)
                self._shm.destroy_shared_memory_region(self.hout)
        finally:
            self.client.close()


def run_requests(workers, total):
    """Closed loop: every worker keeps one request outstanding until ``total``
    requests have been issued."""
    lock = threading.Lock()
    remaining = [total]
    errors = []

    def loop(w):
        while True:
            with lock:
                if remaining[0] <= 0:
                    return
                remaining[0] -= 1
            try:
                w.one()
            except Exception as e:  # surface the first failure, stop everybody
                with lock:
                    errors.append(e)
                    remaining[0] = 0
                return

    threads = [threading.Thread(target=loop, args=(w,)) for w in workers]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def measure(args, url, device, dist, label):
    import numpy as np
    import torch

    from triton_client_amd.ops.preprocess import Preprocessor

    rng = np.random.default_rng(1234 + device)
    images = [rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8) for _ in range(args.batch)]
    pre = Preprocessor(224, 224, 3, "NCHW", "FP16", scaling="INCEPTION", device=device)
    workers = [Worker(i, args, url, device, images, pre) for i in range(args.concurrency)]
    pre.close()
    per_step = args.requests_per_step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        run_requests(workers, max(1, args.warmup) * per_step)
        for w in workers:
            w.latencies.clear()
        barrier()
        t0 = time.perf_counter()
        run_requests(workers, args.steps * per_step)
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        lat = np.concatenate([np.asarray(w.latencies, dtype=np.float64) for w in workers]) if workers else np.zeros(1)
        for w in workers:
            w.close()
    if dist is not None:
        t = torch.tensor([elapsed], device=args.reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # latency samples of every rank -> rank 0 (fixed count per rank, gathered over RCCL)
        mine = torch.zeros(args.steps * per_step, device=args.reduce_device, dtype=torch.float64)
        mine[: len(lat)] = torch.from_numpy(lat[: len(mine)]).to(mine.device)
        gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
        lat = torch.cat(gathered).cpu().numpy()
        lat = lat[lat > 0]
    world = dist.get_world_size() if dist is not None else 1
    infer = world * args.steps * per_step * args.batch
    return {
        "label": label,
        "infer_per_sec": infer / elapsed,
        "elapsed_s": elapsed,
        "p50_us": float(np.percentile(lat, 50)) / 1e3,
        "p90_us": float(np.percentile(lat, 90)) / 1e3,
        "p95_us": float(np.percentile(lat, 95)) / 1e3,
        "p99_us": float(np.percentile(lat, 99)) / 1e3,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs at least one MI355X")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod
    args.reduce_device = f"cuda:{device}" if args.dist_backend == "nccl" else "cpu"

    from triton_client_amd.server import spawn_server

    server = spawn_server(models=[args.model], gpu=device,
                          extra_args=["--grpc-workers", str(max(32, 2 * args.concurrency)), "--http-workers",
                                      str(max(16, 2 * args.concurrency))])
    try:
        url = server.grpc_url if args.protocol == "grpc" else server.http_url
        args.control_url = server.grpc_url
        run = measure_native if args.client == "native" else measure
        main_result = run(args, url, device, dist, f"{args.protocol}+{args.shared_memory}shm")
        extras = {}
        if args.extra:
            for mode in ("none", "system"):
                alt = argparse.Namespace(**vars(args))
                alt.shared_memory = mode
                alt.steps = max(2, args.steps // 4)
                alt.warmup = 1
                extras[mode] = measure(alt, url, device, dist, f"{args.protocol}+{mode}")
    finally:
        server.stop()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        n = world
        ms_per_step = main_result["elapsed_s"] * 1e3 / args.steps
        line = {
            "metric": "ResNet-50 fp16 batch-8 inferences/sec (perf_analyzer-style, gRPC, HIP shared memory I/O)",
            "value": round(main_result["infer_per_sec"], 2),
            "unit": "infer/sec",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp16",
            "data": "synthetic images (375x500 u8) through the HIP preprocessing kernel; random-init ResNet-50 weights",
            "p50_latency_us": round(main_result["p50_us"], 1),
            "p90_latency_us": round(main_result["p90_us"], 1),
            "p95_latency_us": round(main_result["p95_us"], 1),
            "p99_latency_us": round(main_result["p99_us"], 1),
            "config": {
                "model": "resnet50",
                "batch": args.batch,
                "global_batch": args.batch * args.concurrency * n,
                "seq_len": None,
                "concurrency_per_gpu": args.concurrency,
                "requests_per_step_per_gpu": args.requests_per_step,
                "protocol": args.protocol,
                "shared_memory": args.shared_memory,
                "load_generator": "tca_loadgen (C++, in-tree gRPC client)" if args.client == "native"
                else "tritonclient-API threads (Python)",
                "parallelism": f"dp{n} (one model instance per GPU, requests sharded)",
            },
        }
        if extras:
            line["extra"] = {k: {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                             for k, v in extras.items()}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

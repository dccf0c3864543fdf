#!/usr/bin/env python3
"""Headline benchmark: perf_analyzer-style inferences/sec (+ p50/p99 latency)
of ResNet-50 fp16 batch 8 served by one PyTorch-ROCm instance per MI355X, with
inputs and outputs in HIP shared memory (hipshm) and requests over gRPC.

    python bench.py --gpus N --steps K --warmup W

One rank per GPU.  Under ``python -m torch.distributed.run`` (RANK / WORLD_SIZE
in the environment) this process IS a rank; started plainly with ``--gpus N``
(N > 1) it starts the N ranks itself as children (before touching the GPU),
relays rank 0's JSON line and fails if any rank fails.  Every rank starts the
v2 server for its GPU (a child process: HIP IPC handles cannot be opened by
their exporter), drives it with ``--concurrency`` outstanding requests and the
ranks meet on RCCL barriers.  A *step* is ``--requests-per-step`` completed
requests per GPU (each request = one batch of 8 images).  Rank 0 prints one
JSON line; ``value`` is the aggregate inferences/sec over all GPUs.  The line
also carries short measurements of the same requests with the tensors on the
wire and in system shared memory (``extra.wire`` / ``extra.system_shm``), so the
shared-memory-versus-wire comparison comes from the same run.

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
    ap.add_argument("--requests-per-step", type=int, default=1536,
                    help="requests per GPU that make up one timed step")
    ap.add_argument("--protocol", choices=["grpc", "http"], default="grpc")
    ap.add_argument("--shared-memory", choices=["hip", "system", "none"], default="hip")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short wire / system-shared-memory comparison runs")
    ap.add_argument("--extra-requests", type=int, default=12288,
                    help="requests per GPU of each comparison run (wire, system shm)")
    ap.add_argument("--client", choices=["native", "python"], default=os.environ.get("TCA_BENCH_CLIENT", "native"),
                    help="native: C++ load generator (tca_loadgen) over the in-tree gRPC/HTTP client; "
                         "python: tritonclient-API threads in this process")
    ap.add_argument("--connections", type=int, default=8, help="gRPC channels the native load generator spreads over")
    ap.add_argument("--image-requests", type=int, default=16384,
                    help="requests of the end-to-end image pipeline run (extra.image_client)")
    ap.add_argument("--ensemble-requests", type=int, default=3072,
                    help="requests (8 PNG files each) of the server-side preprocessing run (extra.ensemble)")
    ap.add_argument("--image-lanes", type=int, default=3)
    ap.add_argument("--image-depth", type=int, default=6)
    ap.add_argument("--sweep", default="1,4,16,64,256",
                    help="concurrency levels of the HIP shared-memory sweep reported under extra.sweep ('' = none)")
    ap.add_argument("--sweep-seconds", type=float, default=1.5, help="measured window of every sweep level")
    ap.add_argument("--secondary-window-ms", type=int, default=2000)
    ap.add_argument("--generator-timeout", type=float, default=600.0,
                    help="seconds the load generator may take for one phase before the run is abandoned")
    ap.add_argument("--mode", choices=["serve", "fanout", "image_client"], default="serve",
                    help="fanout: also measure resnet50_sharded (one request's batch split over the GPUs), "
                         "reported under extra.fanout; opt-in, with several ranks too")
    ap.add_argument("--fanout-devices", default=None,
                    help="GPU ordinals of the fan-out replicas besides GPU 0 (default: every other rank's GPU; "
                         "'0,0' rehearses three replicas on one GPU)")
    ap.add_argument("--fanout-modes", default="direct,rccl")
    ap.add_argument("--fanout-requests", type=int, default=384)
    ap.add_argument("--fanout-timeout", type=float, default=180.0,
                    help="seconds the fan-out measurement of ONE transport may take before it is abandoned")
    ap.add_argument("--fanout-child", type=int, default=0, help=argparse.SUPPRESS)  # internal: world size
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default=os.environ.get("TCA_BENCH_DIST_BACKEND",
                                                                                       "nccl"),
                    help="nccl (= RCCL) for one rank per GPU; gloo only to rehearse several ranks on ONE GPU, "
                         "which RCCL refuses")
    return ap.parse_args()


LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")
if not os.path.exists(LOADGEN):
    LOADGEN = os.path.join(ROOT, "build", "bin", "tca_loadgen")


def structured_images(count, first=0, height=375, width=500):
    """Synthetic u8 images with structure - colour gradients in a direction, a checkerboard and a filled
    rectangle, all derived from the image's index - so that every image is different from every other one in
    ways that survive resizing: the logits of a random-init network then differ between images by far more
    than fp16 rounding, and a path that mixed up or ignored its inputs would show."""
    import numpy as np

    # built at quarter resolution and repeated 4 x 4: a millisecond per image, and features coarse enough to
    # survive the resize to 224 x 224
    qh, qw = (height + 3) // 4, (width + 3) // 4
    yy, xx = np.mgrid[0:qh, 0:qw].astype(np.float32)
    out = []
    for k in range(first, first + count):
        c0 = np.array([(37 * k + 11) % 256, (91 * k + 60) % 256, (53 * k + 120) % 256], dtype=np.float32)
        c1 = np.array([(101 * k + 200) % 256, (17 * k + 5) % 256, (149 * k + 77) % 256], dtype=np.float32)
        t = (xx / qw, yy / qh, (xx + yy) / (qw + qh), (xx / qw - yy / qh + 1.0) / 2.0)[k % 4]
        img = c0[None, None, :] * (1.0 - t[..., None]) + c1[None, None, :] * t[..., None]
        block = 3 + 2 * (k % 7)
        checker = (((xx // block) + (yy // block)) % 2) * 2.0 - 1.0
        img += 35.0 * checker[..., None] * np.array([1.0, -1.0, 0.5], dtype=np.float32)[(k // 3) % 3]
        y0, x0 = (7 * k) % (qh // 2), (11 * k) % (qw // 2)
        img[y0:y0 + qh // 3, x0:x0 + qw // 3, k % 3] = (200 + 13 * k) % 256
        small = np.clip(img, 0, 255).astype(np.uint8)
        out.append(np.ascontiguousarray(small.repeat(4, axis=0).repeat(4, axis=1)[:height, :width]))
    return out


def read_line(proc, timeout, what):
    """One line of the load generator's stdout, or an error once ``timeout`` seconds have passed without one
    (a generator that stalls must not hold the rank until the driver's own limit)."""
    import select

    ready, _, _ = select.select([proc.stdout], [], [], timeout)
    if not ready:
        proc.kill()
        raise RuntimeError(f"load generator: no {what} within {timeout:.0f} s")
    return proc.stdout.readline()


def check_contents(args, device, slots, hipshm, batch=None, model_name=None):
    """Reads the input and output regions of a few slots back and compares the served logits with a forward of
    the same inputs by a model instance loaded in THIS process (same seeded random-init weights).  The server
    merges requests into batches of other sizes (or splits them over replicas), which run other tile variants,
    so the comparison is to fp16 rounding, not bit for bit; the images differ (structured_images), so do the
    logits, and that is asserted."""
    import numpy as np
    import torch

    from triton_client_amd.models import MODEL_ZOO

    b = batch or args.batch
    chunk = min(b, 64)
    model = MODEL_ZOO[model_name or args.model]()
    model.load(device)
    try:
        served, direct = [], []
        for hin, hout in slots:
            x = hipshm.get_contents_as_numpy(hin, np.float16, [b, 3, 224, 224])
            served.append(np.array(hipshm.get_contents_as_numpy(hout, np.float32, [b, 1000]), copy=True))
            xt = torch.from_numpy(np.array(x, copy=True)).to(f"cuda:{device}")
            for s in range(0, b, chunk):
                out = model.run(xt[s: s + chunk])
                model.stream.synchronize()
                direct.append(out.float().cpu().numpy())
        served, direct = np.concatenate(served), np.concatenate(direct)
    finally:
        model.unload()
    err = float(np.abs(served - direct).max())
    scale = float(np.abs(direct).max())
    # how far apart the logits of two different images are at least (every pair of up to 64 checked images,
    # taken evenly from all of them)
    n = served.shape[0]
    some = served[np.linspace(0, n - 1, min(n, 64)).astype(int)]
    apart = min(float(np.abs(some[i] - some[j]).max()) for i in range(len(some)) for j in range(i + 1, len(some)))
    ok = bool(np.isfinite(served).all() and err <= 0.05 * scale and apart > 4.0 * err
              and (served.argmax(1) == direct.argmax(1)).mean() >= 0.9)
    return {"max_abs_err": round(err, 5), "logit_scale": round(scale, 4), "images_checked": int(n),
            "min_distance_between_images": round(apart, 4),
            "top1_agreement": round(float((served.argmax(1) == direct.argmax(1)).mean()), 4),
            "reference": f"model.run of the same input regions in the benchmark process (batch {chunk})", "ok": ok}


def measure_native(args, url, device, dist, label, mode=None, steps=None, per_step=None, warmup_requests=None,
                   model=None, batch=None, concurrency=None, check=False, sweep=None):
    """Closed loop driven by the C++ load generator.  This process only prepares
    the tensors (HIP preprocessing kernel -> HIP shared-memory regions, or system
    shared memory, or nothing for tensors on the wire), registers the regions and
    brackets the measured phase: the generator warms up, prints READY, waits for
    the go line, then issues exactly steps x requests-per-step requests with
    ``--concurrency`` outstanding."""
    import subprocess

    import numpy as np
    import torch

    from triton_client_amd.ops.preprocess import Preprocessor

    mode = mode or args.shared_memory
    steps = steps or args.steps
    per_step = per_step or args.requests_per_step
    if warmup_requests is None:
        warmup_requests = max(1, args.warmup) * per_step
    if not os.path.exists(LOADGEN):
        raise SystemExit(f"{LOADGEN} is missing: run `python -m triton_client_amd._build cpp`")
    import tritonclient.grpc as grpcclient

    b = batch or args.batch
    model = model or args.model
    concurrency = concurrency or args.concurrency
    in_bytes, out_bytes = b * 3 * 224 * 224 * 2, b * 1000 * 4
    images = structured_images(b, first=1000 * device)
    pre = Preprocessor(224, 224, 3, "NCHW", "FP16", scaling="INCEPTION", device=device, max_batch=max(8, min(b, 64)))
    extra = {}
    image_sets = {0: images}
    levels = [int(v) for v in (sweep or [])]
    slots_wanted = max([concurrency] + levels)
    control = grpcclient.InferenceServerClient(url if args.protocol == "grpc" else args.control_url)
    regions = []
    cleanup = []
    big = {}

    def big_batch(key):
        # large batches (fan-out runs): the preprocessing kernel in chunks of 64 images, eight different sets
        if key not in big:
            x = torch.empty(b, 3, 224, 224, device=f"cuda:{device}", dtype=torch.float16)
            imgs = structured_images(b, first=1000 * device + key * b)
            for s in range(0, b, 64):
                pre(imgs[s: s + 64], x[s: s + 64])
            torch.cuda.synchronize()
            big[key] = x
        return big[key]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    proc = None
    try:
        cmd = [LOADGEN, "-u", url, "-i", args.protocol, "-m", model, "-c", str(concurrency), "-b", str(b),
               "--connections", str(args.connections), "--count", str(steps * per_step), "--warmup-count",
               str(warmup_requests), "--wait-go"]
        if mode == "hip":
            import tritonclient.utils.cuda_shared_memory as hipshm

            for i in range(slots_wanted):
                hin = hipshm.create_shared_memory_region(f"in_{device}_{i}", in_bytes, device)
                hout = hipshm.create_shared_memory_region(f"out_{device}_{i}", out_bytes, device)
                regions += [hin, hout]
                if b > 64:
                    hipshm.set_shared_memory_region_from_dlpack(hin, [big_batch(i % 8)])
                else:
                    # every slot its own images; the HIP kernel writes the batch straight into the region
                    # (the sweep's extra slots repeat the image sets of the first 64)
                    key = i % 64
                    if key not in image_sets:
                        image_sets[key] = structured_images(b, first=1000 * device + key * b)
                    pre(image_sets[key], hin)
                control.register_cuda_shared_memory(f"in_{device}_{i}", hipshm.get_raw_handle(hin), device, in_bytes)
                control.register_cuda_shared_memory(f"out_{device}_{i}", hipshm.get_raw_handle(hout), device, out_bytes)
                cleanup.append(lambda i=i: (control.unregister_cuda_shared_memory(f"in_{device}_{i}"),
                                            control.unregister_cuda_shared_memory(f"out_{device}_{i}")))
            destroy = hipshm.destroy_shared_memory_region
        elif mode == "system":
            import tritonclient.utils.shared_memory as sysshm

            x = torch.empty(b, 3, 224, 224, device=f"cuda:{device}", dtype=torch.float16)
            pre(images, x)
            host = x.cpu().numpy()
            for i in range(concurrency):
                kin, kout = f"/tca_b_in_{os.getpid()}_{i}", f"/tca_b_out_{os.getpid()}_{i}"
                hin = sysshm.create_shared_memory_region(f"in_{device}_{i}", kin, in_bytes)
                hout = sysshm.create_shared_memory_region(f"out_{device}_{i}", kout, out_bytes)
                regions += [hin, hout]
                sysshm.set_shared_memory_region(hin, [host])
                control.register_system_shared_memory(f"in_{device}_{i}", kin, in_bytes)
                control.register_system_shared_memory(f"out_{device}_{i}", kout, out_bytes)
                cleanup.append(lambda i=i: (control.unregister_system_shared_memory(f"in_{device}_{i}"),
                                            control.unregister_system_shared_memory(f"out_{device}_{i}")))
            destroy = sysshm.destroy_shared_memory_region
        else:
            destroy = None
        if mode in ("hip", "system"):
            cmd += ["--shm-input", f"input:FP16:{b},3,224,224:in_{device}_{{slot}}:{in_bytes}",
                    "--shm-output", f"output:out_{device}_{{slot}}:{out_bytes}"]
        else:
            cmd += ["--shape", f"input:FP16:{b},3,224,224"]  # 1.2 MB of tensor bytes in every request
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        ready = read_line(proc, args.generator_timeout, "READY line (warm-up)").strip()
        if ready != "READY":
            raise RuntimeError(f"load generator did not get ready: {ready!r}")
        barrier()
        t0 = time.perf_counter()
        proc.stdin.write("GO\n")
        proc.stdin.flush()
        result = json.loads(read_line(proc, args.generator_timeout, "result of the measured phase"))
        barrier()
        elapsed = time.perf_counter() - t0
        proc.wait(timeout=30)
        if result["errors"] or result["requests"] != steps * per_step:
            raise RuntimeError(f"load generator reported {result['errors']} errors "
                               f"({result['requests']} of {steps * per_step} requests): {result['first_error']}")
        if check and mode == "hip":
            try:
                # slots far apart: their requests were merged into different batches
                picks = sorted({0, concurrency // 3, (2 * concurrency) // 3, concurrency - 1})
                extra["check"] = check_contents(args, device, [(regions[2 * i], regions[2 * i + 1]) for i in picks],
                                                hipshm, batch=b, model_name=check if isinstance(check, str) else None)
            except Exception as e:  # noqa: BLE001 - reported in the line; the throughput result stands
                extra["check"] = {"error": f"{type(e).__name__}: {e}"[:400], "ok": False}
        if levels and mode == "hip":
            extra["sweep"] = run_sweep(args, cmd, levels, barrier, dist, b)
    finally:
        if proc is not None and proc.poll() is None:
            proc.kill()
        for fn in cleanup:
            try:
                fn()
            except Exception:
                pass
        for r in regions:
            destroy(r)
        control.close()
        pre.close()
    pcts = [result["p50_us"], result["p90_us"], result["p95_us"], result["p99_us"]]
    mine = steps * per_step * b / elapsed
    per_gpu = [mine]
    if dist is not None:
        t = torch.tensor([elapsed] + pcts, device=args.reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # slowest rank bounds the job; percentiles: worst rank's
        elapsed, pcts = float(t[0].item()), [float(v) for v in t[1:].tolist()]
        each = torch.zeros(dist.get_world_size(), device=args.reduce_device, dtype=torch.float64)
        each[dist.get_rank()] = mine
        dist.all_reduce(each)
        per_gpu = [float(v) for v in each.tolist()]
    world = dist.get_world_size() if dist is not None else 1
    return {"label": label, "infer_per_sec": world * steps * per_step * b / elapsed, "elapsed_s": elapsed,
            "p50_us": pcts[0], "p90_us": pcts[1], "p95_us": pcts[2], "p99_us": pcts[3],
            "generator_infer_per_sec": result["infer_per_sec"], "per_gpu_infer_per_sec": per_gpu,
            "requests": world * steps * per_step, **extra}


def run_sweep(args, base_cmd, levels, barrier, dist, b):
    """Concurrency sweep over the slots' HIP shared-memory regions, the load generator in its timed mode: every
    level warms up, then records ``--sweep-seconds``.  With several ranks every rank sweeps its own server at the
    same time: throughputs are summed, latencies are the worst rank's."""
    import subprocess

    import torch

    def replaced(cmd, flag, value):
        cmd = list(cmd)
        cmd[cmd.index(flag) + 1] = str(value)
        return cmd

    base = [a for a in base_cmd if a != "--wait-go"]
    for flag in ("--count", "--warmup-count"):  # timed mode: no request counts
        i = base.index(flag)
        del base[i:i + 2]
    out = []
    for c in levels:
        cmd = replaced(base, "-c", c) + ["-d", str(args.sweep_seconds), "--warmup", "0.4"]
        barrier()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        try:
            r = json.loads(read_line(proc, args.sweep_seconds + 60.0, f"result of sweep level {c}"))
            proc.wait(timeout=30)
        finally:
            if proc.poll() is None:
                proc.kill()
        if r["errors"]:
            raise RuntimeError(f"sweep level {c}: {r['errors']} errors: {r['first_error']}")
        vals = [r["infer_per_sec"], r["p50_us"], r["p99_us"], r["duration_s"]]
        if dist is not None:
            t = torch.tensor(vals[:1], device=args.reduce_device, dtype=torch.float64)
            dist.all_reduce(t)
            m = torch.tensor(vals[1:], device=args.reduce_device, dtype=torch.float64)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            vals = [float(t[0].item())] + [float(v) for v in m.tolist()]
        out.append({"concurrency_per_gpu": c, "infer_per_sec": round(vals[0], 1), "p50_latency_us": round(vals[1], 1),
                    "p99_latency_us": round(vals[2], 1), "window_s": round(vals[3], 3), "request_batch": b})
    return out


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


def measure_image_client(args, url, device):
    """The image client's whole loop, pipelined: decoded u8 images (480x640) in page-locked
    memory -> host-to-device copy + fused HIP preprocess -> slot of a HIP shared-memory region
    -> async gRPC request for the top-3 classes -> response.  Image decode is outside the
    timed part; everything else a classification client does per image is inside."""
    import numpy as np

    from triton_client_amd.pipeline import ImagePipeline

    batches = [structured_images(args.batch, first=5000 + 1000 * device + k * args.batch, height=480, width=640)
               for k in range(4)]
    pipe = ImagePipeline(url, args.model, batch=args.batch, lanes=args.image_lanes, depth=args.image_depth,
                         classes=3, device=device)
    try:
        pipe.pin(batches)
        pipe.run(max(64, args.image_requests // 8))  # warm-up: graphs for the slots' addresses, channels
        r = pipe.run(args.image_requests, keep_results=1)
        top = [e.decode() for e in r["results"][0].as_numpy(pipe.output_name).reshape(-1)[:3]]
        return {"images_per_sec": round(r["images_per_sec"], 1), "requests": r["requests"], "batch": args.batch,
                "elapsed_s": round(r["elapsed_s"], 4), "p50_latency_us": round(r["p50_us"], 1),
                "p99_latency_us": round(r["p99_us"], 1), "lanes": args.image_lanes, "slots_per_lane": args.image_depth,
                "source": "480x640x3 u8 in pinned host memory", "client": "Python (tritonclient.grpc async_infer)",
                "first_result_top3": top}
    finally:
        pipe.close()


def measure_ensemble(args, server, device):
    """``preprocess_resnet50_ensemble``: PNG files (224x224, eight per request) inside gRPC requests, top-3
    classes out, 32 requests in flight.  The server decodes on a thread pool, preprocesses with the HIP
    kernel and runs the classifier as a step through resnet50's dynamic batcher.  The first answer is
    compared with the same images through client-side preprocessing + ``resnet50``."""
    import io
    import threading

    import numpy as np
    import tritonclient.grpc as grpcclient
    from PIL import Image

    from triton_client_amd.ops.preprocess import Preprocessor

    b, inflight, total = 8, 32, args.ensemble_requests
    sets = []
    raw_first = None
    for k in range(8):
        imgs = structured_images(b, first=9000 + 1000 * device + k * b, height=224, width=224)
        if raw_first is None:
            raw_first = imgs
        blobs = []
        for img in imgs:
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, "PNG", compress_level=3)
            blobs.append(buf.getvalue())
        inp = grpcclient.InferInput("INPUT", [b, 1], "BYTES")
        inp.set_data_from_numpy(np.array(blobs, dtype=np.object_).reshape(b, 1))
        sets.append([inp])
    outputs = [grpcclient.InferRequestedOutput("OUTPUT", class_count=3)]
    control = grpcclient.InferenceServerClient(server.grpc_url)
    clients = [grpcclient.InferenceServerClient(server.grpc_url) for _ in range(4)]
    try:
        lock = threading.Lock()
        done = threading.Event()
        state = {"issued": 0, "finished": 0, "errors": 0, "first": None, "lat": [], "goal": 0}

        def issue(slot):
            with lock:
                if state["issued"] >= state["goal"]:
                    return
                n = state["issued"]
                state["issued"] += 1
            t0 = time.perf_counter()

            def on_done(result, error, t0=t0, slot=slot, n=n):
                with lock:
                    state["finished"] += 1
                    if error is not None:
                        state["errors"] += 1
                        state["first_error"] = str(error)
                    else:
                        state["lat"].append(time.perf_counter() - t0)
                        if n == 0:
                            state["first"] = result
                    last = state["finished"] >= state["goal"]
                if last:
                    done.set()
                else:
                    issue(slot)

            clients[slot % len(clients)].async_infer("preprocess_resnet50_ensemble", sets[n % len(sets)], on_done,
                                                     outputs=outputs)

        def run(goal):
            with lock:
                state.update(issued=0, finished=0, errors=0, lat=[], goal=goal)
            done.clear()
            t0 = time.perf_counter()
            for slot in range(min(inflight, goal)):
                issue(slot)
            if not done.wait(timeout=args.generator_timeout):
                raise RuntimeError("ensemble run did not finish")
            return time.perf_counter() - t0

        run(max(64, total // 8))
        elapsed = run(total)
        if state["errors"]:
            raise RuntimeError(f"{state['errors']} ensemble requests failed: {state.get('first_error')}")
        lat = sorted(state["lat"])
        top = [e.decode() for e in state["first"].as_numpy("OUTPUT").reshape(b, 3)[0]]
        # the same eight images through client-side preprocessing and the classifier itself
        import torch

        pre = Preprocessor(224, 224, 3, "NCHW", "FP16", scaling="INCEPTION", device=device, max_batch=8)
        x = torch.empty(b, 3, 224, 224, device=f"cuda:{device}", dtype=torch.float16)
        pre(raw_first, x)
        pre.close()
        inp = grpcclient.InferInput("input", [b, 3, 224, 224], "FP16")
        inp.set_data_from_numpy(x.cpu().numpy())
        ref = control.infer(args.model, [inp], outputs=[grpcclient.InferRequestedOutput("output", class_count=3)])
        ref_top = [e.decode() for e in ref.as_numpy("output").reshape(b, 3)[0]]
        same = [t.split(":")[1] for t in top] == [t.split(":")[1] for t in ref_top]
        return {"images_per_sec": round(total * b / elapsed, 1), "requests": total, "batch": b, "concurrency": inflight,
                "elapsed_s": round(elapsed, 4), "p50_latency_us": round(lat[len(lat) // 2] * 1e6, 1),
                "p99_latency_us": round(lat[int(len(lat) * 0.99)] * 1e6, 1),
                "source": "224x224 PNG files inside the requests (BYTES), decoded by the server",
                "client": "Python (tritonclient.grpc async_infer, 4 channels)", "first_result_top3": top,
                "top3_equals_client_side_preprocessing": bool(same)}
    finally:
        control.close()
        for c in clients:
            c.close()


def measure_secondary(args, device, dist):
    """The other configurations of the benchmark set, each on this rank's own server and GPU:
    `simple` over HTTP + system shared memory, `simple` over gRPC with tensors on the wire,
    `simple_string` (BYTES) over gRPC + HIP shared memory, `repeat_int32` (decoupled, 16 responses
    per request) and `simple_sequence` (stateful, sequences of 20 requests) over gRPC streams.
    Closed loop at concurrency 16, two windows of ``--secondary-window-ms`` each; with
    several ranks the throughputs are summed over the GPUs and the latencies are the worst rank's."""
    import contextlib

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from secondary_bench import run_configs

    names = ["simple_http_sysshm", "simple_grpc_wire", "string_grpc_hipshm", "decoupled_stream", "sequence_stream"]
    rates, lats, error = [0.0] * len(names), [0.0] * (2 * len(names)), None
    try:
        with contextlib.redirect_stdout(sys.stderr):  # the analyzer reports on stdout; ours carries one JSON line
            lines = {line["config"]: line for line in run_configs(gpu=device, window_ms=args.secondary_window_ms)}
        for i, name in enumerate(names):
            line = lines[name]
            rates[i] = float(line["responses_per_sec"] if name == "decoupled_stream" else line["infer_per_sec"])
            lats[2 * i], lats[2 * i + 1] = float(line["p50_us"]), float(line["p99_us"])
    except BaseException as e:  # noqa: BLE001 - reported, and the collectives below still run on every rank
        error = f"{type(e).__name__}: {e}"[:500]
    ok = 0.0 if error else 1.0
    if dist is not None:
        t = torch.tensor(rates, device=args.reduce_device, dtype=torch.float64)
        dist.all_reduce(t)
        rates = [float(v) for v in t.tolist()]
        t = torch.tensor(lats + [-ok], device=args.reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lats, ok = [float(v) for v in t[:-1].tolist()], -float(t[-1].item())
    if ok < 1.0:
        return {"error": error or "a configuration failed on another rank"}
    out = {}
    for i, name in enumerate(names):
        key = "responses_per_sec" if name == "decoupled_stream" else "infer_per_sec"
        out[name] = {key: round(rates[i], 1), "p50_latency_us": round(lats[2 * i], 1),
                     "p99_latency_us": round(lats[2 * i + 1], 1), "concurrency_per_gpu": 16}
    out["decoupled_stream"]["responses_per_request"] = 16
    out["sequence_stream"]["sequence_length"] = 20
    return out


def measure_fanout(args, world):
    """``resnet50_sharded`` on GPU 0 with one replica process per further GPU: requests of
    ``32 x replicas`` images in HIP shared memory, split over the replicas and gathered into
    the caller's output region.  One run per transport (peer stores + device flags, RCCL
    point-to-point).  A failure is reported in the result, it does not fail the benchmark."""
    import torch

    from triton_client_amd.server import spawn_server

    if args.fanout_devices is not None:
        devices = [int(v) for v in args.fanout_devices.split(",") if v.strip() != ""]
    else:
        devices = list(range(1, world)) if world > 1 else [0, 0]
    replicas = 1 + len(devices)
    same_gpu = len(set(devices + [0])) < replicas
    batch = 32 * replicas
    out = {"replicas": replicas, "devices": [0] + devices, "batch_per_request": batch, "concurrency": 8,
           "note": "replicas share one GPU (rehearsal)" if same_gpu else "one replica per GPU",
           "parity": "the served logits are compared with a single-GPU forward of the same regions in this run "
                     "(check); no earlier measurement on more than one GPU exists to compare the rates with "
                     "(parity unpinned on >1 GPU)"}
    for mode in [m for m in args.fanout_modes.split(",") if m]:
        if mode == "rccl" and same_gpu:
            out[mode] = {"skipped": "RCCL needs one GPU per rank"}
            continue
        server = None
        try:
            t0 = time.perf_counter()
            server = spawn_server(models=["resnet50_sharded"], gpu=0, startup_timeout=200,
                                  env={"TCA_SHARD_DEVICES": ",".join(str(d) for d in devices), "TCA_SHARD_MODE": mode},
                                  extra_args=["--grpc-workers", "32"])
            start_s = time.perf_counter() - t0
            args.control_url = server.grpc_url
            url = server.grpc_url if args.protocol == "grpc" else server.http_url
            r = measure_native(args, url, 0, None, f"fanout-{mode}", mode="hip", steps=1,
                               per_step=args.fanout_requests, warmup_requests=64, model="resnet50_sharded",
                               batch=batch, concurrency=8, check="resnet50")
            out[mode] = {"infer_per_sec": round(r["infer_per_sec"], 1), "p50_latency_us": round(r["p50_us"], 1),
                         "p99_latency_us": round(r["p99_us"], 1), "requests": r["requests"],
                         "elapsed_s": round(r["elapsed_s"], 4), "startup_s": round(start_s, 1),
                         "check": r.get("check")}
        except BaseException as e:  # noqa: BLE001 - the headline number must survive
            out[mode] = {"error": f"{type(e).__name__}: {e}"[:500]}
        finally:
            if server is not None:
                server.stop(timeout=40)
            torch.cuda.synchronize()
    return out


def fanout_in_child(args, world):
    """Runs :func:`measure_fanout` in a process of its own with a hard time limit: the replicas, their
    IPC mappings and the RCCL group are set up across GPUs this process has never used, and nothing that
    goes wrong there - a hang included - may take the benchmark's main result with it."""
    import signal
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                            "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                                                            "TORCHELASTIC_RUN_ID")}
    merged = {}
    # one child per transport, each with its own limit: a transport that hangs costs its own share of the
    # time and leaves the other one's result alone
    for mode in [m for m in args.fanout_modes.split(",") if m]:
        cmd = [sys.executable, os.path.abspath(__file__), "--fanout-child", str(world), "--protocol", args.protocol,
               "--connections", str(args.connections), "--fanout-modes", mode, "--fanout-requests",
               str(args.fanout_requests), "--batch", str(args.batch)]
        if args.fanout_devices is not None:
            cmd += ["--fanout-devices", args.fanout_devices]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
        result = None
        try:
            out, _ = proc.communicate(timeout=args.fanout_timeout)
            for line in reversed(out.splitlines()):
                if line.startswith("{"):
                    result = json.loads(line)
                    break
            if result is None:
                result = {mode: {"error": f"fan-out process ended with code {proc.returncode} and no result"}}
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)  # the child, its server and the replica processes
            except ProcessLookupError:
                pass
            proc.wait()
            result = {mode: {"error": f"abandoned after {args.fanout_timeout:.0f} s"}}
        for key, value in result.items():
            merged.setdefault(key, value)
    return merged


def preflight(args, dist, rank, world, device):
    """Before anything is measured with several ranks: an all-reduce whose result only comes out right when
    every rank takes part, the identity of the GPU each rank holds (one GPU per rank on the RCCL backend, or
    the run stops here), the time of a small all-reduce and - rank 0 - the peer-access matrix of the visible
    GPUs (hipDeviceCanAccessPeer)."""
    import torch

    dev = args.reduce_device
    t = torch.full((8,), float(rank + 1), device=dev, dtype=torch.float64)
    dist.all_reduce(t)
    reduced_ok = bool((t == world * (world + 1) / 2).all().item())
    prop = torch.cuda.get_device_properties(device)
    ident = f"{int(prop.pci_domain_id):04x}:{int(prop.pci_bus_id):02x}:{int(prop.pci_device_id):02x}"
    held = [None] * world
    dist.all_gather_object(held, {"rank": rank, "gpu": ident, "ordinal": device, "name": prop.name})
    ranks_seen = sorted(h["rank"] for h in held)
    distinct = len({h["gpu"] for h in held})
    small = torch.ones(1024, device=dev)
    for _ in range(3):
        dist.all_reduce(small)
    if dev != "cpu":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dist.all_reduce(small)
    if dev != "cpu":
        torch.cuda.synchronize()
    all_reduce_us = (time.perf_counter() - t0) / 20 * 1e6
    rccl = args.dist_backend == "nccl"
    ok = reduced_ok and ranks_seen == list(range(world)) and (distinct == world or not rccl)
    n = torch.cuda.device_count()
    report = {"rccl_ranks_verified": bool(ok and rccl), "ranks_verified": bool(ok), "backend": "rccl" if rccl else "gloo",
              "ranks": world, "distinct_gpus": distinct, "gpus": [h["gpu"] for h in held],
              "all_reduce_4KB_us": round(all_reduce_us, 1),
              "peer_access": [[1 if a == b else int(torch.cuda.can_device_access_peer(a, b)) for b in range(n)]
                              for a in range(n)] if rank == 0 else None}
    if not ok:
        raise SystemExit(f"rank {rank}: preflight failed before measuring: {json.dumps(report)}")
    return report


def launch_ranks(args):
    """``--gpus N`` without a launcher: start the N ranks as children of this
    process, which has not touched the GPU and only relays.  Rank 0's JSON line
    goes to stdout; any rank failing fails the run."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:
        if line.startswith("{"):
            lines.append(line.strip())
        else:
            sys.stderr.write(line)
    code = proc.wait()
    if code != 0:
        raise SystemExit(code)
    if len(lines) != 1:
        raise SystemExit(f"expected one JSON line from rank 0, got {len(lines)}")
    print(lines[0], flush=True)


def pin_near_gpu(device):
    """Restricts this rank (and the server / load generator processes it starts) to the CPUs of the
    NUMA node its GPU hangs off, read from the GPU's PCI entry in sysfs.  Host memory of the rank is
    then allocated next to the GPU too (first touch), and eight ranks on a two-socket node do not
    wander across sockets.  Any doubt - unknown PCI address, a CPU list that leaves fewer than sixteen
    of the CPUs this process may use - leaves the affinity alone.  ``TCA_BENCH_PIN=0`` turns it off."""
    if os.environ.get("TCA_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import torch

        prop = torch.cuda.get_device_properties(device)
        bdf = f"{int(prop.pci_domain_id):04x}:{int(prop.pci_bus_id):02x}:{int(prop.pci_device_id):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            text = fh.read().strip()
        local = set()
        for part in text.split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            local.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        chosen = allowed & local
        if len(chosen) < 16 or len(chosen) == len(allowed):
            return None
        os.sched_setaffinity(0, chosen)
        return f"{bdf}: {len(chosen)} CPUs ({text})"
    except Exception as e:  # noqa: BLE001 - pinning is an optimisation, never a reason to fail
        return f"not pinned: {type(e).__name__}: {e}"[:200]


def main():
    args = parse_args()
    if args.fanout_child:
        args.control_url = None
        print(json.dumps(measure_fanout(args, args.fanout_child)), flush=True)
        return None
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): one rank per GPU is required")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs at least one MI355X")
    if args.dist_backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         "(--dist-backend gloo rehearses several ranks on one GPU)")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    pinned = pin_near_gpu(device)
    if pinned:
        print(f"[bench] rank {rank} on GPU {device}: {pinned}", file=sys.stderr, flush=True)
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
    checked = preflight(args, dist, rank, world, device) if dist is not None else None

    from triton_client_amd.server import spawn_server

    with_ensemble = (not args.no_extras and args.client == "native" and args.protocol == "grpc" and args.mode == "serve"
                     and world == 1 and args.model == "resnet50")
    server = spawn_server(models=[args.model] + (["preprocess_resnet50_ensemble"] if with_ensemble else []), gpu=device,
                          extra_args=["--grpc-workers", str(max(32, 2 * args.concurrency)), "--http-workers",
                                      str(max(16, 2 * args.concurrency))])
    try:
        url = server.grpc_url if args.protocol == "grpc" else server.http_url
        args.control_url = server.grpc_url
        run = measure_native if args.client == "native" else measure
        main_kwargs = {}
        if args.client == "native" and not args.no_extras and args.mode == "serve":
            main_kwargs = {"check": True, "sweep": [v for v in args.sweep.split(",") if v.strip()]}
        main_result = run(args, url, device, dist, f"{args.protocol}+{args.shared_memory}shm", **main_kwargs)
        extras = {}
        for key in ("check", "sweep"):
            if key in main_result:
                extras[key] = main_result.pop(key)
        if not args.no_extras and args.client == "native":
            # the same requests with the tensors inside the messages / in system shared memory:
            # short runs, same server, same generator, same concurrency
            for key, mode in (("wire", "none"), ("system_shm", "system")):
                if mode == args.shared_memory or args.mode == "image_client":
                    continue
                r = measure_native(args, url, device, dist, f"{args.protocol}+{mode}", mode=mode, steps=1,
                                   per_step=args.extra_requests, warmup_requests=max(64, args.extra_requests // 8))
                extras[key] = {"infer_per_sec": round(r["infer_per_sec"], 2), "p50_latency_us": round(r["p50_us"], 1),
                               "p99_latency_us": round(r["p99_us"], 1), "requests": r["requests"],
                               "elapsed_s": round(r["elapsed_s"], 4)}
                extras[key]["main_over_this"] = round(main_result["infer_per_sec"] / r["infer_per_sec"], 3)
            if args.protocol == "grpc" and (args.mode == "image_client" or world == 1):
                try:
                    extras["image_client"] = measure_image_client(args, url, device)
                except BaseException as e:  # noqa: BLE001 - the headline number must survive
                    extras["image_client"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            if with_ensemble:
                try:
                    extras["ensemble"] = measure_ensemble(args, server, device)
                except BaseException as e:  # noqa: BLE001 - the headline number must survive
                    extras["ensemble"] = {"error": f"{type(e).__name__}: {e}"[:500]}
    finally:
        server.stop()
    if dist is not None:
        dist.barrier()
    if not args.no_extras and args.client == "native" and args.mode == "serve":
        extras["secondary"] = measure_secondary(args, device, dist)  # every rank takes part (collectives inside)
    if not args.no_extras and args.client == "native" and args.mode != "fanout" and world > 1:
        extras["fanout"] = {"skipped": "opt-in (--mode fanout): the cross-GPU data plane of resnet50_sharded has "
                                       "only run with its replicas sharing one GPU; parity unpinned on >1 GPU"}
    if not args.no_extras and args.client == "native" and args.mode == "fanout":
        # every rank's server is down; rank 0 serves resnet50_sharded with one replica per GPU
        # the other ranks wait on the host (a collective barrier would spin on their GPUs,
        # which are the replicas' GPUs now)
        store = None
        if dist is not None:
            from datetime import timedelta

            store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            try:
                extras["fanout"] = fanout_in_child(args, world)
            finally:
                if store is not None:
                    store.set("tca_fanout_done", "1")
        elif store is not None:
            store.wait(["tca_fanout_done"], timedelta(minutes=40))
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
            "data": "synthetic structured images (375x500 u8, different in every slot) through the HIP preprocessing "
                    "kernel; random-init ResNet-50 weights",
            "p50_latency_us": round(main_result["p50_us"], 1),
            "p90_latency_us": round(main_result["p90_us"], 1),
            "p95_latency_us": round(main_result["p95_us"], 1),
            "p99_latency_us": round(main_result["p99_us"], 1),
            "ranks": n,
            "rank_backend": ("rccl" if args.dist_backend == "nccl" else "gloo") if n > 1 else None,
            "per_gpu_infer_per_sec": [round(v, 1) for v in main_result.get("per_gpu_infer_per_sec", [])],
            "config": {
                "model": args.model,
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
        if checked is not None:
            line["preflight"] = checked
        if extras:
            line["extra"] = extras
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

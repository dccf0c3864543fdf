#!/usr/bin/env python3
"""Where the time of ``preprocess_resnet50_ensemble`` goes: requests of eight 224x224 PNG files driven straight
into the serving core from N threads (no transport), with the phases of one request timed."""
import io
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    from PIL import Image

    import bench
    from triton_client_amd.server.core import InferenceCore, InferRequest, InputTensor, RequestedOutput

    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    core = InferenceCore(models=["resnet50", "preprocess_resnet50_ensemble"], gpu=0)
    blobs = []
    for img in bench.structured_images(8, first=9000, height=224, width=224):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "PNG", compress_level=3)
        blobs.append(b.getvalue())
    arr = np.array(blobs, dtype=np.object_).reshape(8, 1)

    def one():
        req = InferRequest("preprocess_resnet50_ensemble", inputs=[InputTensor("INPUT", "BYTES", [8, 1], data=arr)],
                           outputs=[RequestedOutput(name="OUTPUT", class_count=3)])
        return core.infer(req)

    for _ in range(20):
        one()
    model = core._get("preprocess_resnet50_ensemble").model
    phases = {}
    orig_exec = model.execute

    def timed_execute(inputs, ctx):
        t0 = time.perf_counter()
        out = orig_exec(inputs, ctx)
        phases.setdefault("execute", []).append(time.perf_counter() - t0)
        return out

    orig_step = core.infer_tensors

    def timed_step(*a, **k):
        t0 = time.perf_counter()
        out = orig_step(*a, **k)
        phases.setdefault("classifier step", []).append(time.perf_counter() - t0)
        return out

    model.execute = timed_execute
    core.infer_tensors = timed_step
    for n in (1, threads):
        for v in phases.values():
            v.clear()
        count = [0]
        stop = time.perf_counter() + 3.0

        def loop():
            while time.perf_counter() < stop:
                t0 = time.perf_counter()
                one()
                phases.setdefault("request", []).append(time.perf_counter() - t0)
                count[0] += 1

        ts = [threading.Thread(target=loop) for _ in range(n)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{n} threads: {count[0] * 8 / dt:.0f} img/s; mean ms: " +
              ", ".join(f"{k} {1e3 * sum(v) / max(1, len(v)):.2f}" for k, v in phases.items()), flush=True)
    core.shutdown()


if __name__ == "__main__":
    main()

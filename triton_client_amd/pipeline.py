"""End-to-end image classification pipeline on one GPU:

    decoded u8 images in page-locked memory
      -> one host-to-device copy + ONE fused HIP kernel (resize, scale, layout, cast)
      -> a slot of a HIP shared-memory region the server reads in place
      -> asynchronous gRPC inference asking for the top-k classes
      -> callback frees the slot

It is the pipelined form of what the reference's image client does one request at
a time on the CPU (``src/python/examples/image_client.py:153-192`` preprocess,
``:455-514`` batch / send / collect).  Each *lane* is a thread with its own
preprocessor (own HIP stream), its own region cut into ``depth`` batch slots and
its own gRPC channel; while the server works on the requests of the filled slots
the lane already preprocesses the next batch into a free one.  No tensor bytes
cross the wire in either direction except the top-k strings.
"""

from __future__ import annotations

import queue
import threading
import time
from functools import partial

import numpy as np


class ImagePipeline:
    """Classifies batches of decoded images through ``model`` served at ``url`` (gRPC).

    Parameters
    ----------
    url, model : server address (host:port of the gRPC endpoint) and model name
    batch : images per request
    geometry : (c, h, w, layout, dtype) of the model input
    scaling : "NONE" | "INCEPTION" | "VGG"
    classes : top-k entries to ask for (0: the raw output tensor)
    lanes, depth : threads, and batch slots per lane
    device : GPU ordinal of the preprocessor and the regions (the server's GPU)
    """

    def __init__(self, url, model, batch, geometry=(3, 224, 224, "NCHW", "FP16"), scaling="INCEPTION", classes=3,
                 lanes=2, depth=4, device=0, input_name="input", output_name="output", name_prefix="imgpipe"):
        import tritonclient.grpc as grpcclient
        import tritonclient.utils.cuda_shared_memory as hipshm

        from .ops.preprocess import Preprocessor

        c, h, w, layout, dtype = geometry
        self._grpc, self._hipshm = grpcclient, hipshm
        self.model, self.batch, self.depth = model, int(batch), int(depth)
        self._lanes = []
        dims = [c, h, w] if layout == "NCHW" else [h, w, c]
        for i in range(int(lanes)):
            pre = Preprocessor(h, w, c, layout, dtype, scaling, "pil", device=device, max_batch=self.batch)
            slot_bytes = pre.image_bytes * self.batch
            name = f"{name_prefix}_{i}"
            region = hipshm.create_shared_memory_region(name, slot_bytes * self.depth, device)
            client = grpcclient.InferenceServerClient(url)
            client.register_cuda_shared_memory(name, hipshm.get_raw_handle(region), device, slot_bytes * self.depth)
            inputs = []
            for slot in range(self.depth):
                inp = grpcclient.InferInput(input_name, [self.batch] + dims, dtype)
                inp.set_shared_memory(name, slot_bytes, offset=slot * slot_bytes)
                inputs.append([inp])
            free = queue.SimpleQueue()
            for slot in range(self.depth):
                free.put(slot)
            self._lanes.append(dict(pre=pre, region=region, client=client, name=name, inputs=inputs, free=free,
                                    slot_bytes=slot_bytes, batches=[]))
        self._outputs = [grpcclient.InferRequestedOutput(output_name, class_count=int(classes))]
        self.output_name = output_name

    def pin(self, image_batches):
        """Gives every lane its own pinned copy of ``image_batches`` (a list of lists of HWC u8 arrays)."""
        for lane in self._lanes:
            for b in lane["batches"]:
                b.close()
            lane["batches"] = [lane["pre"].pin(images) for images in image_batches]

    def run(self, requests, keep_results=0):
        """Issues ``requests`` requests over the lanes, cycling through the pinned batches.

        Returns a dict with the elapsed time, images per second, request latency
        percentiles (fill start to response) and the first ``keep_results`` results."""
        if not self._lanes or not self._lanes[0]["batches"]:
            raise RuntimeError("pin() the image batches first")
        lock = threading.Lock()
        state = dict(issued=0, done=0, error=None)
        latencies, kept = [], []
        finished = threading.Event()

        def on_done(lane, slot, t_start, result, error):
            now = time.perf_counter()
            lane["free"].put(slot)
            with lock:
                if error is not None and state["error"] is None:
                    state["error"] = error
                latencies.append(now - t_start)
                if result is not None and len(kept) < keep_results:
                    kept.append(result)
                state["done"] += 1
                if state["done"] == requests:
                    finished.set()

        def lane_loop(lane):
            pre, client, region = lane["pre"], lane["client"], lane["region"]
            batches, inputs, slot_bytes = lane["batches"], lane["inputs"], lane["slot_bytes"]
            k = 0
            while state["error"] is None:
                with lock:
                    if state["issued"] >= requests:
                        return
                    state["issued"] += 1
                slot = lane["free"].get()
                t_start = time.perf_counter()
                try:
                    pre.run_pinned(batches[k % len(batches)], region, out_offset=slot * slot_bytes)
                    client.async_infer(self.model, inputs[slot], partial(on_done, lane, slot, t_start),
                                       outputs=self._outputs)
                except Exception as e:  # noqa: BLE001 - reported through the result
                    on_done(lane, slot, t_start, None, e)
                k += 1

        threads = [threading.Thread(target=lane_loop, args=(lane,), daemon=True) for lane in self._lanes]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if not finished.wait(timeout=120) and state["error"] is None:
            state["error"] = TimeoutError(f"{requests - state['done']} responses missing after 120 s")
        elapsed = time.perf_counter() - t0
        if state["error"] is not None:
            raise state["error"]
        lat = np.asarray(latencies) * 1e6
        return dict(requests=requests, images=requests * self.batch, elapsed_s=elapsed,
                    images_per_sec=requests * self.batch / elapsed,
                    p50_us=float(np.percentile(lat, 50)), p99_us=float(np.percentile(lat, 99)), results=kept)

    def close(self):
        for lane in self._lanes:
            try:
                lane["client"].unregister_cuda_shared_memory(lane["name"])
            except Exception:  # noqa: BLE001 - server may already be gone
                pass
            lane["client"].close()
            self._hipshm.destroy_shared_memory_region(lane["region"])
            for b in lane["batches"]:
                b.close()
            lane["pre"].close()
        self._lanes = []

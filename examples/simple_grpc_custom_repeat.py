#!/usr/bin/env python3
"""Decoupled model over a gRPC stream: ONE request to ``repeat_int32`` produces
``repeat_count`` responses.  (Reference: ``src/python/examples/simple_grpc_custom_repeat.py``.)"""
import queue
from functools import partial

import numpy as np
from _common import fail, parser

import tritonclient.grpc as grpcclient


def callback(done, result, error):
    done.put(error if error else result)


if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("--repeat-count", type=int, default=10)
    ap.add_argument("--delay-ms", type=int, default=20)
    flags = ap.parse_args()
    n, offset = flags.repeat_count, 100
    values = np.arange(offset, offset + n, dtype=np.int32)
    inputs = [grpcclient.InferInput("IN", [n], "INT32").set_data_from_numpy(values),
              grpcclient.InferInput("DELAY", [n], "UINT32").set_data_from_numpy(
                  np.full([n], flags.delay_ms, dtype=np.uint32)),
              grpcclient.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(
                  np.array([flags.delay_ms], dtype=np.uint32))]
    done = queue.Queue()
    received = []
    with grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose) as client:
        client.start_stream(callback=partial(callback, done))
        client.async_stream_infer("repeat_int32", inputs, outputs=[grpcclient.InferRequestedOutput("OUT")])
        while len(received) < n:
            item = done.get(timeout=30)
            if isinstance(item, Exception):
                fail(str(item))
            received.append(item.as_numpy("OUT"))
    for i, out in enumerate(received):
        if len(out) != 1:
            fail(f"unexpected number of elements in the output, expected 1, got {len(out)}")
        print(f"{out[0]} : {offset + i}")
        if out[0] != offset + i:
            fail("mismatch in the results")
    print("PASS: Decoupled API")

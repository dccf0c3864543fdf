#!/usr/bin/env python3
"""``async_infer`` over gRPC: the callback receives ``(result, error)``.
(Reference: ``src/python/examples/simple_grpc_async_infer_client.py``.)"""
import queue
from functools import partial

from _common import check_simple, fail, parser, simple_inputs, simple_tensors

import tritonclient.grpc as grpcclient


def on_complete(done, result, error):
    done.put(error if error is not None else result)


if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    client = grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    a, b = simple_tensors()
    inputs = simple_inputs(grpcclient, a, b)
    outputs = [grpcclient.InferRequestedOutput("OUTPUT0"), grpcclient.InferRequestedOutput("OUTPUT1")]
    done = queue.Queue()
    n = 8
    for i in range(n):
        client.async_infer("simple", inputs, partial(on_complete, done), outputs=outputs, request_id=str(i),
                           client_timeout=flags.client_timeout)
    for i in range(n):
        item = done.get(timeout=30)
        if isinstance(item, Exception):
            fail(str(item))
        check_simple(item.as_numpy("OUTPUT0"), item.as_numpy("OUTPUT1"), a, b, quiet=i > 0)
    client.close()
    print("PASS: Async infer")

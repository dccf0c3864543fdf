#!/usr/bin/env python3
"""Creates a new client per iteration (or reuses one with ``--reuse``) for N
inferences on ``custom_identity_int32`` and reports resident-memory growth.
(Reference: ``src/python/examples/memory_growth_test.py``.)"""
import resource

import numpy as np
from _common import fail, parser

import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient


def infer_once(mod, client, data):
    inp = mod.InferInput("INPUT0", list(data.shape), "INT32").set_data_from_numpy(data)
    out = client.infer("custom_identity_int32", [inp]).as_numpy("OUTPUT0")
    if not np.array_equal(out, data):
        fail("identity output differs from its input")


if __name__ == "__main__":
    ap = parser(__doc__, 8000, protocol_flag=True)
    ap.add_argument("-r", "--repetitions", type=int, default=200)
    ap.add_argument("--reuse", action="store_true", help="reuse one client instead of creating one per request")
    ap.add_argument("--max-growth-kb", type=int, default=32768)
    flags = ap.parse_args()
    mod = grpcclient if flags.protocol.lower() == "grpc" else httpclient
    data = np.arange(16, dtype=np.int32).reshape(1, 16)
    shared = mod.InferenceServerClient(flags.url) if flags.reuse else None
    for _ in range(20):  # warm up allocators and caches before sampling
        c = shared or mod.InferenceServerClient(flags.url)
        infer_once(mod, c, data)
        if shared is None:
            c.close()
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(flags.repetitions):
        c = shared or mod.InferenceServerClient(flags.url)
        infer_once(mod, c, data)
        if shared is None:
            c.close()
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print(f"max RSS {before} kB -> {after} kB over {flags.repetitions} requests")
    if after - before > flags.max_growth_kb:
        fail("memory grew more than allowed")
    if shared is not None:
        shared.close()
    print("PASS: memory growth")

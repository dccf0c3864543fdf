#!/usr/bin/env python3
"""The same InferInput / InferRequestedOutput objects alternate between data on
the wire and system shared memory, over HTTP and gRPC.  (Reference:
``src/python/examples/reuse_infer_objects_client.py``.)"""
import numpy as np
from _common import check_simple, parser, simple_tensors

import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient
import tritonclient.utils.shared_memory as shm


def exercise(mod, client, tag):
    a, b = simple_tensors()
    nbytes = a.nbytes
    client.unregister_system_shared_memory()
    hin = shm.create_shared_memory_region(f"in_{tag}", f"/reuse_in_{tag}", nbytes * 2)
    hout = shm.create_shared_memory_region(f"out_{tag}", f"/reuse_out_{tag}", nbytes * 2)
    shm.set_shared_memory_region(hin, [a, b])
    client.register_system_shared_memory(f"in_{tag}", f"/reuse_in_{tag}", nbytes * 2)
    client.register_system_shared_memory(f"out_{tag}", f"/reuse_out_{tag}", nbytes * 2)
    i0 = mod.InferInput("INPUT0", [1, 16], "INT32")
    i1 = mod.InferInput("INPUT1", [1, 16], "INT32")
    o0 = mod.InferRequestedOutput("OUTPUT0")
    o1 = mod.InferRequestedOutput("OUTPUT1")
    try:
        for round_ in range(4):
            if round_ % 2 == 0:  # wire
                i0.set_data_from_numpy(a)
                i1.set_data_from_numpy(b)
                o0.unset_shared_memory()
                o1.unset_shared_memory()
                r = client.infer("simple", [i0, i1], outputs=[o0, o1])
                check_simple(r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1"), a, b, quiet=True)
            else:  # shared memory
                i0.set_shared_memory(f"in_{tag}", nbytes)
                i1.set_shared_memory(f"in_{tag}", nbytes, offset=nbytes)
                o0.set_shared_memory(f"out_{tag}", nbytes)
                o1.set_shared_memory(f"out_{tag}", nbytes, offset=nbytes)
                client.infer("simple", [i0, i1], outputs=[o0, o1])
                check_simple(shm.get_contents_as_numpy(hout, np.int32, [1, 16]),
                             shm.get_contents_as_numpy(hout, np.int32, [1, 16], offset=nbytes), a, b, quiet=True)
    finally:
        client.unregister_system_shared_memory()
        shm.destroy_shared_memory_region(hin)
        shm.destroy_shared_memory_region(hout)


if __name__ == "__main__":
    ap = parser(__doc__, 8000)
    ap.add_argument("--grpc-url", default="localhost:8001")
    flags = ap.parse_args()
    with httpclient.InferenceServerClient(flags.url, verbose=flags.verbose) as c:
        exercise(httpclient, c, "http")
    with grpcclient.InferenceServerClient(flags.grpc_url, verbose=flags.verbose) as c:
        exercise(grpcclient, c, "grpc")
    print("PASS: reuse infer objects")

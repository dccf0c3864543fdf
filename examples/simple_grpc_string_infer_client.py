#!/usr/bin/env python3
"""BYTES tensors over gRPC on ``simple_string`` / ``simple_identity``.
(Reference: ``src/python/examples/simple_grpc_string_infer_client.py``.)"""
import numpy as np
from _common import check_string, fail, parser, string_tensors

import tritonclient.grpc as grpcclient

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    client = grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    ints, in0, in1 = string_tensors()
    i0 = grpcclient.InferInput("INPUT0", [1, 16], "BYTES").set_data_from_numpy(in0)
    i1 = grpcclient.InferInput("INPUT1", [1, 16], "BYTES").set_data_from_numpy(in1)
    result = client.infer("simple_string", [i0, i1], outputs=[grpcclient.InferRequestedOutput("OUTPUT0"),
                                                                grpcclient.InferRequestedOutput("OUTPUT1")])
    check_string(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), ints)
    payload = np.array([[b"a\x00b", b"", b"\x00\x00", b"xyz"]], dtype=np.object_)
    ident = grpcclient.InferInput("INPUT0", [1, 4], "BYTES").set_data_from_numpy(payload)
    if client.infer("simple_identity", [ident]).as_numpy("OUTPUT0").tolist() != payload.tolist():
        fail("identity output differs from its input")
    fixed = np.array([[b"hello", b"world"]], dtype=np.bytes_)
    ident = grpcclient.InferInput("INPUT0", [1, 2], "BYTES").set_data_from_numpy(fixed)
    if client.infer("simple_identity", [ident]).as_numpy("OUTPUT0").tolist() != [[b"hello", b"world"]]:
        fail("np.bytes_ identity mismatch")
    client.close()
    print("PASS: string")

#!/usr/bin/env python3
"""BYTES tensors over HTTP: ``simple_string`` in binary and JSON form and
``simple_identity`` with embedded NULs.  (Reference:
``src/python/examples/simple_http_string_infer_client.py``.)"""
import numpy as np
from _common import check_string, fail, parser, string_tensors

import tritonclient.http as httpclient

if __name__ == "__main__":
    flags = parser(__doc__, 8000).parse_args()
    client = httpclient.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    ints, in0, in1 = string_tensors()
    for binary in (True, False):
        i0 = httpclient.InferInput("INPUT0", [1, 16], "BYTES").set_data_from_numpy(in0, binary_data=binary)
        i1 = httpclient.InferInput("INPUT1", [1, 16], "BYTES").set_data_from_numpy(in1, binary_data=binary)
        outputs = [httpclient.InferRequestedOutput("OUTPUT0", binary_data=binary),
                   httpclient.InferRequestedOutput("OUTPUT1", binary_data=binary)]
        result = client.infer("simple_string", [i0, i1], outputs=outputs)
        check_string(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), ints)
    # bytes with NULs survive the round trip; np.bytes_ arrays are accepted as well
    payload = np.array([[b"a\x00b", b"", b"\x00\x00", b"xyz"]], dtype=np.object_)
    ident = httpclient.InferInput("INPUT0", [1, 4], "BYTES").set_data_from_numpy(payload)
    echoed = client.infer("simple_identity", [ident]).as_numpy("OUTPUT0")
    if echoed.tolist() != payload.tolist():
        fail("identity output differs from its input")
    fixed = np.array([[b"hello", b"world"]], dtype=np.bytes_)
    ident = httpclient.InferInput("INPUT0", [1, 2], "BYTES").set_data_from_numpy(fixed)
    if client.infer("simple_identity", [ident]).as_numpy("OUTPUT0").tolist() != [[b"hello", b"world"]]:
        fail("np.bytes_ identity mismatch")
    client.close()
    print("PASS: string")

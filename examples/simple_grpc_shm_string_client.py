#!/usr/bin/env python3
"""BYTES tensors in system shared memory over GRPC.
(Reference: ``src/python/examples/simple_grpc_shm_string_client.py``.)"""
from _common import parser
from _shm_example import run_string

import tritonclient.grpc as client_module

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    run_string(client_module, client, "system")
    client.close()
    print("PASS: system shared memory")

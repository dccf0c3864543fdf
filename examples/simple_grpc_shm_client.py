#!/usr/bin/env python3
"""System shared memory over GRPC: two tensors per region via offsets.
(Reference: ``src/python/examples/simple_grpc_shm_client.py``.)"""
from _common import parser
from _shm_example import run_simple

import tritonclient.grpc as client_module

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    run_simple(client_module, client, "system")
    client.close()
    print("PASS: system shared memory")

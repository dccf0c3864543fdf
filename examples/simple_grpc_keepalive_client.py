#!/usr/bin/env python3
"""gRPC client with explicit ``KeepAliveOptions``.  (Reference:
``src/python/examples/simple_grpc_keepalive_client.py``.)"""
from _common import check_simple, parser, simple_inputs, simple_tensors

import tritonclient.grpc as grpcclient

if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("--grpc-keepalive-time", type=int, default=2**31 - 1)
    ap.add_argument("--grpc-keepalive-timeout", type=int, default=20000)
    ap.add_argument("--grpc-keepalive-permit-without-calls", action="store_true")
    ap.add_argument("--grpc-http2-max-pings-without-data", type=int, default=2)
    flags = ap.parse_args()
    options = grpcclient.KeepAliveOptions(
        keepalive_time_ms=flags.grpc_keepalive_time, keepalive_timeout_ms=flags.grpc_keepalive_timeout,
        keepalive_permit_without_calls=flags.grpc_keepalive_permit_without_calls,
        http2_max_pings_without_data=flags.grpc_http2_max_pings_without_data)
    client = grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose, keepalive_options=options)
    a, b = simple_tensors()
    result = client.infer("simple", simple_inputs(grpcclient, a, b))
    check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
    client.close()
    print("PASS: KeepAlive")

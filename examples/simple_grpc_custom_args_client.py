#!/usr/bin/env python3
"""gRPC client built from raw ``channel_args`` (they replace the defaults and
the keep-alive options).  (Reference: ``src/python/examples/simple_grpc_custom_args_client.py``.)"""
from _common import check_simple, parser, simple_inputs, simple_tensors

import tritonclient.grpc as grpcclient

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    channel_args = [
        ("grpc.max_send_message_length", 1024 * 1024 * 8),
        ("grpc.max_receive_message_length", 1024 * 1024 * 8),
        ("grpc.keepalive_time_ms", 2**31 - 1),
        ("grpc.keepalive_timeout_ms", 20000),
        ("grpc.keepalive_permit_without_calls", False),
        ("grpc.http2.max_pings_without_data", 2),
        ("grpc.lb_policy_name", "round_robin"),
    ]
    client = grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose, channel_args=channel_args)
    a, b = simple_tensors()
    result = client.infer("simple", simple_inputs(grpcclient, a, b))
    check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
    client.close()
    print("PASS: custom channel args")

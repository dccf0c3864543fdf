#!/usr/bin/env python3
"""Synchronous gRPC inference on ``simple`` with optional TLS, compression and
client timeout; checks statistics and the unknown-model error.  (Reference:
``src/python/examples/simple_grpc_infer_client.py``.)"""
from _common import check_simple, fail, headers_of, parser, simple_inputs, simple_tensors

import tritonclient.grpc as grpcclient
from tritonclient.utils import InferenceServerException

if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("-s", "--ssl", action="store_true")
    ap.add_argument("-C", "--grpc-compression-algorithm", default=None, help="deflate | gzip")
    ap.add_argument("-c", "--certificate-chain", default=None)
    ap.add_argument("-p", "--private-key", default=None)
    ap.add_argument("-r", "--root-certificates", default=None)
    flags = ap.parse_args()
    try:
        client = grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose, ssl=flags.ssl,
                                                  root_certificates=flags.root_certificates,
                                                  private_key=flags.private_key,
                                                  certificate_chain=flags.certificate_chain)
    except Exception as e:
        fail("channel creation failed: " + str(e))
    a, b = simple_tensors()
    inputs = simple_inputs(grpcclient, a, b)
    outputs = [grpcclient.InferRequestedOutput("OUTPUT0"), grpcclient.InferRequestedOutput("OUTPUT1")]
    result = client.infer("simple", inputs, outputs=outputs, client_timeout=flags.client_timeout,
                          headers=headers_of(flags) or {"test": "1"},
                          compression_algorithm=flags.grpc_compression_algorithm)
    check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
    for alg in ("gzip", "deflate"):
        r = client.infer("simple", inputs, compression_algorithm=alg)
        check_simple(r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1"), a, b, quiet=True)
    stats = client.get_inference_statistics(model_name="simple")
    if len(stats.model_stats) != 1:
        fail("FAILED: Inference Statistics")
    try:
        client.infer("wrong_model_name", inputs)
        fail("expected an error for an unknown model")
    except InferenceServerException as e:
        if "Request for unknown model" not in e.message():
            fail("unexpected error: " + str(e))
    client.close()
    print("PASS: infer")

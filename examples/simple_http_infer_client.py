#!/usr/bin/env python3
"""Synchronous HTTP inference on ``simple``: binary and JSON tensors, query
parameters, headers, compression, no-outputs request, statistics and the
unknown-model error.  (Reference: ``src/python/examples/simple_http_infer_client.py``.)"""
from _common import check_simple, fail, headers_of, parser, simple_inputs, simple_tensors

import tritonclient.http as httpclient
from tritonclient.utils import InferenceServerException


def run(client, inputs, outputs, a, b, **kw):
    result = client.infer("simple", inputs, outputs=outputs, **kw)
    check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b, quiet=True)
    return result


if __name__ == "__main__":
    ap = parser(__doc__, 8000)
    ap.add_argument("-s", "--ssl", action="store_true", help="Enable encrypted link to the server using HTTPS")
    ap.add_argument("--key-file", default=None)
    ap.add_argument("--cert-file", default=None)
    ap.add_argument("--ca-certs", default=None)
    ap.add_argument("--insecure", action="store_true")
    ap.add_argument("--request-compression-algorithm", default=None, help="deflate | gzip")
    ap.add_argument("--response-compression-algorithm", default=None, help="deflate | gzip")
    flags = ap.parse_args()
    try:
        ssl_options = {k: v for k, v in (("keyfile", flags.key_file), ("certfile", flags.cert_file),
                                         ("ca_certs", flags.ca_certs)) if v}
        client = httpclient.InferenceServerClient(url=flags.url, verbose=flags.verbose, ssl=flags.ssl,
                                                  ssl_options=ssl_options or None, insecure=flags.insecure)
    except Exception as e:
        fail("channel creation failed: " + str(e))
    a, b = simple_tensors()
    headers = headers_of(flags)
    outputs = [httpclient.InferRequestedOutput("OUTPUT0", binary_data=True),
               httpclient.InferRequestedOutput("OUTPUT1", binary_data=False)]
    # INPUT0 binary, INPUT1 as JSON; OUTPUT0 binary, OUTPUT1 as JSON
    result = run(client, simple_inputs(httpclient, a, b, binary=(True, False)), outputs, a, b,
                 query_params={"test_1": 1, "test_2": 2}, headers=headers, request_id="1",
                 request_compression_algorithm=flags.request_compression_algorithm,
                 response_compression_algorithm=flags.response_compression_algorithm)
    check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
    if "data" not in result.get_output("OUTPUT1") or "data" in result.get_output("OUTPUT0"):
        fail("unexpected tensor representation in the response")
    # both compression schemes, both directions
    for alg in ("gzip", "deflate"):
        run(client, simple_inputs(httpclient, a, b), outputs, a, b, request_compression_algorithm=alg,
            response_compression_algorithm=alg)
    # no outputs requested: the server returns everything in binary form
    run(client, simple_inputs(httpclient, a, b), None, a, b, headers=headers)
    stats = client.get_inference_statistics(model_name="simple", headers=headers)
    if len(stats["model_stats"]) != 1:
        fail("FAILED: Inference Statistics")
    try:
        client.infer("wrong_model_name", simple_inputs(httpclient, a, b))
        fail("expected an error for an unknown model")
    except InferenceServerException as e:
        if "Request for unknown model" not in e.message():
            fail("unexpected error: " + str(e))
    client.close()
    print("PASS: infer")

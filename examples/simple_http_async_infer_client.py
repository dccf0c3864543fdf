#!/usr/bin/env python3
"""``async_infer`` over HTTP: several requests in flight on a pool of
connections, results fetched with ``get_result()``.  (Reference:
``src/python/examples/simple_http_async_infer_client.py``.)"""
from _common import check_simple, fail, parser, simple_inputs, simple_tensors

import tritonclient.http as httpclient

if __name__ == "__main__":
    flags = parser(__doc__, 8000).parse_args()
    request_count = 8
    try:
        # one connection per outstanding request
        client = httpclient.InferenceServerClient(url=flags.url, verbose=flags.verbose, concurrency=request_count)
    except Exception as e:
        fail("context creation failed: " + str(e))
    a, b = simple_tensors()
    inputs = simple_inputs(httpclient, a, b)
    outputs = [httpclient.InferRequestedOutput("OUTPUT0"), httpclient.InferRequestedOutput("OUTPUT1")]
    pending = [client.async_infer("simple", inputs, outputs=outputs, request_id=str(i)) for i in range(request_count)]
    for i, handle in enumerate(pending):
        result = handle.get_result()
        if result.get_response().get("id") != str(i):
            fail("responses must match their requests")
        check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b, quiet=i > 0)
    client.close()
    print("PASS: async infer")

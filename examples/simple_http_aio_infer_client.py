#!/usr/bin/env python3
"""asyncio HTTP client.  (Reference: ``src/python/examples/simple_http_aio_infer_client.py``.)"""
import asyncio

from _common import check_simple, parser, simple_inputs, simple_tensors

import tritonclient.http.aio as httpclient


async def main(flags):
    async with httpclient.InferenceServerClient(url=flags.url, verbose=flags.verbose) as client:
        a, b = simple_tensors()
        inputs = simple_inputs(httpclient, a, b)
        outputs = [httpclient.InferRequestedOutput("OUTPUT0"), httpclient.InferRequestedOutput("OUTPUT1")]
        assert await client.is_server_live() and await client.is_model_ready("simple")
        result = await client.infer("simple", inputs, outputs=outputs, request_id="1")
        check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
        # several requests concurrently on the same session
        results = await asyncio.gather(*[client.infer("simple", inputs, outputs=outputs) for _ in range(8)])
        for r in results:
            check_simple(r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1"), a, b, quiet=True)
    print("PASS: infer")


if __name__ == "__main__":
    asyncio.run(main(parser(__doc__, 8000).parse_args()))

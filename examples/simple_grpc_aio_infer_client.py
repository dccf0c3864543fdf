#!/usr/bin/env python3
"""asyncio gRPC client.  (Reference: ``src/python/examples/simple_grpc_aio_infer_client.py``.)"""
import asyncio

from _common import check_simple, parser, simple_inputs, simple_tensors

import tritonclient.grpc.aio as grpcclient


async def main(flags):
    async with grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose) as client:
        a, b = simple_tensors()
        inputs = simple_inputs(grpcclient, a, b)
        outputs = [grpcclient.InferRequestedOutput("OUTPUT0"), grpcclient.InferRequestedOutput("OUTPUT1")]
        assert await client.is_server_live() and await client.is_model_ready("simple")
        result = await client.infer("simple", inputs, outputs=outputs, client_timeout=flags.client_timeout,
                                    headers={"test": "1"})
        check_simple(result.as_numpy("OUTPUT0"), result.as_numpy("OUTPUT1"), a, b)
        results = await asyncio.gather(*[client.infer("simple", inputs, outputs=outputs) for _ in range(8)])
        for r in results:
            check_simple(r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1"), a, b, quiet=True)
        stats = await client.get_inference_statistics(model_name="simple")
        assert len(stats.model_stats) == 1
    print("PASS: infer")


if __name__ == "__main__":
    asyncio.run(main(parser(__doc__, 8001).parse_args()))

#!/usr/bin/env python3
"""Sequences over the asyncio gRPC ``stream_infer``: an async generator yields
one dict per request; responses are iterated as ``(result, error)``.
(Reference: ``src/python/examples/simple_grpc_aio_sequence_stream_infer_client.py``.)"""
import asyncio
import uuid

import numpy as np
from _common import fail, parser
from _sequence_example import VALUES, check

import tritonclient.grpc.aio as grpcclient


async def main(flags):
    int_model = "simple_dyna_sequence" if flags.dyna else "simple_sequence"
    str_model = "simple_string_dyna_sequence" if flags.dyna else "simple_sequence"
    id0, id1 = 1000 + flags.offset * 2, 1001 + flags.offset * 2
    sid = str(1002) if flags.dyna else str(uuid.uuid4())
    series = [(id0, int_model, [0] + VALUES), (id1, int_model, [100] + [-v for v in VALUES]),
              (sid, str_model, [20] + [-v for v in VALUES])]
    total = len(VALUES) + 1

    async def requests():
        for i in range(total):
            for seq_id, model, values in series:
                data = np.full([1, 1], values[i], dtype=np.int32)
                inp = grpcclient.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(data)
                yield {"model_name": model, "inputs": [inp], "outputs": [grpcclient.InferRequestedOutput("OUTPUT")],
                       "request_id": f"{seq_id}_{i}", "sequence_id": seq_id, "sequence_start": i == 0,
                       "sequence_end": i == total - 1}

    got = {str(id0): [], str(id1): [], sid: []}
    async with grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose) as client:
        responses = client.stream_infer(inputs_iterator=requests(), stream_timeout=flags.client_timeout)
        count = 0
        async for result, error in responses:
            if error:
                fail(str(error))
            got[result.get_response().id.rsplit("_", 1)[0]].append(int(result.as_numpy("OUTPUT")[0][0]))
            count += 1
            if count == 3 * total:
                break
    check(got[str(id0)], got[str(id1)], got[sid], id0, id1, sid, flags.dyna)
    print("PASS: Sequence")


if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("-d", "--dyna", action="store_true")
    ap.add_argument("-o", "--offset", type=int, default=0)
    asyncio.run(main(ap.parse_args()))

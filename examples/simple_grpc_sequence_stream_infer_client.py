#!/usr/bin/env python3
"""Sequences over ONE gRPC bidirectional stream: three sequences are
interleaved request by request and responses arrive through the stream
callback.  (Reference: ``src/python/examples/simple_grpc_sequence_stream_infer_client.py``.)"""
import queue
import uuid
from functools import partial

import numpy as np
from _common import fail, parser
from _sequence_example import VALUES, check

import tritonclient.grpc as grpcclient


def callback(done, result, error):
    done.put(error if error else result)


def enqueue(client, value, sequence_id, model, index, total):
    data = np.full([1, 1], value, dtype=np.int32)
    inp = grpcclient.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(data)
    client.async_stream_infer(model, [inp], outputs=[grpcclient.InferRequestedOutput("OUTPUT")],
                              request_id=f"{sequence_id}_{index}", sequence_id=sequence_id,
                              sequence_start=(index == 0), sequence_end=(index == total - 1))


if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("-d", "--dyna", action="store_true")
    ap.add_argument("-o", "--offset", type=int, default=0)
    flags = ap.parse_args()
    int_model = "simple_dyna_sequence" if flags.dyna else "simple_sequence"
    str_model = "simple_string_dyna_sequence" if flags.dyna else "simple_sequence"
    id0, id1 = 1000 + flags.offset * 2, 1001 + flags.offset * 2
    sid = str(1002) if flags.dyna else str(uuid.uuid4())
    series = {id0: [0] + VALUES, id1: [100] + [-v for v in VALUES], sid: [20] + [-v for v in VALUES]}
    done = queue.Queue()
    with grpcclient.InferenceServerClient(url=flags.url, verbose=flags.verbose) as client:
        client.start_stream(callback=partial(callback, done), stream_timeout=flags.client_timeout)
        total = len(VALUES) + 1
        for i in range(total):
            enqueue(client, series[id0][i], id0, int_model, i, total)
            enqueue(client, series[id1][i], id1, int_model, i, total)
            enqueue(client, series[sid][i], sid, str_model, i, total)
        got = {str(id0): [], str(id1): [], sid: []}
        for _ in range(3 * total):
            item = done.get(timeout=30)
            if isinstance(item, Exception):
                fail(str(item))
            request_id = item.get_response().id
            got[request_id.rsplit("_", 1)[0]].append(int(item.as_numpy("OUTPUT")[0][0]))
        client.stop_stream()
    check(got[str(id0)], got[str(id1)], got[sid], id0, id1, sid, flags.dyna)
    print("PASS: Sequence")

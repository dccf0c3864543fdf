#!/usr/bin/env python3
"""Stateful sequences with synchronous HTTP requests: two integer-id
sequences and one string-id sequence (``-d`` selects the dyna models).
(Reference: ``src/python/examples/simple_http_sequence_sync_infer_client.py``.)"""
import uuid

import numpy as np
from _common import parser
from _sequence_example import VALUES, check

import tritonclient.http as client_module


def send(client, results, values, sequence_id, model):
    for i, value in enumerate(values):
        data = np.full([1, 1], value, dtype=np.int32)
        inp = client_module.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(data)
        out = client_module.InferRequestedOutput("OUTPUT")
        r = client.infer(model, [inp], outputs=[out], sequence_id=sequence_id, sequence_start=(i == 0),
                         sequence_end=(i == len(values) - 1))
        results.append(int(r.as_numpy("OUTPUT")[0][0]))


if __name__ == "__main__":
    ap = parser(__doc__, 8000)
    ap.add_argument("-d", "--dyna", action="store_true", help="Assume dynamic sequence model")
    ap.add_argument("-o", "--offset", type=int, default=0, help="Add offset to sequence ID used")
    flags = ap.parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    int_model = "simple_dyna_sequence" if flags.dyna else "simple_sequence"
    str_model = "simple_string_dyna_sequence" if flags.dyna else "simple_sequence"
    id0, id1 = 1000 + flags.offset * 2, 1001 + flags.offset * 2
    sid = str(1002) if flags.dyna else str(uuid.uuid4())
    r0, r1, rs = [], [], []
    send(client, r0, [0] + VALUES, id0, int_model)
    send(client, r1, [100] + [-v for v in VALUES], id1, int_model)
    send(client, rs, [20] + [-v for v in VALUES], sid, str_model)
    check(r0, r1, rs, id0, id1, sid, flags.dyna)
    client.close()
    print("PASS: Sequence")

"""Helpers shared by the example programs: command line flags with the
conventional spellings (-v, -u, -i, -t, -H 'K:V'), tensors for the ``simple``
add/sub model and result checking.  Every example prints ``PASS...`` and exits
non-zero on a mismatch, so the examples double as integration tests."""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parser(description, default_port, protocol_flag=False):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("-v", "--verbose", action="store_true", default=False, help="Enable verbose output")
    ap.add_argument("-u", "--url", type=str, default=f"localhost:{default_port}",
                    help=f"Inference server URL. Default is localhost:{default_port}.")
    ap.add_argument("-t", "--client-timeout", type=float, default=None, help="Client timeout in seconds")
    ap.add_argument("-H", dest="http_headers", metavar="HTTP_HEADER", action="append",
                    help="HTTP headers to add to the request, format 'Header:Value' (may be repeated)")
    if protocol_flag:
        ap.add_argument("-i", "--protocol", type=str, default="HTTP", help="Protocol (HTTP/gRPC). Default is HTTP.")
    return ap


def headers_of(flags):
    out = {}
    for item in getattr(flags, "http_headers", None) or []:
        key, _, value = item.partition(":")
        out[key.strip()] = value.strip()
    return out or None


def fail(msg):
    print(msg)
    sys.exit(1)


def simple_tensors(batch=1):
    a = np.arange(16 * batch, dtype=np.int32).reshape(batch, 16)
    b = np.ones((batch, 16), dtype=np.int32)
    return a, b


def simple_inputs(mod, a, b, binary=(True, True)):
    i0 = mod.InferInput("INPUT0", list(a.shape), "INT32")
    i1 = mod.InferInput("INPUT1", list(b.shape), "INT32")
    if mod.__name__.split(".")[-1] == "http" or mod.__name__.endswith("http.aio"):
        i0.set_data_from_numpy(a, binary_data=binary[0])
        i1.set_data_from_numpy(b, binary_data=binary[1])
    else:
        i0.set_data_from_numpy(a)
        i1.set_data_from_numpy(b)
    return [i0, i1]


def check_simple(out0, out1, a, b, quiet=False):
    if out0 is None or out1 is None:
        fail("missing OUTPUT0 / OUTPUT1 in the response")
    for i in range(a.shape[-1]):
        if not quiet:
            print(f"{a[0][i]} + {b[0][i]} = {out0[0][i]}")
            print(f"{a[0][i]} - {b[0][i]} = {out1[0][i]}")
    if not np.array_equal(out0, a + b):
        fail("sync infer error: incorrect sum")
    if not np.array_equal(out1, a - b):
        fail("sync infer error: incorrect difference")


def string_tensors():
    ints = np.arange(16, dtype=np.int32).reshape(1, 16)
    in0 = np.array([str(x).encode("utf-8") for x in ints.reshape(-1)], dtype=np.object_).reshape(1, 16)
    in1 = np.array([b"1"] * 16, dtype=np.object_).reshape(1, 16)
    return ints, in0, in1


def check_string(out0, out1, ints):
    for i in range(16):
        s, d = int(out0[0][i]), int(out1[0][i])
        print(f"{ints[0][i]} + 1 = {s}")
        print(f"{ints[0][i]} - 1 = {d}")
        if s != ints[0][i] + 1:
            fail("error: incorrect sum")
        if d != ints[0][i] - 1:
            fail("error: incorrect difference")

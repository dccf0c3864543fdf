"""Shared body of the four shared-memory examples on ``simple`` /
``simple_string``: region creation, registration, inference with both inputs in
one region (offsets) and both outputs in another, verification, cleanup."""
import numpy as np
from _common import check_simple, check_string, fail, simple_tensors, string_tensors

import tritonclient.utils as utils


def run_simple(mod, client, kind, device=0):
    """``kind`` is "system" or "device" (hipshm)."""
    a, b = simple_tensors()
    nbytes = a.nbytes
    if kind == "system":
        import tritonclient.utils.shared_memory as shm

        client.unregister_system_shared_memory()
        hout = shm.create_shared_memory_region("output_data", "/output_simple", nbytes * 2)
        hin = shm.create_shared_memory_region("input_data", "/input_simple", nbytes * 2)
        shm.set_shared_memory_region(hin, [a])
        shm.set_shared_memory_region(hin, [b], offset=nbytes)
        client.register_system_shared_memory("output_data", "/output_simple", nbytes * 2)
        client.register_system_shared_memory("input_data", "/input_simple", nbytes * 2)
        status = client.get_system_shared_memory_status
        unregister = client.unregister_system_shared_memory
        read = lambda off: shm.get_contents_as_numpy(hout, np.int32, [1, 16], offset=off)  # noqa: E731
    else:
        import tritonclient.utils.cuda_shared_memory as shm  # hipshm on this stack

        client.unregister_cuda_shared_memory()
        hout = shm.create_shared_memory_region("output_data", nbytes * 2, device)
        hin = shm.create_shared_memory_region("input_data", nbytes * 2, device)
        shm.set_shared_memory_region(hin, [a, b])
        client.register_cuda_shared_memory("output_data", shm.get_raw_handle(hout), device, nbytes * 2)
        client.register_cuda_shared_memory("input_data", shm.get_raw_handle(hin), device, nbytes * 2)
        status = client.get_cuda_shared_memory_status
        unregister = client.unregister_cuda_shared_memory
        read = lambda off: shm.get_contents_as_numpy(hout, np.int32, [1, 16], offset=off)  # noqa: E731
    try:
        i0 = mod.InferInput("INPUT0", [1, 16], "INT32")
        i0.set_shared_memory("input_data", nbytes)
        i1 = mod.InferInput("INPUT1", [1, 16], "INT32")
        i1.set_shared_memory("input_data", nbytes, offset=nbytes)
        o0 = mod.InferRequestedOutput("OUTPUT0")
        o0.set_shared_memory("output_data", nbytes)
        o1 = mod.InferRequestedOutput("OUTPUT1")
        o1.set_shared_memory("output_data", nbytes, offset=nbytes)
        results = client.infer("simple", [i0, i1], outputs=[o0, o1])
        if results.get_output("OUTPUT0") is None or results.get_output("OUTPUT1") is None:
            fail("OUTPUT0 / OUTPUT1 is missing in the response")
        check_simple(read(0), read(nbytes), a, b)
        print(status())
    finally:
        unregister()
        shm.destroy_shared_memory_region(hin)
        shm.destroy_shared_memory_region(hout)
    leftovers = shm.mapped_shared_memory_regions() if kind == "system" else shm.allocated_shared_memory_regions()
    if len(leftovers) != 0:
        fail("shared memory regions were not cleaned up")


def run_string(mod, client, kind, device=0):
    ints, in0, in1 = string_tensors()
    s0, s1 = utils.serialize_byte_tensor(in0), utils.serialize_byte_tensor(in1)
    n0, n1 = utils.serialized_byte_size(s0), utils.serialized_byte_size(s1)
    out_bytes = 256  # generous: 16 strings of a few digits each plus prefixes
    if kind == "system":
        import tritonclient.utils.shared_memory as shm

        client.unregister_system_shared_memory()
        hin = shm.create_shared_memory_region("input_data", "/input_simple_str", n0 + n1)
        hout = shm.create_shared_memory_region("output_data", "/output_simple_str", out_bytes * 2)
        shm.set_shared_memory_region(hin, [s0])
        shm.set_shared_memory_region(hin, [s1], offset=n0)
        client.register_system_shared_memory("input_data", "/input_simple_str", n0 + n1)
        client.register_system_shared_memory("output_data", "/output_simple_str", out_bytes * 2)
        unregister = client.unregister_system_shared_memory
    else:
        import tritonclient.utils.cuda_shared_memory as shm

        client.unregister_cuda_shared_memory()
        hin = shm.create_shared_memory_region("input_data", n0 + n1, device)
        hout = shm.create_shared_memory_region("output_data", out_bytes * 2, device)
        shm.set_shared_memory_region(hin, [s0, s1])
        client.register_cuda_shared_memory("input_data", shm.get_raw_handle(hin), device, n0 + n1)
        client.register_cuda_shared_memory("output_data", shm.get_raw_handle(hout), device, out_bytes * 2)
        unregister = client.unregister_cuda_shared_memory
    try:
        i0 = mod.InferInput("INPUT0", [1, 16], "BYTES")
        i0.set_shared_memory("input_data", n0)
        i1 = mod.InferInput("INPUT1", [1, 16], "BYTES")
        i1.set_shared_memory("input_data", n1, offset=n0)
        o0 = mod.InferRequestedOutput("OUTPUT0")
        o0.set_shared_memory("output_data", out_bytes)
        o1 = mod.InferRequestedOutput("OUTPUT1")
        o1.set_shared_memory("output_data", out_bytes, offset=out_bytes)
        client.infer("simple_string", [i0, i1], outputs=[o0, o1])
        out0 = shm.get_contents_as_numpy(hout, np.object_, [1, 16])
        out1 = shm.get_contents_as_numpy(hout, np.object_, [1, 16], offset=out_bytes)
        check_string(out0, out1, ints)
    finally:
        unregister()
        shm.destroy_shared_memory_region(hin)
        shm.destroy_shared_memory_region(hout)

"""Batched device-to-device copies: many (source, destination, size) triples in
one kernel launch (``csrc/hip/batched_copy.hip``).  Used by the dynamic batcher
to gather request inputs into the batch buffer and to scatter output rows into
the clients' shared-memory regions."""

from __future__ import annotations

import ctypes

from ._device import check, lib


def batched_copy(triples, stream=None):
    """``triples``: iterable of ``(src_ptr, dst_ptr, nbytes)`` device addresses.
    ``stream``: torch stream, raw handle or None (current torch stream)."""
    triples = list(triples)
    n = len(triples)
    if n == 0:
        return
    srcs = (ctypes.c_void_p * n)(*[t[0] for t in triples])
    dsts = (ctypes.c_void_p * n)(*[t[1] for t in triples])
    sizes = (ctypes.c_uint64 * n)(*[t[2] for t in triples])
    if stream is None:
        import torch

        handle = torch.cuda.current_stream().cuda_stream
    else:
        handle = getattr(stream, "cuda_stream", stream)
    check(lib().tca_batched_copy_launch(srcs, dsts, sizes, n, handle), "tca_batched_copy_launch")

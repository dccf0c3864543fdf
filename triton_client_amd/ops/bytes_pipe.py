"""BYTES tensors processed on the GPU for whole batches of requests (``csrc/hip/bytes_models.hip``):
prefix walk, the model's arithmetic and the re-packing into the callers' output regions are three
kernel launches per batch; the host only hands over addresses and collects one status word per
request.  The wire form is the reference's (``tritonclient/utils/__init__.py:219-276``).

``DeviceBytes`` describes a BYTES tensor that sits in device memory (a range of a registered device
shared-memory region); ``BytesPipe`` owns the scratch of one batch in flight.
"""

from __future__ import annotations

import ctypes
import threading

from ._device import lib

OP_ADD_SUB = 0
OP_IDENTITY = 1

STATUS_TEXT = {
    1: "malformed BYTES tensor",
    2: "unexpected number of string elements",
    3: "shared memory size specified with the request is too small to hold the results",
    4: "expects decimal integer strings",
}


class DeviceBytes:
    """A BYTES tensor (wire form) in device memory: ``nbytes`` at ``address``.  ``mapping`` /
    ``offset`` name the same bytes as a range of a device shared-memory mapping (for consumers
    that want them on the host)."""

    __slots__ = ("address", "nbytes", "shape", "mapping", "offset", "device")
    dtype = None

    def __init__(self, address, nbytes, shape, mapping=None, offset=0, device=0):
        self.address = int(address)
        self.nbytes = int(nbytes)
        self.shape = tuple(shape)
        self.mapping = mapping
        self.offset = int(offset)
        self.device = int(device)

    def to_host(self):
        """The elements as a numpy object array (device -> host copy + parse)."""
        from ..utils import deserialize_bytes_tensor
        from ..utils import hip_shared_memory as hipshm

        raw = hipshm.read_mapping(self.mapping, self.offset, self.nbytes)
        return deserialize_bytes_tensor(bytes(raw)).reshape(self.shape)


class DeviceBytesResult:
    """Marks an output that the model has already written into its device destination."""

    __slots__ = ("shape", "nbytes")

    def __init__(self, shape, nbytes):
        self.shape = tuple(shape)
        self.nbytes = int(nbytes)


def max_stream_bytes():
    return int(lib().tca_bytes_pipe_max_stream_bytes())


class BytesPipe:
    """Scratch and status of one batch in flight on ``device``."""

    def __init__(self, device=0, max_requests=64, max_count=4096):
        self.device, self.max_requests, self.max_count = int(device), int(max_requests), int(max_count)
        h = ctypes.c_void_p()
        rc = lib().tca_bytes_pipe_create(self.device, self.max_requests, self.max_count, ctypes.byref(h))
        self._h = h
        if rc != 0:
            self.close()
            raise RuntimeError(f"bytes pipe: {lib().tca_bytes_pipe_error().decode()} (code {rc})")
        self.lock = threading.Lock()

    def run(self, op, count, in_ptrs, in_bytes, out_ptrs, out_caps, stream=0):
        """One batch, start to finish: ``len(out_ptrs) / outputs`` requests of ``count`` elements each.
        Returns ``(codes, out_bytes)``: per request the verdict (0 = fine) and the bytes of its outputs."""
        outs = 2 if op == OP_ADD_SUB else 1
        n = len(out_ptrs) // outs
        U64 = ctypes.c_uint64
        a, b = (U64 * len(in_ptrs))(*in_ptrs), (U64 * len(in_bytes))(*in_bytes)
        c, d = (U64 * len(out_ptrs))(*out_ptrs), (U64 * len(out_caps))(*out_caps)
        codes = (ctypes.c_uint32 * n)()
        produced = (U64 * (2 * n))()
        L = lib()
        with self.lock:
            rc = L.tca_bytes_pipe_submit(self._h, op, n, int(count), a, b, c, d, ctypes.c_void_p(stream))
            if rc == 0:
                rc = L.tca_bytes_pipe_wait(self._h, n, codes, produced)
        if rc != 0:
            raise RuntimeError(f"bytes pipe: {L.tca_bytes_pipe_error().decode()} (code {rc})")
        return list(codes), [produced[2 * i: 2 * i + outs] for i in range(n)]

    def close(self):
        h, self._h = self._h, None
        if h:
            lib().tca_bytes_pipe_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchCombiner:
    """Concurrent callers share kernel launches: the first caller to arrive while nothing runs
    becomes the leader and executes batches until the queue is empty; callers that arrive
    meanwhile are picked up by the next batch and sleep until their answer is there."""

    def __init__(self, run_batch, max_batch):
        self._run_batch = run_batch  # list of jobs -> list of results (or exceptions), same order
        self._max = int(max_batch)
        self._cv = threading.Condition()
        self._pending = []
        self._busy = False

    def submit(self, job):
        slot = [job, None, False]  # job, result, done
        with self._cv:
            self._pending.append(slot)
            if self._busy:
                while not slot[2]:
                    self._cv.wait()
                return self._unwrap(slot[1])
            self._busy = True
        while True:
            with self._cv:
                batch, self._pending = self._pending[: self._max], self._pending[self._max:]
                if not batch:
                    self._busy = False
                    break
            try:
                results = self._run_batch([s[0] for s in batch])
            except Exception as e:  # the whole batch failed the same way
                results = [e] * len(batch)
            with self._cv:
                for s, r in zip(batch, results):
                    s[1], s[2] = r, True
                self._cv.notify_all()
        return self._unwrap(slot[1])

    @staticmethod
    def _unwrap(result):
        if isinstance(result, Exception):
            raise result
        return result

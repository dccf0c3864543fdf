"""Bidirectional ``ModelStreamInfer`` plumbing for the synchronous client.

Threads involved: the caller (enqueues requests), grpc's own sender (pulls them
through :class:`_RequestIterator`) and one reader thread that turns responses
into ``callback(result=..., error=...)`` invocations.  Parity:
``tritonclient/grpc/_infer_stream.py:39-191``.
"""

import queue
import threading

import grpc

from ..utils import InferenceServerException, raise_error
from ._infer_result import InferResult
from ._utils import get_cancelled_error, get_error_grpc

_END = object()


class _RequestIterator:
    """Blocking iterator grpc pulls requests from; ``_END`` closes the stream."""

    def __init__(self, pending):
        self._pending = pending

    def __iter__(self):
        return self

    def __next__(self):
        item = self._pending.get()
        if item is _END:
            raise StopIteration
        return item


class _InferStream:
    def __init__(self, callback, verbose):
        self._callback = callback
        self._verbose = verbose
        self._pending = queue.Queue()
        self._responses = None
        self._reader = None
        self._alive = True
        self._lock = threading.Lock()

    def request_iterator(self):
        return _RequestIterator(self._pending)

    def start(self, response_iterator):
        """Begin consuming ``response_iterator`` on the reader thread."""
        self._responses = response_iterator
        self._reader = threading.Thread(target=self._drain, name="tca-grpc-stream", daemon=True)
        self._reader.start()
        if self._verbose:
            print("stream started...")

    def is_active(self):
        with self._lock:
            return self._alive

    def enqueue(self, request):
        """Queue ``request`` for sending; raises once the stream has died."""
        if not self.is_active():
            raise_error("The stream is no longer in valid state, the error detail is reported through provided "
                        "callback. A new stream should be started after stopping the current stream.")
        self._pending.put(request)

    def close(self, cancel_requests=False):
        """Half-close (or cancel) the stream and join the reader."""
        if cancel_requests and self._responses is not None:
            self._responses.cancel()
        self._pending.put(_END)
        if self._reader is not None and self._reader is not threading.current_thread():
            self._reader.join()
            self._reader = None
            if self._verbose:
                print("stream stopped...")

    def _drain(self):
        try:
            for response in self._responses:
                if self._verbose:
                    print(response)
                if response.error_message != "":
                    self._callback(result=None, error=InferenceServerException(msg=response.error_message))
                else:
                    self._callback(result=InferResult(response.infer_response), error=None)
        except grpc.RpcError as rpc_error:
            with self._lock:
                self._alive = False
            if rpc_error.cancelled():
                error = get_cancelled_error()
            else:
                error = get_error_grpc(rpc_error)
            self._callback(result=None, error=error)

"""Client-side request sharding across several server endpoints (one per GPU).

``EndpointPool`` hands out endpoints round-robin and skips the ones marked
dead; ``ShardedClient`` wraps N protocol clients behind the single-client
``infer`` signature, retries a failed request on another endpoint and
periodically probes dead endpoints so that a restarted GPU worker rejoins.
Requests of a stateful sequence (``sequence_id``) always go to the endpoint the
id hashes to - the sequence state lives in that server - and the streaming API
(``start_stream`` / ``async_stream_infer`` / ``stop_stream``, gRPC) keeps one
bidirectional stream per endpoint, so decoupled models and interleaved
sequences scale over the GPUs the same way unary requests do.

Reference hook: the Java client's endpoint provider that never returns the same
endpoint twice in a row during retries
(``src/java/src/main/java/triton/client/endpoint/AbstractEndpoint.java:53-67``).
"""

from __future__ import annotations

import threading
import time

from ..utils import InferenceServerException


class EndpointPool:
    def __init__(self, urls, probe_interval_s=2.0):
        if not urls:
            raise ValueError("at least one endpoint is required")
        self.urls = list(urls)
        self._alive = [True] * len(urls)
        self._dead_since = [0.0] * len(urls)
        self._next = 0
        self._lock = threading.Lock()
        self.probe_interval_s = probe_interval_s

    def acquire(self, exclude=()):
        """Index of the next live endpoint not in ``exclude`` (round robin).  A
        dead endpoint becomes eligible again once its probe interval elapsed."""
        now = time.monotonic()
        with self._lock:
            n = len(self.urls)
            for _ in range(n):
                i = self._next
                self._next = (self._next + 1) % n
                if i in exclude:
                    continue
                if self._alive[i] or now - self._dead_since[i] >= self.probe_interval_s:
                    return i
        return None

    def mark_dead(self, i):
        with self._lock:
            self._alive[i] = False
            self._dead_since[i] = time.monotonic()

    def mark_alive(self, i):
        with self._lock:
            self._alive[i] = True

    def alive(self):
        with self._lock:
            return [u for u, a in zip(self.urls, self._alive) if a]


def _is_transport_failure(err):
    status = err.status() if isinstance(err, InferenceServerException) else None
    if status is None:
        return True  # socket level failure (no HTTP / gRPC status at all)
    return status in ("StatusCode.UNAVAILABLE", "503", "502", "504")


class ShardedClient:
    """Spread requests of one logical client over several endpoints.

    Parameters
    ----------
    urls : list of "host:port"
    protocol : "http" | "grpc"
    retries : number of *other* endpoints tried when one fails at transport level
    client_kwargs : forwarded to each ``InferenceServerClient``
    """

    def __init__(self, urls, protocol="grpc", retries=2, probe_interval_s=2.0, **client_kwargs):
        if protocol == "grpc":
            from .. import grpc as mod
        else:
            from .. import http as mod
        self.mod = mod
        self.pool = EndpointPool(urls, probe_interval_s)
        self.clients = [mod.InferenceServerClient(u, **client_kwargs) for u in urls]
        self.retries = retries
        self.sent = [0] * len(urls)

    def _home(self, sequence_id):
        """Endpoint index that owns a sequence (stable for the life of the client)."""
        import zlib

        return zlib.crc32(str(sequence_id).encode("utf-8")) % len(self.clients)

    def infer(self, model_name, inputs, **kwargs):
        seq = kwargs.get("sequence_id", 0)
        if seq not in (0, "", None):
            # stateful: no failover, the state cannot follow the request
            i = self._home(seq)
            result = self.clients[i].infer(model_name, inputs, **kwargs)
            self.sent[i] += 1
            return result
        tried = []
        last = None
        for _ in range(self.retries + 1):
            i = self.pool.acquire(exclude=tried)
            if i is None:
                break
            tried.append(i)
            try:
                result = self.clients[i].infer(model_name, inputs, **kwargs)
                self.pool.mark_alive(i)
                self.sent[i] += 1
                return result
            except InferenceServerException as e:
                if not _is_transport_failure(e):
                    raise  # the server answered: a retry elsewhere would fail the same way
                self.pool.mark_dead(i)
                last = e
        raise last or InferenceServerException("no live endpoint available")

    # ------------------------------------------------------------- streaming
    def start_stream(self, callback, **kwargs):
        """One bidirectional stream per endpoint; ``callback(result, error)`` receives
        the responses of all of them (from the endpoints' reader threads)."""
        for c in self.clients:
            c.start_stream(callback=callback, **kwargs)
        self._streaming = True

    def async_stream_infer(self, model_name, inputs, **kwargs):
        seq = kwargs.get("sequence_id", 0)
        if seq not in (0, "", None):
            i = self._home(seq)
        else:
            i = self.pool.acquire()
            if i is None:
                raise InferenceServerException("no live endpoint available")
        self.clients[i].async_stream_infer(model_name, inputs, **kwargs)
        self.sent[i] += 1
        return i

    def stop_stream(self, **kwargs):
        for c in self.clients:
            try:
                c.stop_stream(**kwargs)
            except Exception:
                pass
        self._streaming = False

    def live_endpoints(self):
        return self.pool.alive()

    def close(self):
        for c in self.clients:
            try:
                c.close()
            except Exception:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

"""HTTP/REST front on the native HTTP/1.1 engine (``_tca_pyext.HttpFront``).

Connection handling, request parsing (Content-Length and chunked bodies,
``Expect: 100-continue``, keep-alive) and response writing run in C++ on epoll
reactor threads; Python sees one call per complete request and answers with
one call, from whichever thread finishes the request.  A request for a GPU
model behind the dynamic batcher never occupies a thread while it waits: the
reactor callback parses and queues it, the batcher's completer thread renders
and sends the answer.  Microsecond-scale CPU models run inline on the reactor
callback; everything else (control endpoints that may block, slow models) on a
small worker pool.

Same REST surface as :mod:`.http_front` - both use its ``Router`` - selected
with ``Server(http_impl="native")``.  Request bodies of 128 KiB and more stay
in the C++ buffer and are aliased by the tensors, not copied.
"""

from __future__ import annotations

import os
from concurrent import futures

from .core import ASYNC, ServerError
from .http_front import Router, _dumps


class NativeHttpFront:
    def __init__(self, core, host="127.0.0.1", port=8000, workers=16, reactors=2, tls=None, native=None):
        from .._native import pyext

        self.core = core
        self.router = Router(core)
        self.host = host
        self.port = port
        self._pool = futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tca-nhttp")
        # stored plans for repeated request bodies (TCA_REQUEST_PLANS=0: parse and validate every request)
        self._plans = os.environ.get("TCA_REQUEST_PLANS", "1") != "0"
        kwargs = {}
        if tls:  # (certificate chain, private key, client CA or None) as PEM file paths
            kwargs = {"tls_cert": tls[0], "tls_key": tls[1], "tls_client_ca": tls[2]}
        if native is not None:  # NativeExecutor: ordinary requests for its models never reach _on_request
            kwargs["native"] = native
        self._front = pyext().HttpFront(host, int(port), int(reactors), self._on_request, None, **kwargs)

    def start(self):
        self.port = self._front.start()
        return self.port

    def stop(self, grace=0.5):
        self._front.stop()
        self._pool.shutdown(wait=False)

    def stats(self):
        return self._front.stats()

    # ---------------------------------------------------------------- requests
    def _fail(self, call, error):
        if isinstance(error, ServerError):
            self._front.respond(call, error.code, None, [_dumps({"error": error.message})])
        else:
            self._front.respond(call, 500, None,
                                [_dumps({"error": f"internal error: {type(error).__name__}: {error}"})])

    def _on_request(self, call, method, target, headers, body):
        try:
            routed = self.router.route(method, target, headers, body)
            kind = routed[0]
            if kind == "infer":
                _, model, version, body = routed
                plan_key = None
                if (self._plans and len(body) <= 4096 and "inference-header-content-length" not in headers
                        and "content-encoding" not in headers):
                    # a small, plain JSON body (tensors by reference): a repeat is queued from its stored plan
                    plan_key = (model, version, bytes(body))
                    if self.core.try_planned(plan_key, on_done_for=lambda req: self._done_callback(call, req, headers)):
                        return
                req = self.router.parse_infer(model, version, headers, body)
                if self.core.uses_batcher(model) or self.core.runs_inline(model):
                    # queues and returns / answers within microseconds
                    self._infer(call, req, headers, plan_key)
                else:
                    self._pool.submit(self._infer, call, req, headers)
            elif kind == "blocking":
                self._pool.submit(self._blocking, call, routed[1])
            else:
                self._front.respond(call, routed[0], routed[1], routed[2])
        except Exception as e:  # noqa: BLE001 - every request gets an answer
            self._fail(call, e)

    def _blocking(self, call, fn):
        try:
            status, rh, parts = fn()
            self._front.respond(call, status, rh, parts)
        except Exception as e:  # noqa: BLE001
            self._fail(call, e)

    def _done_callback(self, call, req, headers):
        def done(resp, err):
            if err is not None:
                self._fail(call, err if isinstance(err, Exception) else ServerError(str(err), 500))
                return
            try:
                status, rh, parts = Router.render_infer(resp, req, headers)
                self._front.respond(call, status, rh, parts)
            except Exception as e:  # noqa: BLE001
                self._fail(call, e)

        return done

    def _infer(self, call, req, headers, plan_key=None):
        done = self._done_callback(call, req, headers)
        try:
            result = self.core.infer(req, on_done=done, plan_key=plan_key)
            if result is not ASYNC:
                done(result, None)
        except Exception as e:  # noqa: BLE001
            self._fail(call, e)

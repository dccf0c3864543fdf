"""gRPC front on the native HTTP/2 engine (``_tca_pyext.GrpcFront``).

The wire work (HTTP/2 framing, HPACK, flow control, gRPC message framing)
runs in C++ on epoll reactor threads; Python only sees complete request
messages and hands back complete response messages.  Requests for GPU models
behind the dynamic batcher never occupy a thread while they wait for the GPU:
the reactor callback parses and queues the request, and the batcher's
completer thread serialises and sends the answer.  Everything else (control
RPCs, CPU models, streams) runs on a small worker pool so a slow model cannot
stall the reactors.

Same service surface as :mod:`.grpc_front` (it reuses that module's servicer
for every control RPC and its protobuf conversions), same client-visible
behaviour; selected with ``Server(grpc_impl="native")``.
"""

from __future__ import annotations

import collections
import os
import threading
from concurrent import futures

from ..grpc import service_pb2 as pb
from .core import ASYNC, ServerError
from .grpc_front import Servicer, mark_final, request_from_proto, response_to_proto

# gRPC status codes
_OK, _INVALID_ARGUMENT, _NOT_FOUND, _UNIMPLEMENTED, _INTERNAL, _UNAVAILABLE = 0, 3, 5, 12, 13, 14
_STATUS = {400: _INVALID_ARGUMENT, 404: _NOT_FOUND, 503: _UNAVAILABLE, 500: _INTERNAL}


class _Abort(Exception):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.message = message


class _Context:
    """The slice of ``grpc.ServicerContext`` the servicer uses."""

    def abort(self, code, details):
        raise _Abort(code.value[0], details)


class _Stream:
    __slots__ = ("lock", "pending", "ended", "ordered", "draining", "closed", "written")

    def __init__(self):
        self.lock = threading.Lock()
        self.pending = 0        # requests whose responses are still owed
        self.ended = False      # client half-closed
        self.ordered = collections.deque()  # sequence requests: executed one at a time, in arrival order
        self.draining = False
        self.closed = False
        self.written = {}       # model name -> responses written (fault injection only)


class NativeGrpcFront:
    def __init__(self, core, host="127.0.0.1", port=8001, workers=16, reactors=2, tls=None, native=None):
        from .._native import pyext

        self.core = core
        self.host = host
        self.port = port
        self._servicer = Servicer(core, stream_workers=1)
        self._pool = futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tca-ngrpc")
        self._streams = {}
        self._streams_lock = threading.Lock()
        # stored plans for repeated request messages (TCA_REQUEST_PLANS=0: parse and validate every request)
        self._plans = os.environ.get("TCA_REQUEST_PLANS", "1") != "0"
        self._methods = {}
        svc = pb.DESCRIPTOR.services_by_name["GRPCInferenceService"]
        for m in svc.methods:
            if m.name in ("ModelInfer", "ModelStreamInfer"):
                continue
            self._methods[m.name] = (getattr(pb, m.input_type.name), getattr(self._servicer, m.name))
        on_unary = self._on_unary
        self._prof = None
        if os.environ.get("TCA_PROFILE_DIR"):  # developer aid: profile the reactor-side Python work
            import cProfile

            self._prof = cProfile.Profile()
            lock = threading.Lock()

            def on_unary(call, method, data, blob=None, spans=None, _inner=self._on_unary):
                with lock:
                    self._prof.runcall(_inner, call, method, data, blob, spans)

        kwargs = {}
        if tls:  # (certificate chain, private key, client CA or None) as PEM file paths
            kwargs = {"tls_cert": tls[0], "tls_key": tls[1], "tls_client_ca": tls[2]}
        if native is not None:  # NativeExecutor: ordinary requests for its models never reach on_unary
            kwargs["native"] = native
        self._front = pyext().GrpcFront(host, int(port), int(reactors), on_unary, self._on_stream_message,
                                        self._on_stream_end, self._on_cancel, **kwargs)

    def start(self):
        self.port = self._front.start()
        return self.port

    def stop(self, grace=0.5):
        self._front.stop()
        self._pool.shutdown(wait=False)
        if self._prof is not None:
            out = os.environ["TCA_PROFILE_DIR"]
            os.makedirs(out, exist_ok=True)
            self._prof.dump_stats(os.path.join(out, "reactor.prof"))

    def stats(self):
        return self._front.stats()

    # ------------------------------------------------------------------ unary
    def _on_unary(self, call, method, data, blob=None, spans=None):
        if method == "ModelInfer":
            if blob is None and self._plans and self.core.try_planned(data, self._done_callback(call)):
                return  # a repeat of a message seen before: queued from its stored plan
            try:
                raw = None
                if blob is not None:
                    # large request: `data` is the message without its tensors, which stay in
                    # the C++ buffer and are aliased, not copied
                    whole = memoryview(blob)
                    raw = [whole[off: off + size] for off, size in spans]
                req = request_from_proto(pb.ModelInferRequest.FromString(data), raw=raw)
            except ServerError as e:
                self._front.finish(call, _STATUS.get(e.code, _INTERNAL), e.message)
                return
            except Exception as e:
                self._front.finish(call, _INVALID_ARGUMENT, f"malformed ModelInferRequest: {e}")
                return
            if self.core.uses_batcher(req.model_name) or self.core.runs_inline(req.model_name):
                # queues and returns / answers within microseconds
                self._infer(call, req, data if blob is None and self._plans else None)
            else:
                self._pool.submit(self._infer, call, req)
            return
        entry = self._methods.get(method)
        if entry is None:
            self._front.finish(call, _UNIMPLEMENTED, f"Method not found: {method}")
            return
        self._pool.submit(self._control, call, entry, data)

    def _control(self, call, entry, data):
        request_cls, handler = entry
        try:
            response = handler(request_cls.FromString(data), _Context())
            self._front.reply(call, response.SerializeToString())
        except _Abort as e:
            self._front.finish(call, e.status, e.message)
        except ServerError as e:
            self._front.finish(call, _STATUS.get(e.code, _INTERNAL), e.message)
        except Exception as e:
            self._front.finish(call, _INTERNAL, f"internal error: {type(e).__name__}: {e}")

    def _done_callback(self, call):
        def done(resp, err):
            if err is not None:
                self._front.finish(call, _STATUS.get(getattr(err, "code", 500), _INTERNAL),
                                   str(getattr(err, "message", err)))
            else:
                self._front.reply(call, response_to_proto(resp).SerializeToString())

        return done

    def _infer(self, call, req, plan_key=None):
        done = self._done_callback(call)
        try:
            result = self.core.infer(req, on_done=done, plan_key=plan_key)
            if result is ASYNC:
                return
            if not hasattr(result, "outputs"):
                raise ServerError("ModelInfer RPC doesn't support models with decoupled transaction policy", 400)
            done(result, None)
        except ServerError as e:
            done(None, e)
        except Exception as e:
            done(None, ServerError(f"internal error: {type(e).__name__}: {e}", 500))

    # ---------------------------------------------------------------- streams
    def _stream(self, call, create=True):
        with self._streams_lock:
            st = self._streams.get(call)
            if st is None and create:
                st = self._streams[call] = _Stream()
            return st

    def _on_stream_message(self, call, method, data):
        st = self._stream(call)
        try:
            msg = pb.ModelInferRequest.FromString(data)
        except Exception as e:
            self._front.send(call, pb.ModelStreamInferResponse(
                error_message=f"malformed ModelInferRequest: {e}").SerializeToString())
            return
        with st.lock:
            st.pending += 1
        if "sequence_id" in msg.parameters:
            with st.lock:
                st.ordered.append(msg)
                start = not st.draining
                st.draining = True
            if start:
                self._pool.submit(self._drain_ordered, call, st)
            return
        try:
            req = request_from_proto(msg, streaming=True)
        except ServerError as e:
            self._stream_error(call, st, msg, e.message)
            return
        if self.core.uses_batcher(req.model_name):
            self._stream_infer(call, st, msg, req)
        else:
            self._pool.submit(self._stream_infer, call, st, msg, req)

    def _drain_ordered(self, call, st):
        while True:
            with st.lock:
                if not st.ordered:
                    st.draining = False
                    return
                msg = st.ordered.popleft()
            try:
                req = request_from_proto(msg, streaming=True)
            except ServerError as e:
                self._stream_error(call, st, msg, e.message)
                continue
            self._stream_infer(call, st, msg, req)

    def _stream_error(self, call, st, msg, text):
        err = pb.ModelStreamInferResponse(error_message=text)
        err.infer_response.id = msg.id
        err.infer_response.model_name = msg.model_name
        self._front.send(call, err.SerializeToString())
        self._stream_done(call, st)

    def _stream_send(self, call, st, model_name, message):
        """Write one stream response; applies the ``drop_stream_after_n`` fault knob."""
        self._front.send(call, message.SerializeToString())
        limit = self.core.stream_drop_after(model_name)
        if not limit:
            return
        with st.lock:
            st.written[model_name] = st.written.get(model_name, 0) + 1
            drop = st.written[model_name] >= limit and not st.closed
            if drop:
                st.closed = True
                st.ordered.clear()
        if drop:
            self._front.finish(call, _UNAVAILABLE,
                               f"injected stream failure for model '{model_name}' (drop_stream_after_n={limit})")
            with self._streams_lock:
                self._streams.pop(call, None)

    def _stream_done(self, call, st):
        with st.lock:
            st.pending -= 1
            finish = st.ended and st.pending == 0 and not st.closed
            if finish:
                st.closed = True
        if finish:
            self._front.finish(call, _OK)
            with self._streams_lock:
                self._streams.pop(call, None)

    def _stream_infer(self, call, st, msg, req):
        def done(resp, err):
            if err is not None:
                self._stream_error(call, st, msg, str(getattr(err, "message", err)))
                return
            mark_final(req, resp)
            self._stream_send(call, st, req.model_name, pb.ModelStreamInferResponse(
                infer_response=response_to_proto(resp)))
            self._stream_done(call, st)

        try:
            result = self.core.infer(req, on_done=done)
            if result is ASYNC:
                return
            if hasattr(result, "outputs"):
                done(result, None)
                return
            for item in result:  # decoupled model: any number of responses
                if st.closed:
                    break
                self._stream_send(call, st, req.model_name, pb.ModelStreamInferResponse(
                    infer_response=response_to_proto(item)))
            self._stream_done(call, st)
        except ServerError as e:
            self._stream_error(call, st, msg, e.message)
        except Exception as e:
            self._stream_error(call, st, msg, f"internal error: {type(e).__name__}: {e}")

    def _on_stream_end(self, call):
        st = self._stream(call)
        with st.lock:
            st.ended = True
            finish = st.pending == 0 and not st.closed
            if finish:
                st.closed = True
        if finish:
            self._front.finish(call, _OK)
            with self._streams_lock:
                self._streams.pop(call, None)

    def _on_cancel(self, call):
        with self._streams_lock:
            st = self._streams.pop(call, None)
        if st is not None:
            with st.lock:
                st.closed = True
                st.ordered.clear()

"""HTTP/REST front of the in-repo v2 server (asyncio, HTTP/1.1 keep-alive).

Implements the endpoint table the clients use (reference client calls:
``tritonclient/http/_client.py:364-1484``): health, metadata, config,
repository index/load/unload, statistics, trace and log settings, system and
device shared memory, and ``infer`` with the binary tensor extension
(``Inference-Header-Content-Length``) and gzip/deflate in both directions.

One ``asyncio.Protocol`` instance per connection parses requests
incrementally; inference is executed on a thread pool so the event loop keeps
accepting while models run.
"""

from __future__ import annotations

import asyncio
import gzip
import json
import re
import zlib
from concurrent.futures import ThreadPoolExecutor
from urllib.parse import unquote

import numpy as np

from .core import ASYNC, InferRequest, InputTensor, RequestedOutput, ServerError, ShmRef, encode_wire

_REASONS = {200: "OK", 400: "Bad Request", 404: "Not Found", 405: "Method Not Allowed", 500: "Internal Server Error",
            503: "Service Unavailable"}

_MODEL_RE = re.compile(r"^v2/models/(?P<model>[^/]+)(?:/versions/(?P<version>[^/]+))?(?:/(?P<action>.+))?$")
_SHM_RE = re.compile(r"^v2/(?P<kind>systemsharedmemory|cudasharedmemory)(?:/region/(?P<region>[^/]+))?/"
                     r"(?P<action>status|register|unregister)$")
_REPO_RE = re.compile(r"^v2/repository/models/(?P<model>[^/]+)/(?P<action>load|unload)$")


def _json_default(o):
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.bool_):
        return bool(o)
    if isinstance(o, bytes):
        return o.decode("utf-8")
    raise TypeError(type(o))


def _dumps(obj):
    return json.dumps(obj, separators=(",", ":"), default=_json_default).encode("utf-8")


def parse_infer_body(model, version, body, header_length):
    """HTTP infer body -> :class:`InferRequest`."""
    view = memoryview(body)
    if header_length is None:
        head, tail = view, memoryview(b"")
    else:
        if header_length < 0 or header_length > len(view):
            raise ServerError("Inference-Header-Content-Length is larger than the request body", 400)
        head, tail = view[:header_length], view[header_length:]
    try:
        doc = json.loads(bytes(head))
    except ValueError as e:
        raise ServerError(f"failed to parse the request JSON buffer: {e}", 400) from None
    if not isinstance(doc, dict) or "inputs" not in doc:
        raise ServerError("inference request must contain an 'inputs' array", 400)
    req = InferRequest(model_name=model, model_version=version or "", id=doc.get("id", ""),
                       parameters=dict(doc.get("parameters") or {}))
    cursor = 0
    for item in doc["inputs"]:
        try:
            t = InputTensor(name=item["name"], datatype=item["datatype"], shape=list(item["shape"]))
        except (KeyError, TypeError):
            raise ServerError("each input needs 'name', 'datatype' and 'shape'", 400) from None
        params = item.get("parameters") or {}
        if "shared_memory_region" in params:
            t.shm = ShmRef(params["shared_memory_region"], int(params.get("shared_memory_byte_size", 0)),
                           int(params.get("shared_memory_offset", 0)))
        elif "binary_data_size" in params:
            size = int(params["binary_data_size"])
            if size < 0:
                raise ServerError(f"negative binary_data_size for input '{t.name}'", 400)
            if cursor + size > len(tail):
                raise ServerError(f"unexpected end of binary data for input '{t.name}'", 400)
            t.raw = tail[cursor: cursor + size]
            cursor += size
        elif "data" in item:
            t.json_data = item["data"]
        else:
            raise ServerError(f"input '{t.name}' has neither 'data', binary data nor a shared memory region", 400)
        req.inputs.append(t)
    if cursor != len(tail):
        raise ServerError(
            f"unexpected additional input data for model '{model}' ({len(tail) - cursor} trailing bytes)", 400)
    default_binary = bool(req.parameters.pop("binary_data_output", False))
    outs = doc.get("outputs")
    if outs:
        req.outputs = []
        for item in outs:
            params = item.get("parameters") or {}
            ro = RequestedOutput(name=item["name"], binary=bool(params.get("binary_data", default_binary)),
                                 class_count=int(params.get("classification", 0)))
            if "shared_memory_region" in params:
                ro.shm = ShmRef(params["shared_memory_region"], int(params.get("shared_memory_byte_size", 0)),
                                int(params.get("shared_memory_offset", 0)))
            req.outputs.append(ro)
    req.http_default_binary = default_binary
    return req


def render_infer_response(resp, default_binary):
    """:class:`InferResponse` -> ``(body_parts, json_length | None)``."""
    doc = {"model_name": resp.model_name, "model_version": resp.model_version}
    if resp.id:
        doc["id"] = resp.id
    if resp.parameters:
        doc["parameters"] = resp.parameters
    blobs = []
    outputs = []
    for out in resp.outputs:
        entry = {"name": out.name, "datatype": out.datatype, "shape": [int(s) for s in out.shape]}
        if out.in_shm is not None:
            params = {"shared_memory_region": out.in_shm.region, "shared_memory_byte_size": out.in_shm.byte_size}
            if out.in_shm.offset:
                params["shared_memory_offset"] = out.in_shm.offset
            entry["parameters"] = params
        else:
            binary = default_binary if out.binary is None else out.binary
            if binary:
                blob = encode_wire(out.datatype, out.data)
                entry["parameters"] = {"binary_data_size": len(blob)}
                blobs.append(blob)
            elif out.datatype == "BYTES":
                try:
                    entry["data"] = [v.decode("utf-8") if isinstance(v, bytes) else str(v)
                                     for v in out.data.reshape(-1).tolist()]
                except UnicodeDecodeError:
                    raise ServerError(
                        f"output '{out.name}' holds non UTF-8 bytes and cannot be returned as JSON; request "
                        "binary_data=true", 400) from None
            elif out.datatype == "BF16":
                raise ServerError("BF16 outputs must be requested as binary data", 400)
            else:
                entry["data"] = out.data.reshape(-1).tolist()
        outputs.append(entry)
    doc["outputs"] = outputs
    head = _dumps(doc)
    if blobs:
        return [head, *blobs], len(head)
    return [head], None


class Router:
    """The REST surface without the transport: maps one request to its answer.

    ``route`` returns ``(status, headers, parts)`` for endpoints answered on the spot,
    ``("blocking", fn)`` for the ones that may take long (model load / unload; ``fn()``
    gives the answer) and ``("infer", model, version, body)`` for an inference, which the
    front runs its own way (thread pool, batcher callback, inline)."""

    def __init__(self, core):
        self.core = core

    def route(self, method, target, headers, body):
        path, _, _query = target.partition("?")
        path = (unquote(path) if "%" in path else path).strip("/")
        core = self.core
        encoding = headers.get("content-encoding")
        if encoding == "gzip":
            body = gzip.decompress(body)
        elif encoding == "deflate":
            body = zlib.decompress(body)

        def ok(obj=None):
            return 200, {}, [_dumps(obj) if obj is not None else b""]

        m = _MODEL_RE.match(path) if path != "v2/models/stats" else None
        if m:
            model, version, action = m["model"], m["version"] or "", m["action"]
            if action == "infer":
                if method != "POST":
                    raise ServerError("infer requires POST", 405)
                return "infer", model, version, body
            if action is None:
                return ok(core.model_metadata(model, version))
            if action == "ready":
                return (200 if core.is_model_ready(model, version) else 400), {}, [b""]
            if action == "config":
                return ok(core.model_config(model, version))
            if action == "stats":
                return ok(core.statistics(model, version))
            if action == "trace/setting":
                if method == "POST":
                    return ok(core.update_trace(model, json.loads(bytes(body)) if bytes(body).strip() else {}))
                return ok(core.get_trace(model))
            raise ServerError(f"no handler for {method} /{path}", 404)
        body = bytes(body)
        if path == "v2/health/live":
            return (200 if core.live else 400), {}, [b""]
        if path == "v2/health/ready":
            return (200 if core.ready else 400), {}, [b""]
        if path == "v2":
            return ok(core.server_metadata())
        if path == "v2/models/stats":
            return ok(core.statistics())
        if path == "metrics":
            return 200, {"Content-Type": "text/plain; version=0.0.4; charset=utf-8"}, [core.metrics_text().encode()]
        if path == "v2/repository/index":
            ready = bool(json.loads(body).get("ready", False)) if body.strip() else False
            return ok(core.repository_index(ready))
        if path == "v2/trace/setting":
            if method == "POST":
                return ok(core.update_trace(None, json.loads(body) if body.strip() else {}))
            return ok(core.get_trace(None))
        if path == "v2/logging":
            if method == "POST":
                return ok(core.update_log(json.loads(body) if body.strip() else {}))
            return ok(dict(core.log_settings))
        m = _REPO_RE.match(path)
        if m:
            doc = json.loads(body) if body.strip() else {}
            params = doc.get("parameters") or {}
            if m["action"] == "load":
                files = {k: v for k, v in params.items() if k.startswith("file:")}

                def run(name=m["model"]):
                    core.load_model(name, params.get("config"), files or None)
                    return ok()
            else:
                def run(name=m["model"]):
                    core.unload_model(name, bool(params.get("unload_dependents", False)))
                    return ok()
            return "blocking", run
        m = _SHM_RE.match(path)
        if m:
            return self._shm(m["kind"], m["region"], m["action"], body, ok)
        raise ServerError(f"no handler for {method} /{path}", 404)

    def _shm(self, kind, region, action, body, ok):
        core = self.core
        reg = core.sysshm if kind == "systemsharedmemory" else core.devshm
        if action == "status":
            return ok(reg.status(region or ""))
        if action == "unregister":
            reg.unregister(region or "")
            return ok()
        if not region:
            raise ServerError("register requires a region name", 400)
        doc = json.loads(body) if body.strip() else {}
        try:
            if kind == "systemsharedmemory":
                reg.register(region, doc["key"], int(doc.get("offset", 0)), int(doc["byte_size"]))
            else:
                import base64

                raw = base64.b64decode(doc["raw_handle"]["b64"])
                reg.register(region, raw, int(doc["device_id"]), int(doc["byte_size"]))
        except KeyError as e:
            raise ServerError(f"missing field {e} in shared memory register request", 400) from None
        return ok()

    def parse_infer(self, model, version, headers, body):
        hl = headers.get("inference-header-content-length")
        try:
            header_length = int(hl) if hl is not None else None
        except ValueError:
            raise ServerError("malformed Inference-Header-Content-Length", 400) from None
        return parse_infer_body(model, version, body, header_length)

    @staticmethod
    def render_infer(resp, req, headers):
        """Response of an inference -> ``(200, headers, parts)`` (binary extension, compression)."""
        if not hasattr(resp, "outputs"):
            raise ServerError(
                f"model '{req.model_name}' is decoupled and can only be used through the gRPC streaming API", 400)
        parts, json_len = render_infer_response(resp, getattr(req, "http_default_binary", False))
        rh = {}
        if json_len is not None:
            rh["Inference-Header-Content-Length"] = str(json_len)
            rh["Content-Type"] = "application/octet-stream"
        accept = headers.get("accept-encoding", "")
        if "gzip" in accept:
            parts = [gzip.compress(b"".join(parts), compresslevel=1)]
            rh["Content-Encoding"] = "gzip"
        elif "deflate" in accept:
            parts = [zlib.compress(b"".join(parts), 1)]
            rh["Content-Encoding"] = "deflate"
        return 200, rh, parts


class _HttpProtocol(asyncio.Protocol):
    def __init__(self, front):
        self.front = front
        self.transport = None
        self.buf = bytearray()
        self.busy = False
        self.closing = False

    def connection_made(self, transport):
        self.transport = transport
        sock = transport.get_extra_info("socket")
        if sock is not None:
            import socket as _s

            try:
                sock.setsockopt(_s.IPPROTO_TCP, _s.TCP_NODELAY, 1)
            except OSError:
                pass

    def connection_lost(self, exc):
        self.closing = True

    def data_received(self, data):
        self.buf += data
        if not self.busy:
            self._pump()

    def _pump(self):
        while not self.busy and not self.closing:
            end = self.buf.find(b"\r\n\r\n")
            if end < 0:
                if len(self.buf) > (1 << 20):
                    self._send(400, {}, [_dumps({"error": "request header section too large"})], close=True)
                return
            head = bytes(self.buf[:end]).split(b"\r\n")
            try:
                method, target, _ = head[0].decode("latin-1").split(" ", 2)
            except ValueError:
                self._send(400, {}, [_dumps({"error": "malformed request line"})], close=True)
                return
            headers = {}
            for line in head[1:]:
                k, sep, v = line.partition(b":")
                if sep:
                    headers[k.strip().lower().decode("latin-1")] = v.strip().decode("latin-1")
            length = int(headers.get("content-length", "0") or 0)
            total = end + 4 + length
            if len(self.buf) < total:
                if headers.get("expect", "").lower() == "100-continue" and not getattr(self, "_continued", False):
                    self._continued = True
                    self.transport.write(b"HTTP/1.1 100 Continue\r\n\r\n")
                return
            self._continued = False
            body = bytes(self.buf[end + 4: total])
            del self.buf[:total]
            self.busy = True
            task = asyncio.ensure_future(self._handle(method, target, headers, body))
            self.front.tasks.add(task)
            task.add_done_callback(self.front.tasks.discard)

    async def _handle(self, method, target, headers, body):
        keep = headers.get("connection", "").lower() != "close"
        try:
            status, rh, parts = await self.front.dispatch(method, target, headers, body)
        except ServerError as e:
            status, rh, parts = e.code, {}, [_dumps({"error": e.message})]
        except Exception as e:  # defensive: never kill the connection silently
            status, rh, parts = 500, {}, [_dumps({"error": f"internal error: {type(e).__name__}: {e}"})]
        self._send(status, rh, parts, close=not keep)
        self.busy = False
        if self.buf:
            self._pump()

    def _send(self, status, rh, parts, close=False):
        if self.closing:
            return
        total = sum(len(p) for p in parts)
        lines = [f"HTTP/1.1 {status} {_REASONS.get(status, 'Error')}", f"Content-Length: {total}"]
        if "Content-Type" not in rh:
            lines.append("Content-Type: application/json")
        for k, v in rh.items():
            lines.append(f"{k}: {v}")
        if close:
            lines.append("Connection: close")
        head = ("\r\n".join(lines) + "\r\n\r\n").encode("latin-1")
        if total <= 65536:
            # one write = one send() and one segment train: a small response written as
            # header + body costs two system calls and wakes the client twice
            self.transport.write(b"".join([head, *parts]))
        else:
            self.transport.write(head)
            for p in parts:
                if len(p):
                    self.transport.write(p)  # large tensors: no extra copy
        if close:
            self.transport.close()
            self.closing = True


class HttpFront:
    def __init__(self, core, host="127.0.0.1", port=8000, workers=16, tls=None):
        self.core = core
        self.router = Router(core)
        self.host = host
        self.port = port
        self._tls = tls  # (certificate chain, private key, client CA or None) as PEM file paths
        self.tasks = set()  # requests being handled: drained (then cancelled) at shutdown
        self.pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tca-http-srv")
        self.server = None

    async def start(self):
        loop = asyncio.get_running_loop()
        context = None
        if self._tls:
            import ssl

            cert, key, client_ca = self._tls
            context = ssl.SSLContext(ssl.PROTOCOL_TLS_SERVER)
            context.load_cert_chain(cert, key)
            if client_ca:
                context.load_verify_locations(client_ca)
                context.verify_mode = ssl.CERT_REQUIRED
        self.server = await loop.create_server(lambda: _HttpProtocol(self), self.host, self.port, reuse_address=True,
                                               backlog=1024, ssl=context)
        self.port = self.server.sockets[0].getsockname()[1]
        return self.port

    async def stop(self, grace=1.0):
        if self.server is not None:
            self.server.close()
            await self.server.wait_closed()
        pending = [t for t in self.tasks if not t.done()]
        if pending:  # let requests in flight finish, then drop what is still stuck
            _, late = await asyncio.wait(pending, timeout=grace)
            for t in late:
                t.cancel()
            if late:
                await asyncio.gather(*late, return_exceptions=True)
        self.pool.shutdown(wait=False)

    # ------------------------------------------------------------- routing
    async def dispatch(self, method, target, headers, body):
        routed = self.router.route(method, target, headers, body)
        if routed[0] == "infer":
            return await self._infer(routed[1], routed[2], headers, routed[3])
        if routed[0] == "blocking":
            return await asyncio.get_running_loop().run_in_executor(self.pool, routed[1])
        return routed

    async def _infer(self, model, version, headers, body):
        req = self.router.parse_infer(model, version, headers, body)
        loop = asyncio.get_running_loop()
        if self.core.uses_batcher(model):
            # GPU model behind the dynamic batcher: queue from this thread, the completer
            # thread hands the outcome back to the loop - no pool thread parked per request
            fut = loop.create_future()

            def settle(response, error):
                if fut.done():
                    return
                if error is not None:
                    fut.set_exception(error)
                else:
                    fut.set_result(response)

            def done(response, error):
                loop.call_soon_threadsafe(settle, response, error)

            resp = self.core.infer(req, on_done=done)
            if resp is ASYNC:
                resp = await fut
        elif self.core.runs_inline(model):
            # microsecond-scale CPU model without delay / fault knobs: two thread hand-offs
            # would cost several times the inference
            resp = self.core.infer(req)
        else:
            resp = await loop.run_in_executor(self.pool, self.core.infer, req)
        return Router.render_infer(resp, req, headers)

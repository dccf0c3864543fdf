"""Blocking HTTP/1.1 transport with a keep-alive connection pool.

The reference rides on ``geventhttpclient`` (``http/_client.py:182-191``), which
is not available here, and its ``async_infer`` costs a 10 ms cooperative sleep
per submit.  This transport is a small purpose-built client on plain sockets:

* ``concurrency`` persistent connections, handed out through a LIFO queue;
* request bodies are sent as a gather list (``socket.sendmsg``) – JSON header
  and tensor buffers are never concatenated;
* responses are read into one pre-sized ``bytearray`` (``Content-Length``) or
  de-chunked (``Transfer-Encoding: chunked``);
* ``TCP_NODELAY`` and 4 MiB socket buffers for large tensors.
"""

from __future__ import annotations

import queue
import socket
import ssl as _ssl
import threading

from ..utils import InferenceServerException

_CRLF = b"\r\n"
_MAX_HEADER = 1 << 20


class HttpResponse:
    """Minimal response object: ``status_code``, ``get(header)``, ``read()``."""

    __slots__ = ("status_code", "headers", "_body", "_pos", "reason")

    def __init__(self, status_code, headers, body, reason=""):
        self.status_code = status_code
        self.headers = headers  # lower-cased names
        self._body = body
        self._pos = 0
        self.reason = reason

    def get(self, name, default=None):
        return self.headers.get(name.lower(), default)

    def read(self, length=-1):
        view = memoryview(self._body)
        if length is None or length < 0:
            out = view[self._pos :]
            self._pos = len(view)
        else:
            out = view[self._pos : self._pos + length]
            self._pos += len(out)
        return out

    def __repr__(self):
        return f"<HttpResponse {self.status_code} {self.reason} headers={self.headers} body={len(self._body)}B>"


class _Connection:
    def __init__(self, host, port, connect_timeout, io_timeout, ssl_context, server_hostname):
        self._args = (host, port, connect_timeout, io_timeout, ssl_context, server_hostname)
        self.sock = None
        self._rbuf = bytearray()

    def _open(self):
        host, port, connect_timeout, io_timeout, ctx, server_hostname = self._args
        sock = socket.create_connection((host, port), timeout=connect_timeout)
        try:
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            sock.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, 4 << 20)
            sock.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4 << 20)
        except OSError:
            pass
        if ctx is not None:
            sock = ctx.wrap_socket(sock, server_hostname=server_hostname)
        sock.settimeout(io_timeout)
        self.sock = sock
        self._rbuf = bytearray()

    def close(self):
        if self.sock is not None:
            try:
                self.sock.close()
            except OSError:
                pass
            self.sock = None

    # -- send --
    def _send_all(self, parts):
        sock = self.sock
        if isinstance(sock, _ssl.SSLSocket) or not hasattr(sock, "sendmsg"):
            for p in parts:
                sock.sendall(p)
            return
        views = [memoryview(p).cast("B") for p in parts if len(p)]
        while views:
            sent = sock.sendmsg(views[:64])
            while sent and views:
                head = views[0]
                if sent >= head.nbytes:
                    sent -= head.nbytes
                    views.pop(0)
                else:
                    views[0] = head[sent:]
                    sent = 0

    # -- receive --
    def _fill(self):
        chunk = self.sock.recv(262144)
        if not chunk:
            raise ConnectionError("connection closed by peer")
        self._rbuf += chunk

    def _read_headers(self):
        while True:
            end = self._rbuf.find(b"\r\n\r\n")
            if end >= 0:
                break
            if len(self._rbuf) > _MAX_HEADER:
                raise ConnectionError("response header section too large")
            self._fill()
        head = bytes(self._rbuf[:end])
        del self._rbuf[: end + 4]
        lines = head.split(_CRLF)
        status_line = lines[0].split(None, 2)
        if len(status_line) < 2 or not status_line[0].startswith(b"HTTP/"):
            raise ConnectionError("malformed HTTP status line")
        status = int(status_line[1])
        reason = status_line[2].decode("latin-1") if len(status_line) > 2 else ""
        headers = {}
        for line in lines[1:]:
            name, sep, value = line.partition(b":")
            if sep:
                headers[name.strip().lower().decode("latin-1")] = value.strip().decode("latin-1")
        return status, reason, headers

    def _read_exact(self, n):
        out = bytearray(n)
        have = min(n, len(self._rbuf))
        out[:have] = self._rbuf[:have]
        del self._rbuf[:have]
        view = memoryview(out)
        while have < n:
            got = self.sock.recv_into(view[have:], n - have)
            if got == 0:
                raise ConnectionError("connection closed by peer mid-body")
            have += got
        return out

    def _read_line(self):
        while True:
            end = self._rbuf.find(_CRLF)
            if end >= 0:
                line = bytes(self._rbuf[:end])
                del self._rbuf[: end + 2]
                return line
            self._fill()

    def _read_chunked(self):
        body = bytearray()
        while True:
            size = int(self._read_line().split(b";", 1)[0].strip() or b"0", 16)
            if size == 0:
                while self._read_line():  # trailers
                    pass
                return body
            body += self._read_exact(size)
            self._read_line()

    def _read_until_close(self):
        body = bytearray(self._rbuf)
        self._rbuf = bytearray()
        while True:
            chunk = self.sock.recv(262144)
            if not chunk:
                return body
            body += chunk

    def request(self, method, target, host_header, headers, body_parts):
        """One request/response exchange.  Retries once on a stale keep-alive
        connection (server closed it while idle)."""
        total = sum(len(p) for p in body_parts)
        lines = [f"{method} {target} HTTP/1.1", f"Host: {host_header}"]
        lower = set()
        for k, v in headers.items():
            lines.append(f"{k}: {v}")
            lower.add(k.lower())
        if method != "GET" or total:
            lines.append(f"Content-Length: {total}")
        if "connection" not in lower:
            lines.append("Connection: keep-alive")
        head = ("\r\n".join(lines) + "\r\n\r\n").encode("latin-1")
        for attempt in (0, 1):
            fresh = self.sock is None
            if fresh:
                self._open()
            try:
                self._send_all([head, *body_parts])
                status, reason, rh = self._read_headers()
                break
            except (ConnectionError, BrokenPipeError, OSError) as e:
                self.close()
                if fresh or attempt == 1 or isinstance(e, socket.timeout):
                    raise
        while status == 100:  # interim response
            status, reason, rh = self._read_headers()
        if method == "HEAD" or status in (204, 304):
            body = bytearray()
        elif "chunked" in rh.get("transfer-encoding", "").lower():
            body = self._read_chunked()
        elif "content-length" in rh:
            body = self._read_exact(int(rh["content-length"]))
        else:
            body = self._read_until_close()
            self.close()
        if rh.get("connection", "").lower() == "close":
            self.close()
        return HttpResponse(status, rh, body, reason)


class ConnectionPool:
    """Fixed-size pool of keep-alive connections to one origin."""

    def __init__(self, host, port, size, connect_timeout, io_timeout, ssl_context=None, server_hostname=None):
        self._host_header = host if port in (80, 443) else f"{host}:{port}"
        self._idle = queue.LifoQueue()
        self._all = []
        for _ in range(max(1, int(size))):
            conn = _Connection(host, port, connect_timeout, io_timeout, ssl_context, server_hostname)
            self._all.append(conn)
            self._idle.put(conn)
        self._closed = False
        self._lock = threading.Lock()

    def request(self, method, target, headers, body_parts=()):
        if self._closed:
            raise InferenceServerException("the client has been closed")
        conn = self._idle.get()
        try:
            return conn.request(method, target, self._host_header, headers or {}, list(body_parts))
        except socket.timeout:
            conn.close()
            raise InferenceServerException("Deadline Exceeded", status="499") from None
        except (OSError, ConnectionError) as e:
            conn.close()
            raise InferenceServerException(f"failed to reach the inference server: {e}") from None
        finally:
            self._idle.put(conn)

    def close(self):
        with self._lock:
            if self._closed:
                return
            self._closed = True
        for conn in self._all:
            conn.close()

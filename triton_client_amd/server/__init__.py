"""Minimal KServe-v2 / Triton-protocol server used by tests, examples and the
load generator.

``Server``            in-process: HTTP on an asyncio loop thread + gRPC server
``spawn_server``      the same in a child process (required for hipshm: a HIP
                      IPC handle cannot be opened by the process that exported
                      it, see the reference README's ``--pid host`` note,
                      ``README.md:196-204``)

CLI: ``python -m triton_client_amd.server --http-port 8000 --grpc-port 8001
--gpu 0 --models simple,resnet50``
"""

from __future__ import annotations

import asyncio
import json
import os
import subprocess
import sys
import threading
import time

from .core import InferenceCore, ServerError  # noqa: F401


_arenas_limited = False


def _limit_malloc_arenas():
    """Caps the number of glibc malloc arenas of this process (``TCA_MALLOC_ARENAS``, default 8; 0 leaves the
    allocator alone).  Message bodies of megabytes are allocated on a reactor thread and freed on a completer
    thread; whenever a thread finds its arena locked glibc moves it to another one - up to eight per CPU core -
    and every arena keeps the large blocks that were freed into it.  On a machine with hundreds of cores the
    resident set of a server that receives 4.8 MB requests grew by 16 MB per second without end
    (``tools/dev/soak.py``); with the cap it levels off."""
    global _arenas_limited
    if _arenas_limited:
        return
    _arenas_limited = True
    try:
        n = int(os.environ.get("TCA_MALLOC_ARENAS", "8"))
        if n > 0 and "MALLOC_ARENA_MAX" not in os.environ:
            import ctypes

            ctypes.CDLL(None).mallopt(-8, n)  # M_ARENA_MAX
    except Exception:  # noqa: BLE001 - another C library: nothing to cap
        pass


class Server:
    """In-process server.  ``http_port`` / ``grpc_port`` of 0 pick free ports."""

    def __init__(self, models=None, gpu=None, http_port=0, grpc_port=0, host="127.0.0.1", model_control="none",
                 enable_http=True, enable_grpc=True, http_workers=16, grpc_workers=32, grpc_impl=None,
                 tls_cert=None, tls_key=None, tls_client_ca=None, grpc_reactors=None, http_impl=None,
                 http_reactors=None, native_exec=None):
        _limit_malloc_arenas()
        self.core = InferenceCore(models=models, gpu=gpu, model_control=model_control)
        self.host = host
        self._http_port = http_port
        self._grpc_port = grpc_port
        self._enable_http = enable_http
        self._enable_grpc = enable_grpc
        self._http_workers = http_workers
        self._grpc_workers = grpc_workers
        # "native": C++ HTTP/2 reactors + asynchronous completion; "grpcio": the stock
        # Python gRPC server.  Default from TCA_GRPC_IMPL, else native.
        self._grpc_impl = (grpc_impl or os.environ.get("TCA_GRPC_IMPL") or "native").lower()
        # "native": C++ HTTP/1.1 reactors + asynchronous completion; "asyncio": the pure-Python
        # front.  Default from TCA_HTTP_IMPL, else native.
        self._http_impl = (http_impl or os.environ.get("TCA_HTTP_IMPL") or "native").lower()
        # (four as well: `simple` over REST + system shm, answered on the reactors, 186k -> 270k infer/s at
        # concurrency 16)
        self._http_reactors = int(http_reactors or os.environ.get("TCA_HTTP_REACTORS") or 4)
        # TLS on both fronts when a certificate + key (PEM files) are given; `tls_client_ca`
        # additionally requires client certificates signed by that CA (mutual TLS)
        self._tls = (tls_cert, tls_key, tls_client_ca) if tls_cert and tls_key else None
        # epoll reactor threads of the native gRPC front (each owns the connections it accepted)
        # eight epoll threads: neutral for requests that carry region references; for requests that carry their
        # tensors (1.2 MB each at batch 8) receiving and staging is the reactors' work, and with four of them
        # the rate fell once more than ~32 requests were in flight (47k img/s at 32, 36k at 64, small batches on
        # the GPU); with eight it keeps rising: 49k / 54k / 58k img/s at 32 / 64 / 128
        self._grpc_reactors = int(grpc_reactors or os.environ.get("TCA_GRPC_REACTORS") or 8)
        # requests for models with a C++ kernel are answered by the native fronts without the
        # interpreter (tca/native_executor.h); TCA_NATIVE_EXEC=0 / native_exec=False turns that off
        if native_exec is None:
            native_exec = os.environ.get("TCA_NATIVE_EXEC", "1") != "0"
        self._native = None
        if native_exec:
            try:
                from .._native import pyext

                self._native = pyext().NativeExecutor()
                self.core.attach_native(self._native)
            except Exception:  # noqa: BLE001 - extension not built: everything takes the general path
                self._native = None
        self._loop = None
        self._thread = None
        self.http = None
        self.grpc = None

    @property
    def http_url(self):
        return f"{self.host}:{self._http_port}"

    @property
    def grpc_url(self):
        return f"{self.host}:{self._grpc_port}"

    def start(self):
        if self._enable_grpc:
            if self._grpc_impl == "native":
                from .native_grpc_front import NativeGrpcFront

                self.grpc = NativeGrpcFront(self.core, self.host, self._grpc_port, workers=self._grpc_workers,
                                            reactors=self._grpc_reactors, tls=self._tls, native=self._native)
            elif self._grpc_impl == "grpcio":
                from .grpc_front import GrpcFront

                self.grpc = GrpcFront(self.core, self.host, self._grpc_port, workers=self._grpc_workers, tls=self._tls)
            else:
                raise ValueError(f"unknown gRPC implementation '{self._grpc_impl}' (native | grpcio)")
            self._grpc_port = self.grpc.start()
        if self._enable_http and self._http_impl == "native":
            from .native_http_front import NativeHttpFront

            self.http = NativeHttpFront(self.core, self.host, self._http_port, workers=self._http_workers,
                                        reactors=self._http_reactors, tls=self._tls, native=self._native)
            self._http_port = self.http.start()
        elif self._enable_http:
            if self._http_impl != "asyncio":
                raise ValueError(f"unknown HTTP implementation '{self._http_impl}' (native | asyncio)")
            from .http_front import HttpFront

            ready = threading.Event()
            failure = []

            def run():
                loop = asyncio.new_event_loop()
                asyncio.set_event_loop(loop)
                self._loop = loop
                self.http = HttpFront(self.core, self.host, self._http_port, workers=self._http_workers,
                                      tls=self._tls)
                try:
                    self._http_port = loop.run_until_complete(self.http.start())
                except Exception as e:  # bind failure
                    failure.append(e)
                    ready.set()
                    return
                ready.set()
                loop.run_forever()
                loop.run_until_complete(self.http.stop())
                loop.close()

            self._thread = threading.Thread(target=run, name="tca-http-loop", daemon=True)
            self._thread.start()
            ready.wait()
            if failure:
                raise failure[0]
        return self

    def stop(self):
        if self.grpc is not None:
            self.grpc.stop()
            self.grpc = None
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._loop.stop)
            self._thread.join(timeout=5)
            self._loop = None
        elif self.http is not None:
            self.http.stop()
        self.http = None
        self.core.shutdown()

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()


class ServerProcess:
    """Handle of a server running in a child process."""

    def __init__(self, proc, info):
        self.proc = proc
        self.info = info
        self.http_url = f"127.0.0.1:{info['http_port']}"
        self.grpc_url = f"127.0.0.1:{info['grpc_port']}"

    def stop(self, timeout=10):
        if self.proc.poll() is None:
            try:
                self.proc.stdin.close()  # the child exits when its stdin closes
            except Exception:
                pass
            try:
                self.proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.wait()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()


def spawn_server(models=None, gpu=None, http_port=0, grpc_port=0, env=None, startup_timeout=600.0, extra_args=()):
    """Start ``python -m triton_client_amd.server`` as a child and wait until it
    reports its ports (one JSON line on stdout)."""
    cmd = [sys.executable, "-u", "-m", "triton_client_amd.server", "--http-port", str(http_port), "--grpc-port",
           str(grpc_port), "--announce", "--exit-on-stdin-close"]
    if models:
        cmd += ["--models", ",".join(models)]
    if gpu is not None:
        cmd += ["--gpu", str(gpu)]
    cmd += list(extra_args)
    child_env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    child_env["PYTHONPATH"] = root + os.pathsep + child_env.get("PYTHONPATH", "")
    if env:
        child_env.update(env)
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=child_env, cwd=root)
    deadline = time.time() + startup_timeout
    info = None
    while time.time() < deadline:
        line = proc.stdout.readline()
        if not line:
            if proc.poll() is not None:
                raise RuntimeError(f"server process exited with code {proc.returncode} before becoming ready")
            continue
        text = line.decode("utf-8", "replace").strip()
        if text.startswith("{") and "grpc_port" in text:
            info = json.loads(text)
            break
    if info is None:
        proc.kill()
        raise RuntimeError("server process did not become ready in time")
    # keep draining stdout so the child never blocks on a full pipe
    threading.Thread(target=lambda: [None for _ in iter(proc.stdout.readline, b"")], daemon=True).start()
    return ServerProcess(proc, info)

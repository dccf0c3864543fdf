"""Command line entry point of the in-repo v2 server."""

from __future__ import annotations

import argparse
import json
import os
import signal
import sys
import threading

from . import Server


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m triton_client_amd.server")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--http-port", type=int, default=8000)
    ap.add_argument("--grpc-port", type=int, default=8001)
    ap.add_argument("--gpu", type=int, default=None, help="MI355X ordinal served by this instance (omit: CPU only)")
    ap.add_argument("--models", default="", help="comma separated model names (default: every CPU model, plus the "
                                                 "GPU models when --gpu is given)")
    ap.add_argument("--model-control-mode", default="none", choices=["none", "explicit"])
    ap.add_argument("--http-workers", type=int, default=16)
    ap.add_argument("--grpc-workers", type=int, default=32)
    ap.add_argument("--grpc-impl", choices=["native", "grpcio"], default=None,
                    help="gRPC front: native C++ HTTP/2 reactors (default) or the stock Python gRPC server")
    ap.add_argument("--http-impl", choices=["native", "asyncio"], default=None,
                    help="HTTP front: C++ reactors (default) or the pure-Python asyncio one")
    ap.add_argument("--http-reactors", type=int, default=None,
                    help="epoll threads of the native HTTP front (default 4)")
    ap.add_argument("--grpc-reactors", type=int, default=None,
                    help="epoll threads of the native gRPC front (default 4)")
    ap.add_argument("--tls-cert", default=None, help="PEM certificate chain: serve HTTPS / gRPC over TLS")
    ap.add_argument("--tls-key", default=None, help="PEM private key of --tls-cert")
    ap.add_argument("--tls-client-ca", default=None, help="PEM CA bundle: require client certificates signed by it")
    ap.add_argument("--announce", action="store_true", help="print one JSON line with the bound ports when ready")
    ap.add_argument("--exit-on-stdin-close", action="store_true")
    args = ap.parse_args(argv)

    from ..models import DEFAULT_CPU_MODELS, MODEL_ZOO

    if args.models:
        models = [m for m in args.models.split(",") if m]
    else:
        models = list(DEFAULT_CPU_MODELS)
        if args.gpu is not None:
            models += [n for n, c in MODEL_ZOO.items() if c.device == "gpu" and getattr(c, "autoload", True)]
    if os.environ.get("TCA_SWITCH_INTERVAL_US"):
        sys.setswitchinterval(float(os.environ["TCA_SWITCH_INTERVAL_US"]) * 1e-6)
    server = Server(models=models, gpu=args.gpu, http_port=args.http_port, grpc_port=args.grpc_port, host=args.host,
                    model_control=args.model_control_mode, http_workers=args.http_workers,
                    grpc_workers=args.grpc_workers, grpc_impl=args.grpc_impl, grpc_reactors=args.grpc_reactors,
                    http_impl=args.http_impl,
                    http_reactors=args.http_reactors, tls_cert=args.tls_cert,
                    tls_key=args.tls_key, tls_client_ca=args.tls_client_ca).start()
    if args.announce:
        print(json.dumps({"http_port": server._http_port, "grpc_port": server._grpc_port,
                          "models": [m["name"] for m in server.core.repository_index(True)],
                          "failed": server.core._failed}), flush=True)
    else:
        print(f"serving HTTP on {server.http_url}, gRPC on {server.grpc_url}", flush=True)
    stop = threading.Event()
    signal.signal(signal.SIGTERM, lambda *_: stop.set())
    signal.signal(signal.SIGINT, lambda *_: stop.set())
    if args.exit_on_stdin_close:
        def watch():
            try:
                sys.stdin.buffer.read()
            finally:
                stop.set()

        threading.Thread(target=watch, daemon=True).start()
    stop.wait()
    server.stop()


if __name__ == "__main__":
    main()

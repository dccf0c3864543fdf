"""TLS end to end: the C++ HTTP and gRPC clients (in-tree TLS session on OpenSSL)
and the Python clients against the in-repo server serving HTTPS / gRPC-over-TLS,
including both gRPC fronts and mutual TLS.

The reference exposes the same switches (`HttpSslOptions`,
``src/c++/library/http_client.h:45-78``; `SslOptions`,
``src/c++/library/grpc_client.h:43-60``; Python ``ssl=`` arguments) but ships no
test for them; certificates here are generated per session with the openssl CLI."""

import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "bin")

pytestmark = pytest.mark.skipif(shutil.which("openssl") is None, reason="openssl CLI not available")


def _run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, **kw)


@pytest.fixture(scope="module")
def pki(tmp_path_factory):
    """A CA, a server certificate for localhost / 127.0.0.1 and a client certificate."""
    d = tmp_path_factory.mktemp("pki")
    p = lambda n: str(d / n)  # noqa: E731
    ext = d / "san.cnf"
    ext.write_text("subjectAltName=DNS:localhost,IP:127.0.0.1\n")

    def sh(*cmd):
        r = _run(list(cmd))
        assert r.returncode == 0, r.stderr

    sh("openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-days", "2", "-subj", "/CN=tca test CA",
       "-keyout", p("ca.key"), "-out", p("ca.pem"))
    for name, cn, extra in (("server", "localhost", ["-extfile", str(ext)]), ("client", "tca client", [])):
        sh("openssl", "req", "-newkey", "rsa:2048", "-nodes", "-subj", f"/CN={cn}", "-keyout", p(f"{name}.key"),
           "-out", p(f"{name}.csr"))
        sh("openssl", "x509", "-req", "-in", p(f"{name}.csr"), "-CA", p("ca.pem"), "-CAkey", p("ca.key"),
           "-CAcreateserial", "-days", "2", "-out", p(f"{name}.pem"), *extra)
    # an unrelated CA nobody trusts
    sh("openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-days", "2", "-subj", "/CN=other CA",
       "-keyout", p("other.key"), "-out", p("other.pem"))
    return {n: p(n) for n in ("ca.pem", "server.pem", "server.key", "client.pem", "client.key", "other.pem")}


@pytest.fixture(scope="module")
def cpp_bins():
    from triton_client_amd import _build

    _build.build_cpp()
    return BIN


def _server(pki, grpc_impl, client_ca=None):
    from triton_client_amd.server import Server

    return Server(models=["simple", "simple_sequence"], grpc_impl=grpc_impl, tls_cert=pki["server.pem"],
                  tls_key=pki["server.key"],
                  tls_client_ca=client_ca)


@pytest.mark.parametrize("grpc_impl", ["native", "grpcio"])
def test_cpp_clients_over_tls(pki, cpp_bins, grpc_impl):
    with _server(pki, grpc_impl) as srv:
        http = f"https://localhost:{srv._http_port}"
        grpc_url = f"localhost:{srv._grpc_port}"
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", http, "--ca-certs", pki["ca.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        r = _run([os.path.join(cpp_bins, "simple_http_async_infer_client"), "-u", http, "--ca-certs", pki["ca.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        r = _run([os.path.join(cpp_bins, "simple_grpc_infer_client"), "-u", grpc_url, "--ssl", "--root-certificates",
                  pki["ca.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        r = _run([os.path.join(cpp_bins, "simple_grpc_sequence_stream_infer_client"), "-u", grpc_url, "--ssl",
                  "--root-certificates", pki["ca.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr

        # a CA that did not sign the server certificate: verification must fail ...
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", http, "--ca-certs", pki["other.pem"]])
        assert r.returncode != 0 and "certificate" in r.stderr.lower(), r.stderr
        r = _run([os.path.join(cpp_bins, "simple_grpc_infer_client"), "-u", grpc_url, "--ssl", "--root-certificates",
                  pki["other.pem"]])
        assert r.returncode != 0 and "certificate" in r.stderr.lower(), r.stderr
        # ... unless the caller switches verification off
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", http, "--ca-certs", pki["other.pem"],
                  "--verify-peer", "0", "--verify-host", "0"])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        # the certificate names localhost and 127.0.0.1 only
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", f"https://127.0.0.1:{srv._http_port}",
                  "--ca-certs", pki["ca.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        # a cleartext client against the TLS port fails cleanly instead of hanging
        r = _run([os.path.join(cpp_bins, "simple_grpc_infer_client"), "-u", grpc_url, "-t", "2000000"])
        assert r.returncode != 0


@pytest.mark.parametrize("grpc_impl", ["native", "grpcio"])
def test_mutual_tls(pki, cpp_bins, grpc_impl):
    with _server(pki, grpc_impl, client_ca=pki["ca.pem"]) as srv:
        http = f"https://localhost:{srv._http_port}"
        grpc_url = f"localhost:{srv._grpc_port}"
        with_cert = ["--cert-file", pki["client.pem"], "--key-file", pki["client.key"]]
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", http, "--ca-certs", pki["ca.pem"],
                  *with_cert])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        r = _run([os.path.join(cpp_bins, "simple_http_infer_client"), "-u", http, "--ca-certs", pki["ca.pem"]])
        assert r.returncode != 0, "the server must refuse a client without a certificate"
        r = _run([os.path.join(cpp_bins, "simple_grpc_infer_client"), "-u", grpc_url, "--ssl", "--root-certificates",
                  pki["ca.pem"], "--private-key", pki["client.key"], "--certificate-chain", pki["client.pem"]])
        assert r.returncode == 0 and "PASS" in r.stdout, r.stderr
        r = _run([os.path.join(cpp_bins, "simple_grpc_infer_client"), "-u", grpc_url, "--ssl", "--root-certificates",
                  pki["ca.pem"]])
        assert r.returncode != 0, "the server must refuse a client without a certificate"


def test_python_clients_over_tls(pki):
    import tritonclient.grpc as grpcclient
    import tritonclient.http as httpclient

    a = np.arange(16, dtype=np.int32).reshape(1, 16)
    b = np.ones((1, 16), dtype=np.int32)
    with _server(pki, "native") as srv:
        client = httpclient.InferenceServerClient(f"localhost:{srv._http_port}", ssl=True,
                                                  ssl_options={"ca_certs": pki["ca.pem"]})
        inputs = [httpclient.InferInput("INPUT0", [1, 16], "INT32"), httpclient.InferInput("INPUT1", [1, 16], "INT32")]
        inputs[0].set_data_from_numpy(a)
        inputs[1].set_data_from_numpy(b)
        result = client.infer("simple", inputs)
        np.testing.assert_array_equal(result.as_numpy("OUTPUT0"), a + b)
        client.close()

        # grpcio client -> native TLS front (ALPN h2 negotiated by our server-side session)
        gclient = grpcclient.InferenceServerClient(f"localhost:{srv._grpc_port}", ssl=True,
                                                   root_certificates=pki["ca.pem"])
        ginputs = [grpcclient.InferInput("INPUT0", [1, 16], "INT32"), grpcclient.InferInput("INPUT1", [1, 16], "INT32")]
        ginputs[0].set_data_from_numpy(a)
        ginputs[1].set_data_from_numpy(b)
        gresult = gclient.infer("simple", ginputs)
        np.testing.assert_array_equal(gresult.as_numpy("OUTPUT1"), a - b)
        gclient.close()

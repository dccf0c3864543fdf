"""Race / memory-error detection for the native code: the client library, its
unit tests, a multi-threaded client stress test and the native serving fronts
(``grpc_front_stress`` / ``http_front_stress``: epoll reactors answered from completer
threads, the serving benchmark's hot path, driven in-process by client threads and raw
sockets) are rebuilt with
``-fsanitize=thread`` and with ``-fsanitize=address,undefined`` and run against
the in-repo server.  (GPU code cannot run under sanitizers on this pool; these
are host-only builds.)  The reference configures no sanitizers at all."""

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGETS = ("stress_test", "cc_unit_test", "grpc_client_test", "grpc_front_stress", "http_front_stress",
           "native_executor_test")


def _build(kind, san):
    out = os.path.join(ROOT, "build", kind)
    cmd = ["make", "-C", os.path.join(ROOT, "csrc"), "-j8", f"BUILD={out}", f"SAN={san}", "lib"]
    cmd += [os.path.join(out, "bin", t) for t in TARGETS]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return os.path.join(out, "bin")


@pytest.mark.parametrize("kind,san,env", [
    ("tsan", "thread", {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66 second_deadlock_stack=1"}),
    ("asan", "address,undefined", {"ASAN_OPTIONS": "detect_leaks=1 exitcode=66",
                                   "UBSAN_OPTIONS": "halt_on_error=1 print_stacktrace=1"}),
])
def test_native_clients_under_sanitizer(server, kind, san, env):
    bins = _build(kind, san)
    run_env = dict(os.environ, **env)

    def run(cmd, timeout=600):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=run_env)
        assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-1500:]}\n{r.stderr[-4000:]}"
        assert "Sanitizer" not in r.stderr, r.stderr[-4000:]
        return r

    run([os.path.join(bins, "cc_unit_test")])
    run([os.path.join(bins, "grpc_client_test"), "-u", server.grpc_url])
    r = run([os.path.join(bins, "stress_test"), "-u", server.http_url, "-g", server.grpc_url, "-n", "6", "-r", "10"])
    assert "PASS : stress" in r.stdout
    r = run([os.path.join(bins, "grpc_front_stress"), "-n", "6", "-r", "6"])
    assert "PASS : front stress" in r.stdout
    r = run([os.path.join(bins, "http_front_stress"), "-n", "6", "-r", "6"])
    assert "PASS : http front stress" in r.stdout
    r = run([os.path.join(bins, "native_executor_test")])
    assert "PASS : native executor" in r.stdout

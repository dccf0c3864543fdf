"""Load generator against the in-repo CPU server: concurrency sweep, request
rate mode, system shared memory, CSV export, multi-endpoint sharding."""
import csv

import pytest


@pytest.fixture(scope="module")
def two_servers():
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.server import Server

    a = Server(models=DEFAULT_CPU_MODELS).start()
    b = Server(models=DEFAULT_CPU_MODELS).start()
    yield a, b
    a.stop()
    b.stop()


def _run(argv):
    from triton_client_amd.perf import main

    return main(argv)


@pytest.mark.parametrize("proto", ["grpc", "http"])
def test_concurrency_sweep_and_csv(two_servers, proto, tmp_path):
    a, _ = two_servers
    url = a.grpc_url if proto == "grpc" else a.http_url
    out = tmp_path / "sweep.csv"
    rows = _run(["-m", "simple", "-u", url, "-i", proto, "--concurrency-range", "1:2", "-p", "300", "--warmup-ms",
                 "100", "-r", "4", "-s", "50", "--worker-processes", "2", "-f", str(out)])
    assert [r["level"] for r in rows] == [1, 2]
    for r in rows:
        assert r["infer_per_sec"] > 50 and r["p50_us"] > 0 and r["p99_us"] >= r["p50_us"] and r["errors"] == 0
        assert r["server_compute_infer_us"] >= 0
    with open(out) as fh:
        got = list(csv.DictReader(fh))
    assert len(got) == 2 and float(got[0]["infer_per_sec"]) == rows[0]["infer_per_sec"]


def test_system_shm_and_batch(two_servers):
    a, _ = two_servers
    rows = _run(["-m", "simple", "-u", a.grpc_url, "-b", "4", "--shared-memory", "system", "--concurrency-range", "2",
                 "-p", "300", "--warmup-ms", "100", "-r", "4", "-s", "50", "--worker-processes", "1"])
    assert rows[0]["errors"] == 0 and rows[0]["infer_per_sec"] > 100
    assert rows[0]["infer_per_sec"] == pytest.approx(rows[0]["requests_per_sec"] * 4, rel=1e-3)


def test_request_rate_mode(two_servers):
    a, _ = two_servers
    rows = _run(["-m", "simple", "-u", a.http_url, "-i", "http", "--request-rate-range", "100", "--max-threads", "4",
                 "-p", "400", "--warmup-ms", "100", "-r", "4", "-s", "50", "--worker-processes", "1",
                 "--request-distribution", "poisson"])
    assert 40 < rows[0]["requests_per_sec"] < 200


def test_multi_endpoint_sharding_and_string_model(two_servers):
    a, b = two_servers
    rows = _run(["-m", "simple_string", "--urls", f"{a.grpc_url},{b.grpc_url}", "--concurrency-range", "4", "-p",
                 "300", "--warmup-ms", "100", "-r", "4", "-s", "50", "--worker-processes", "2"])
    assert rows[0]["errors"] == 0 and rows[0]["infer_per_sec"] > 50
    # both servers saw traffic
    import tritonclient.grpc as g

    for srv in (a, b):
        with g.InferenceServerClient(srv.grpc_url) as c:
            st = c.get_inference_statistics("simple_string", as_json=True)
            assert int(st["model_stats"][0]["inference_stats"]["success"]["count"]) > 0


@pytest.mark.parametrize("proto,shm", [("grpc", "none"), ("http", "system"), ("grpc", "system")])
def test_native_generator_mode(two_servers, proto, shm):
    """--native: levels are driven by tca_loadgen processes, one per endpoint."""
    from triton_client_amd import _build

    _build.build_cpp()
    a, b = two_servers
    urls = f"{a.grpc_url},{b.grpc_url}" if proto == "grpc" else f"{a.http_url},{b.http_url}"
    rows = _run(["-m", "simple", "--urls", urls, "-i", proto, "-b", "2", "--shared-memory", shm, "--native",
                 "--concurrency-range", "2:4:2", "-p", "200", "--stability-windows", "2", "--warmup-ms", "100"])
    assert [r["level"] for r in rows] == [2, 4]
    for r in rows:
        assert r["errors"] == 0 and r["stable"] and r["requests"] > 20
        assert r["infer_per_sec"] == pytest.approx(r["requests_per_sec"] * 2, rel=1e-3)
        assert 0 < r["p50_us"] <= r["p99_us"]
    # regions were unregistered again
    import tritonclient.grpc as g

    with g.InferenceServerClient(a.grpc_url) as c:
        assert not c.get_system_shared_memory_status(as_json=True).get("regions")


@pytest.mark.parametrize("proto,shm", [("grpc", "none"), ("grpc", "system"), ("http", "system")])
def test_native_generator_drives_bytes_tensors(two_servers, proto, shm):
    """BYTES inputs (length-prefixed decimal strings the model parses) on the wire and
    serialized into shared-memory regions, with the BYTES outputs placed in regions."""
    from triton_client_amd import _build

    _build.build_cpp()
    a, _ = two_servers
    url = a.grpc_url if proto == "grpc" else a.http_url
    rows = _run(["-m", "simple_string", "-u", url, "-i", proto, "--shared-memory", shm, "--native",
                 "--concurrency-range", "3", "-p", "200", "--stability-windows", "2", "--warmup-ms", "100"])
    assert rows[0]["errors"] == 0 and rows[0]["requests"] > 20 and rows[0]["infer_per_sec"] > 50


def test_native_streaming_drives_a_decoupled_model(two_servers):
    """--native --streaming: requests on bidirectional streams; a decoupled model's request is
    complete at its final response and every response is counted."""
    from triton_client_amd import _build

    _build.build_cpp()
    a, _ = two_servers
    base = ["-u", a.grpc_url, "-i", "grpc", "--native", "--concurrency-range", "4", "-p", "200",
            "--stability-windows", "2",
            "--warmup-ms", "100"]
    rows = _run(["-m", "repeat_int32", "--streaming", "--shape", "IN:5", "--shape", "DELAY:5", "--shape",
                 "WAIT:1"] + base)
    r = rows[0]
    assert r["errors"] == 0 and r["requests"] > 10
    assert r["responses_per_sec"] == pytest.approx(5 * r["requests_per_sec"], rel=0.1)
    # request/response models work over the stream too: one response per request
    rows = _run(["-m", "simple", "--streaming"] + base)
    assert rows[0]["errors"] == 0 and rows[0]["responses_per_sec"] == pytest.approx(rows[0]["requests_per_sec"],
                                                                                    rel=0.05)
    with pytest.raises(SystemExit):  # without the stream a decoupled model cannot be measured
        _run(["-m", "repeat_int32", "--shape", "IN:5", "--shape", "DELAY:5", "--shape", "WAIT:1"] + base)


@pytest.mark.parametrize("mode", ["python-grpc", "python-http", "native-grpc", "native-http", "native-stream"])
def test_stateful_models_are_driven_with_sequences(two_servers, mode):
    """A model with a sequence batcher: every slot sends sequences (start flag, correlation id, end
    flag) - by default 20 requests long, `--sequence-length` changes it - and no request fails, which
    it would if a flag or an id were wrong (the model refuses a request without a started sequence)."""
    from triton_client_amd import _build

    _build.build_cpp()
    a, _ = two_servers
    proto = "http" if mode.endswith("http") else "grpc"
    argv = ["-m", "simple_sequence", "-u", a.http_url if proto == "http" else a.grpc_url, "-i", proto,
            "--concurrency-range", "3", "-p", "200", "--stability-windows", "2", "--warmup-ms", "100", "-r", "4", "-s",
            "50", "--worker-processes", "1"]
    if mode.startswith("native"):
        argv.append("--native")
    if mode == "native-stream":
        argv.append("--streaming")
    if mode in ("python-http", "native-grpc"):
        argv += ["--sequence-length", "3"]
    import tritonclient.grpc as g

    def counts():
        with g.InferenceServerClient(a.grpc_url) as c:
            st = c.get_inference_statistics("simple_sequence", as_json=True)["model_stats"][0]["inference_stats"]
            return int(st["success"].get("count", 0)), int(st["fail"].get("count", 0))

    ok0, fail0 = counts()
    rows = _run(argv)
    ok1, fail1 = counts()
    assert rows[0]["errors"] == 0 and rows[0]["requests"] > 10
    assert fail1 == fail0 and ok1 - ok0 >= rows[0]["requests"]
    # a timed native run ends the sequences it began: the server holds no state for them afterwards
    if mode.startswith("native"):
        assert not a.core._loaded["simple_sequence"].sequences
        assert a._native is None or a._native.open_sequences("simple_sequence") == 0


def test_native_mode_rejects_open_loop(two_servers):
    a, _ = two_servers
    with pytest.raises(SystemExit):
        _run(["-m", "simple", "-u", a.grpc_url, "--native", "--request-rate-range", "100"])


def test_resume_skips_finished_levels(two_servers, tmp_path):
    a, _ = two_servers
    progress = tmp_path / "progress.json"
    base = ["-m", "simple", "-u", a.grpc_url, "-p", "200", "--warmup-ms", "50", "-r", "3", "-s", "60",
            "--stability-windows", "2", "--worker-processes", "1", "--resume", str(progress)]
    first = _run(base + ["--concurrency-range", "1:2"])
    assert [r["level"] for r in first] == [1, 2]
    # same settings, longer range: levels 1 and 2 come from the file (bit-identical rows), 3 is measured
    second = _run(base + ["--concurrency-range", "1:3"])
    assert [r["level"] for r in second] == [1, 2, 3]
    assert second[0] == first[0] and second[1] == first[1]
    import json

    assert [r["level"] for r in json.load(open(progress))["rows"]] == [1, 2, 3]
    # different settings invalidate the file
    third = _run(base + ["--concurrency-range", "1", "-b", "2"])
    assert third[0]["level"] == 1 and third[0] != first[0]


def test_gpu_metrics_sampling(two_servers, tmp_path, monkeypatch):
    """--gpu-metrics with a stand-in rocm-smi on PATH (no GPU in CI)."""
    import json
    import os
    import stat

    from triton_client_amd.perf.analyzer import GpuSampler

    sample = {"card0": {"GPU use (%)": "80", "Average Graphics Package Power (W)": "500.0",
                        "GPU Memory Allocated (VRAM%)": "10"},
              "card1": {"GPU use (%)": "60", "Current Socket Graphics Package Power (W)": "300.0",
                        "GPU Memory Allocated (VRAM%)": "30"},
              "system": {"Driver version": "6.0"}}
    parsed = GpuSampler.parse("WARNING: AMD GPU device(s) is/are in a low-power state.\n\n" + json.dumps(sample))
    assert parsed == {"gpu_util_pct": 70.0, "gpu_power_w": 400.0, "gpu_mem_pct": 20.0}

    tool = tmp_path / "rocm-smi"
    tool.write_text("#!/bin/sh\ncat <<'JSON'\n" + json.dumps(sample) + "\nJSON\n")
    tool.chmod(tool.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}{os.pathsep}{os.environ['PATH']}")
    a, _ = two_servers
    rows = _run(["-m", "simple", "-u", a.grpc_url, "--concurrency-range", "2", "-p", "400", "--warmup-ms", "50", "-r",
                 "3", "-s", "60", "--stability-windows", "2", "--worker-processes", "1", "--gpu-metrics"])
    assert rows[0]["gpu_util_pct"] == 70.0 and rows[0]["gpu_power_w"] == 400.0 and rows[0]["gpu_mem_pct"] == 20.0


def test_dead_endpoint_is_dropped_and_load_resharded():
    """A sweep over two endpoints keeps going on the survivor after one dies."""
    from triton_client_amd import _build
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.server import Server

    _build.build_cpp()
    a = Server(models=DEFAULT_CPU_MODELS).start()
    b = Server(models=DEFAULT_CPU_MODELS).start()
    try:
        urls = f"{a.grpc_url},{b.grpc_url}"
        b.stop()
        for extra in (["--native"], ["--worker-processes", "1", "-r", "3", "-s", "60"]):
            rows = _run(["-m", "simple", "--urls", urls, "--concurrency-range", "2", "-p", "200",
                         "--stability-windows", "2", "--warmup-ms", "50"] + extra)
            assert rows[0]["errors"] == 0 and rows[0]["requests"] > 10
    finally:
        a.stop()

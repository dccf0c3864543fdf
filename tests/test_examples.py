"""Runs every program under ``examples/`` against the in-repo server, the way
the reference's CI runs its examples against a live Triton.  Each example
checks its own results and prints ``PASS``; a non-zero exit fails the test.

Device shared-memory examples use host-backed regions (``TCA_HIPSHM_FAKE=1``)
on a machine without a GPU and real HIP IPC on the GPU box."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = os.path.join(ROOT, "examples")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


SHM_ENV = {} if _has_gpu() else {"TCA_HIPSHM_FAKE": "1"}


@pytest.fixture(scope="module")
def live():
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.server import spawn_server

    srv = spawn_server(models=DEFAULT_CPU_MODELS, env=SHM_ENV)
    yield srv
    srv.stop()


@pytest.fixture(scope="module")
def images(tmp_path_factory):
    from PIL import Image

    root = tmp_path_factory.mktemp("images")
    rng = np.random.default_rng(7)
    for i, (h, w) in enumerate([(300, 400), (256, 256), (120, 333)]):
        base = rng.integers(0, 255, size=(h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        img = Image.fromarray(base).resize((w, h), Image.BILINEAR)
        img.save(root / f"img{i}.{'png' if i == 1 else 'jpg'}")
    return str(root)


def _class_rows(text):
    """``(value, index, label)`` of every printed classification line."""
    rows = []
    for line in text.splitlines():
        if line.startswith("    "):
            value, rest = line.strip().split(" (", 1)
            index, label = rest.split(") = ", 1)
            rows.append((float(value), int(index), label))
    return rows


def run_example(name, *args, env=None, timeout=120):
    full_env = dict(os.environ)
    full_env["PYTHONPATH"] = ROOT + os.pathsep + full_env.get("PYTHONPATH", "")
    full_env.update(SHM_ENV)
    if env:
        full_env.update(env)
    proc = subprocess.run([sys.executable, os.path.join(EXAMPLES, name), *args], capture_output=True, text=True,
                          timeout=timeout, env=full_env, cwd=EXAMPLES)
    assert proc.returncode == 0, f"{name} failed\nstdout:\n{proc.stdout[-3000:]}\nstderr:\n{proc.stderr[-3000:]}"
    assert "PASS" in proc.stdout, proc.stdout[-2000:]
    return proc.stdout


HTTP_EXAMPLES = [
    "simple_http_infer_client.py",
    "simple_http_async_infer_client.py",
    "simple_http_aio_infer_client.py",
    "simple_http_string_infer_client.py",
    "simple_http_shm_client.py",
    "simple_http_shm_string_client.py",
    "simple_http_cudashm_client.py",
    "simple_http_sequence_sync_infer_client.py",
    "simple_http_health_metadata.py",
    "simple_http_model_control.py",
]
GRPC_EXAMPLES = [
    "simple_grpc_infer_client.py",
    "simple_grpc_async_infer_client.py",
    "simple_grpc_aio_infer_client.py",
    "simple_grpc_string_infer_client.py",
    "simple_grpc_shm_client.py",
    "simple_grpc_shm_string_client.py",
    "simple_grpc_cudashm_client.py",
    "simple_grpc_sequence_sync_infer_client.py",
    "simple_grpc_sequence_stream_infer_client.py",
    "simple_grpc_aio_sequence_stream_infer_client.py",
    "simple_grpc_custom_repeat.py",
    "simple_grpc_health_metadata.py",
    "simple_grpc_model_control.py",
    "simple_grpc_keepalive_client.py",
    "simple_grpc_custom_args_client.py",
    "grpc_client.py",
    "grpc_explicit_int_content_client.py",
    "grpc_explicit_int8_content_client.py",
    "grpc_explicit_byte_content_client.py",
]


@pytest.mark.parametrize("name", HTTP_EXAMPLES)
def test_http_example(live, name):
    run_example(name, "-u", live.http_url)


@pytest.mark.parametrize("name", GRPC_EXAMPLES)
def test_grpc_example(live, name):
    run_example(name, "-u", live.grpc_url)


@pytest.mark.parametrize("name", ["simple_http_sequence_sync_infer_client.py",
                                  "simple_grpc_sequence_sync_infer_client.py",
                                  "simple_grpc_sequence_stream_infer_client.py",
                                  "simple_grpc_aio_sequence_stream_infer_client.py"])
def test_dyna_sequence_example(live, name):
    url = live.http_url if "_http_" in name else live.grpc_url
    run_example(name, "-u", url, "-d", "-o", "3")


@pytest.mark.parametrize("name", ["simple_http_cudashm_client.py", "simple_grpc_cudashm_client.py"])
def test_device_shm_string_example(live, name):
    url = live.http_url if "_http_" in name else live.grpc_url
    run_example(name, "-u", url, "--string")


def test_reuse_infer_objects(live):
    run_example("reuse_infer_objects_client.py", "-u", live.http_url, "--grpc-url", live.grpc_url)


@pytest.mark.parametrize("protocol", ["http", "grpc"])
@pytest.mark.parametrize("reuse", [False, True])
def test_memory_growth(live, protocol, reuse):
    url = live.http_url if protocol == "http" else live.grpc_url
    run_example("memory_growth_test.py", "-u", url, "-i", protocol, "-r", "100", *(["--reuse"] if reuse else []))


@pytest.mark.parametrize("protocol,mode", [("http", ()), ("http", ("-a",)), ("grpc", ()), ("grpc", ("-a",)),
                                           ("grpc", ("--streaming",))])
def test_image_client(live, images, protocol, mode):
    url = live.http_url if protocol == "http" else live.grpc_url
    out = run_example("image_client.py", "-u", url, "-i", protocol, "-m", "tiny_classifier", "-b", "2", "-c", "3", "-s",
                      "INCEPTION", *mode, images)
    assert out.count("Request ") == 2 and out.count("Image '") == 4
    assert " = " in out


def test_image_client_nhwc_single(live, images):
    out = run_example("image_client.py", "-u", live.http_url, "-m", "tiny_classifier_nhwc", "-s", "VGG",
                      os.path.join(images, "img0.jpg"))
    assert out.count("Image '") == 1


@pytest.mark.parametrize("mode", [(), ("-a",), ("--streaming",)])
def test_grpc_image_client(live, images, mode):
    out = run_example("grpc_image_client.py", "-u", live.grpc_url, "-m", "tiny_classifier", "-b", "3", "-c", "2", *mode,
                      images)
    assert out.count("Image '") == 3


def test_image_clients_agree(live, images):
    """Library client and raw-stub client must print identical classes."""
    a = run_example("image_client.py", "-u", live.grpc_url, "-i", "grpc", "-m", "tiny_classifier", "-c", "3", "-s",
                    "INCEPTION", images)
    b = run_example("grpc_image_client.py", "-u", live.grpc_url, "-m", "tiny_classifier", "-c", "3", "-s", "INCEPTION",
                    images)
    pick = lambda text: [line for line in text.splitlines() if line.startswith("    ")]  # noqa: E731
    assert pick(a) == pick(b) and len(pick(a)) == 9


@pytest.mark.parametrize("protocol", ["http", "grpc"])
def test_ensemble_image_client(live, images, protocol):
    url = live.http_url if protocol == "http" else live.grpc_url
    out = run_example("ensemble_image_client.py", "-u", url, "-i", protocol, "-b", "2", "-c", "2", images)
    assert out.count("Image '") == 4


def test_ensemble_matches_client_side_preprocessing(live, images):
    """Server-side decode+preprocess (ensemble) and client-side preprocessing
    feed the same classifier, so the printed classes must agree."""
    a = run_example("ensemble_image_client.py", "-u", live.http_url, "-c", "3", os.path.join(images, "img1.png"))
    b = run_example("image_client.py", "-u", live.http_url, "-m", "tiny_classifier", "-c", "3", "-s", "INCEPTION",
                    os.path.join(images, "img1.png"))
    rows_a, rows_b = _class_rows(a), _class_rows(b)
    assert [r[1:] for r in rows_a] == [r[1:] for r in rows_b] and len(rows_a) == 3
    assert np.allclose([r[0] for r in rows_a], [r[0] for r in rows_b], atol=1e-4)


@pytest.mark.gpu
def test_image_client_hip_preprocess_matches_host(live, images):
    """GPU preprocessing into device shared memory gives the same classes as
    the host pipeline (the kernel is bit-exact with Pillow before scaling)."""
    common = ["-u", live.grpc_url, "-i", "grpc", "-m", "tiny_classifier", "-b", "2", "-c", "3", "-s", "INCEPTION"]
    host = run_example("image_client.py", *common, images)
    dev = run_example("image_client.py", *common, "--preprocess", "hip", images)
    pick = lambda text: [line.split(" (")[1] for line in text.splitlines() if line.startswith("    ")]  # noqa: E731
    assert pick(host) == pick(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("-a",), ("--streaming",)])
def test_image_client_hip_ring_keeps_requests_in_flight(live, images, mode):
    """More requests than ring slots: a slot is refilled only after the response of the request
    that used it, and the classes still match the host pipeline."""
    common = ["-u", live.grpc_url, "-i", "grpc", "-m", "tiny_classifier", "-b", "1", "-c", "2", "-s", "INCEPTION"]
    host = run_example("image_client.py", *common, images)
    dev = run_example("image_client.py", *common, *mode, "--preprocess", "hip", "--ring-depth", "2", images)
    pick = lambda text: [line.split(" (")[1] for line in text.splitlines() if line.startswith("    ")]  # noqa: E731
    assert pick(host) == pick(dev) and len(pick(dev)) == 6

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def server():
    """In-process CPU server with every plumbing model, shared by the session."""
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.server import Server

    srv = Server(models=DEFAULT_CPU_MODELS).start()
    yield srv
    srv.stop()


@pytest.fixture(params=["http", "grpc"])
def proto(request):
    return request.param


@pytest.fixture
def client(server, proto):
    if proto == "http":
        import tritonclient.http as mod

        c = mod.InferenceServerClient(server.http_url, concurrency=4)
    else:
        import tritonclient.grpc as mod

        c = mod.InferenceServerClient(server.grpc_url)
    yield mod, c
    c.close()

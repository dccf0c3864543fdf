"""The user guide names real things: every client call, utility and switch it lists exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(name):
    with open(os.path.join(ROOT, "docs", name)) as fh:
        return fh.read()


def test_python_client_calls_exist():
    import tritonclient.grpc as grpcclient
    import tritonclient.http as httpclient
    import tritonclient.utils as utils
    import tritonclient.utils.cuda_shared_memory as hipshm
    import tritonclient.utils.shared_memory as shm

    text = _read("python_client.md")
    calls = text.split("## Calls (both protocols)")[1].split("## HTTP options")[0]
    names = set(re.findall(r"`([a-z_]+)(?:\(|`)", calls))
    names |= {"unregister_system_shared_memory", "unregister_cuda_shared_memory"}
    grpc_only = {"start_stream", "async_stream_infer", "stop_stream"}
    for name in sorted(names - {"headers", "query_params", "client_timeout", "request_id", "sequence_id",
                                "sequence_start", "sequence_end", "priority", "timeout", "parameters"}):
        assert hasattr(grpcclient.InferenceServerClient, name), name
        if name not in grpc_only:
            assert hasattr(httpclient.InferenceServerClient, name), name
    for name in re.findall(r"`([a-z_0-9]+)`", text.split("## Utilities")[1].split("## Plugins")[0]):
        if name.startswith("_"):
            continue
        assert hasattr(utils, name) or name == "raise_error", name
    for fn in ("create_shared_memory_region", "set_shared_memory_region", "get_contents_as_numpy",
               "destroy_shared_memory_region"):
        assert hasattr(shm, fn) and hasattr(hipshm, fn), fn
    for fn in ("get_raw_handle", "set_shared_memory_region_from_dlpack", "as_shared_memory_tensor",
               "set_shared_memory_region_bf16", "get_contents_as_numpy_bf16", "device_memory_info"):
        assert hasattr(hipshm, fn), fn
    assert hasattr(httpclient.InferenceServerClient, "generate_request_body")
    assert hasattr(httpclient.InferenceServerClient, "parse_response_body")
    assert hasattr(grpcclient, "KeepAliveOptions")
    from triton_client_amd.ops.preprocess import Preprocessor
    from triton_client_amd.pipeline import ImagePipeline

    assert hasattr(Preprocessor, "pin") and hasattr(Preprocessor, "run_pinned") and hasattr(ImagePipeline, "run")


def test_documented_models_headers_tools_and_switches_exist():
    from triton_client_amd.models import MODEL_ZOO

    server = _read("server.md")
    table = server.split("## Models")[1].split("A model is a subclass")[0]
    for name in re.findall(r"`([a-z0-9_]+)`", table):
        if name in ("execute_delay_ms", "identity", "repeat", "bytes_add_sub", "bytes_identity", "sequence",
                    "dyna_sequence"):
            continue  # a parameter and the native kernels the notes column names
        assert name in MODEL_ZOO, name
    kernels = re.findall(r'spec\.kernel == "([a-z_]+)"', open(os.path.join(ROOT, "csrc/server/native_executor.cc")).read())
    for name in ("identity", "repeat", "bytes_add_sub", "bytes_identity", "add_sub", "sequence", "dyna_sequence"):
        assert name in kernels, name
    cpp = _read("cpp_client.md")
    for header in set(re.findall(r"`(tca/[a-z0-9_]+\.h)`", cpp)):
        assert os.path.exists(os.path.join(ROOT, "csrc", "include", header)), header
    for tool in ("client_timeout_test", "memory_leak_test", "stress_test", "grpc_front_stress", "http_front_stress",
                 "native_executor_test"):
        assert os.path.exists(os.path.join(ROOT, "csrc", "tests", tool + ".cc")), tool
    perf = _read("perf.md") + _read("kernels.md") + _read("multi_gpu.md")
    for path in set(re.findall(r"`(tools/[A-Za-z0-9_/.]+\.(?:py|sh))`", perf)):
        assert os.path.exists(os.path.join(ROOT, path)), path
    for path in set(re.findall(r"`([a-z_]+\.(?:hip|cc))`", _read("kernels.md"))):
        assert os.path.exists(os.path.join(ROOT, "csrc", "hip", path)), path
    # environment switches named in the README are read somewhere in the sources
    readme = open(os.path.join(ROOT, "README.md")).read()
    switches = set(re.findall(r"`(TCA_[A-Z_]+)`", readme.split("## Switches")[1].split("## Layout")[0]))
    sources = ""
    for base, _, files in os.walk(ROOT):
        if any(part in base for part in (".git", "gpurun_out", "build", "__pycache__", "profiles", "docs")):
            continue
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip", ".sh")):
                with open(os.path.join(base, f), errors="ignore") as fh:
                    sources += fh.read()
    for name in sorted(switches):
        probe = name if not name.endswith("_") else name.rstrip("_")
        assert probe in sources, name

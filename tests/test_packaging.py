"""Packaging: the pure-Python wheel builds offline and carries every package, the proto sources and
the tuned MIOpen databases; the CMake project configures and exports the package config."""
import os
import shutil
import subprocess
import sys
import zipfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generic_wheel_contents(tmp_path):
    out = tmp_path / "wheels"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_wheel.py"), "--dest-dir", str(out), "--generic"],
                   check=True, capture_output=True, timeout=300)
    wheels = [f for f in os.listdir(out) if f.endswith(".whl")]
    assert len(wheels) == 1 and wheels[0].endswith("py3-none-any.whl")
    names = zipfile.ZipFile(out / wheels[0]).namelist()
    for needed in ("triton_client_amd/__init__.py", "triton_client_amd/grpc/protos/grpc_service.proto",
                   "triton_client_amd/models/miopen_db/", "tritonclient/__init__.py", "tritongrpcclient/__init__.py",
                   "triton_client_amd/server/native_grpc_front.py", "triton_client_amd/perf/analyzer.py"):
        assert any(n.startswith(needed) for n in names), needed
    assert not any(n.endswith(".so") for n in names)
    meta = [n for n in names if n.endswith("METADATA")][0]
    text = zipfile.ZipFile(out / wheels[0]).read(meta).decode()
    assert "Provides-Extra: grpc" in text and "Provides-Extra: hip" in text and "Provides-Extra: all" in text


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_project_configures(tmp_path):
    build = tmp_path / "b"
    r = subprocess.run(["cmake", "-S", os.path.join(ROOT, "csrc"), "-B", str(build), "-DTCA_ENABLE_EXAMPLES=OFF",
                        "-DTCA_ENABLE_TESTS=OFF"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (build / "TcaClientConfig.cmake").exists()


def test_cmake_build_install_and_find_package(tmp_path):
    """The CMake path: configure, build the library and the unit tests, run ctest, install,
    then build a consumer project that uses ``find_package(TcaClient)``."""
    import shutil
    import subprocess

    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build, prefix = tmp_path / "b", tmp_path / "prefix"

    def run(*cmd, cwd=None):
        p = subprocess.run([str(c) for c in cmd], cwd=cwd, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, f"{cmd}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
        return p.stdout

    run("cmake", "-S", os.path.join(root, "csrc"), "-B", build, "-DCMAKE_BUILD_TYPE=Release",
        "-DTCA_ENABLE_EXAMPLES=OFF", f"-DCMAKE_INSTALL_PREFIX={prefix}")
    run("cmake", "--build", build, "-j", "8", "--target", "tcaclient", "tcaclient_static", "cc_unit_test",
        "tca_loadgen")
    out = run("ctest", "-R", "cc_unit_test", "--output-on-failure", cwd=build)
    assert "100% tests passed" in out
    run("cmake", "--build", build, "-j", "8")
    run("cmake", "--install", build)
    assert (prefix / "include" / "tca" / "http_client.h").exists()
    consumer = tmp_path / "consumer"
    consumer.mkdir()
    (consumer / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.18)\nproject(consumer LANGUAGES CXX)\nset(CMAKE_CXX_STANDARD 17)\n"
        "find_package(TcaClient REQUIRED)\nadd_executable(app main.cc)\n"
        "target_link_libraries(app PRIVATE TcaClient::tcaclient_static)\n")
    (consumer / "main.cc").write_text(
        '#include "tca/http_client.h"\n#include <iostream>\nint main() {\n'
        '  std::unique_ptr<tca::client::InferenceServerHttpClient> c;\n'
        '  auto err = tca::client::InferenceServerHttpClient::Create(&c, "127.0.0.1:1");\n'
        '  std::cout << (err.IsOk() ? "created" : err.Message()) << std::endl;\n  return 0;\n}\n')
    run("cmake", "-S", consumer, "-B", consumer / "b", f"-DCMAKE_PREFIX_PATH={prefix}")
    run("cmake", "--build", consumer / "b")
    assert "created" in run(consumer / "b" / "app")

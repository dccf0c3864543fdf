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

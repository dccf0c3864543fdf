#!/usr/bin/env python3
"""Packaging of the MI355X-native client stack.

``pip install .`` (or ``python tools/build_wheel.py``) installs ``triton_client_amd`` together with
the API-compatible alias packages (``tritonclient``, ``tritonhttpclient``, ``tritongrpcclient``,
``tritonclientutils``, ``tritonshmutils``).  The native libraries are compiled by
``triton_client_amd._build`` during ``build_py`` (host parts with g++, HIP parts with hipcc for
gfx950; hipcc cross-compiles without a GPU) and shipped inside ``triton_client_amd/_lib``.

Extras mirror the reference wheel (``src/python/library/setup.py:69-86``): ``http``, ``grpc``,
``hip`` (the reference's ``cuda``) and ``all``.
"""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _req(name):
    path = os.path.join(HERE, "requirements", name)
    with open(path) as f:
        return [line.strip() for line in f if line.strip() and not line.startswith("#")]


class BuildWithNative(build_py):
    """Compile the shared objects before the package data is collected."""

    def run(self):
        if os.environ.get("TCA_SKIP_NATIVE_BUILD") != "1":
            sys.path.insert(0, HERE)
            from triton_client_amd import _build

            targets = ["core", "pyext"]
            if os.environ.get("TCA_SKIP_HIP_BUILD") != "1":
                targets.append("hip")
            _build.build(targets)
        super().run()


extras = {"http": _req("requirements_http.txt"), "grpc": _req("requirements_grpc.txt"),
          "hip": _req("requirements_hip.txt")}
extras["cuda"] = extras["hip"]  # name used by the reference wheel
extras["all"] = sorted({r for group in extras.values() for r in group})

setup(
    name="triton_client_amd",
    version=os.environ.get("TCA_VERSION", "0.1.0"),
    description="MI355X-native inference client stack with the tritonclient API (HTTP/gRPC, hipshm, HIP kernels)",
    long_description=open(os.path.join(HERE, "README.md")).read(),
    long_description_content_type="text/markdown",
    python_requires=">=3.8",
    packages=find_packages(include=["triton_client_amd*", "tritonclient*", "tritonhttpclient*", "tritongrpcclient*",
                                    "tritonclientutils*", "tritonshmutils*"]),
    package_data={
        "triton_client_amd": ["_lib/*.so", "_lib/tca_loadgen", "grpc/protos/*.proto", "models/miopen_db/*.txt",
                              "ops/*.json"],
    },
    install_requires=_req("requirements.txt"),
    extras_require=extras,
    cmdclass={"build_py": BuildWithNative},
    entry_points={"console_scripts": ["tca-perf-analyzer=triton_client_amd.perf.__main__:main",
                                      "tca-server=triton_client_amd.server.__main__:main"]},
    zip_safe=False,
    classifiers=["Programming Language :: Python :: 3", "Operating System :: POSIX :: Linux",
                 "Topic :: Scientific/Engineering :: Artificial Intelligence"],
)

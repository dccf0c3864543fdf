#!/usr/bin/env python3
"""Builds the wheel(s) of this repository into ``--dest-dir``.

Two flavours, like the reference's ``src/python/library/build_wheel.py:106-212``:

``--linux``    platform wheel carrying the native libraries (``libtca_core.so``, ``_tca_pyext``,
               ``libtca_hip.so`` for gfx950 and the ``tca_loadgen`` load generator)
``--generic``  pure-Python wheel (no shared objects; the codec falls back to numpy, system shared
               memory and device shared memory are unavailable until the libraries are built with
               ``python -m triton_client_amd._build``)

No network access is needed: the build uses the installed setuptools / wheel only.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(dest, linux, version, skip_hip):
    env = dict(os.environ)
    env["TCA_VERSION"] = version
    if not linux:
        env["TCA_SKIP_NATIVE_BUILD"] = "1"
    if skip_hip:
        env["TCA_SKIP_HIP_BUILD"] = "1"
    stage = tempfile.mkdtemp(prefix="tca_wheel_")
    try:
        # stage a copy so that build/ and *.egg-info never appear in the source tree
        for name in ("triton_client_amd", "tritonclient", "tritonhttpclient", "tritongrpcclient", "tritonclientutils",
                     "tritonshmutils", "requirements", "csrc"):
            shutil.copytree(os.path.join(ROOT, name), os.path.join(stage, name),
                            ignore=shutil.ignore_patterns("__pycache__", "*.pyc",
                                                          *(() if linux else ("*.so", "tca_loadgen"))))
        for name in ("setup.py", "README.md"):
            shutil.copy(os.path.join(ROOT, name), stage)
        cmd = [sys.executable, "setup.py", "-q", "bdist_wheel", "-d", os.path.abspath(dest)]
        if linux:
            cmd += ["--plat-name", "manylinux2014_x86_64"]
        subprocess.check_call(cmd, cwd=stage, env=env)
    finally:
        shutil.rmtree(stage, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dest-dir", required=True)
    ap.add_argument("--linux", action="store_true", help="platform wheel with the native libraries")
    ap.add_argument("--generic", action="store_true", help="pure-Python wheel")
    ap.add_argument("--version", default="0.1.0")
    ap.add_argument("--skip-hip", action="store_true", help="leave libtca_hip.so out of the linux wheel")
    args = ap.parse_args()
    if not (args.linux or args.generic):
        args.generic = True
    os.makedirs(args.dest_dir, exist_ok=True)
    if args.generic:
        build(args.dest_dir, False, args.version, True)
    if args.linux:
        build(args.dest_dir, True, args.version, args.skip_hip)
    for f in sorted(os.listdir(args.dest_dir)):
        print(os.path.join(args.dest_dir, f))


if __name__ == "__main__":
    main()

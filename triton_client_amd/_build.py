"""In-tree native build driver.

``python -m triton_client_amd._build [targets...]`` compiles the native pieces
into ``triton_client_amd/_lib`` (shared objects loaded by Python) and
``build/`` (C++ client library, examples and tests).  Everything targets
gfx950 only; ``hipcc`` cross-compiles without a GPU so the same command works
on a CPU-only build host.

Targets
-------
core   : libtca_core.so      (g++   – system shm, codec, base64, image decoding C ABI)
pyext  : _tca_pyext*.so      (g++   – CPython BYTES pack/unpack, native gRPC serving front)
hip    : libtca_hip.so       (hipcc – hipshm + CDNA4 kernels)
cpp    : build/libtcaclient.{a,so}, C++ examples, tests and the perf driver
all    : everything above (default)

A target is rebuilt only when one of its sources is newer than its output.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "csrc"
LIBDIR = Path(__file__).resolve().parent / "_lib"
BUILD = ROOT / "build"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
GFX = "gfx950"

CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-I", str(CSRC / "include")]


def _stale(out: Path, srcs) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = list(srcs) + list((CSRC / "include" / "tca").glob("*.h"))
    return any(Path(s).stat().st_mtime > t for s in deps)


def _run(cmd):
    cmd = [str(c) for c in cmd]
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _compile_atomic(cmd, out: Path):
    """Run a compile that writes ``out`` via a temp name then renames, so that
    a concurrently importing process never maps a half-written library."""
    tmp = out.with_name(out.name + f".tmp{os.getpid()}")
    _run([tmp if c is out else c for c in cmd])
    os.replace(tmp, out)


def build_core(force=False):
    out = LIBDIR / "libtca_core.so"
    srcs = [CSRC / "core" / "base64.cc", CSRC / "core" / "codec.cc", CSRC / "core" / "shm_utils.cc",
            CSRC / "core" / "common.cc", CSRC / "core" / "image_io.cc", CSRC / "core" / "jpeg.cc",
            CSRC / "capi" / "core_capi.cc"]
    if force or _stale(out, srcs):
        LIBDIR.mkdir(exist_ok=True)
        _compile_atomic(["g++", *CXXFLAGS, "-shared", "-o", out, *srcs, "-lrt", "-lz", "-pthread"], out)
    return out


def build_pyext(force=False):
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    out = LIBDIR / f"_tca_pyext{suffix}"
    srcs = [CSRC / "capi" / "pyext.cc", CSRC / "capi" / "py_grpc_front.cc", CSRC / "capi" / "py_http_front.cc",
            CSRC / "capi" / "py_native_executor.cc", CSRC / "server" / "native_executor.cc",
            CSRC / "grpc" / "grpc_server.cc", CSRC / "http" / "http_server.cc",
            CSRC / "grpc" / "grpc_compress.cc", CSRC / "h2" / "hpack.cc", CSRC / "h2" / "connection.cc",
            CSRC / "net" / "tls.cc"]
    if force or _stale(out, srcs):
        LIBDIR.mkdir(exist_ok=True)
        inc = sysconfig.get_paths()["include"]
        _compile_atomic(["g++", *CXXFLAGS, "-I", inc, "-shared", "-o", out, *srcs, "-pthread", "-lssl", "-lcrypto",
                         "-lz"], out)
    return out


def hip_sources():
    return sorted((CSRC / "hip").glob("*.hip")) + sorted((CSRC / "hip").glob("*.cc"))


def build_hip(force=False):
    out = LIBDIR / "libtca_hip.so"
    srcs = hip_sources()
    hdrs = list((CSRC / "hip").glob("*.h")) + list((CSRC / "hip").glob("*.hpp"))
    if not srcs:
        return None
    if force or _stale(out, srcs + hdrs):
        LIBDIR.mkdir(exist_ok=True)
        hipcc = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
        objs = []
        objdir = BUILD / "hip_obj"
        objdir.mkdir(parents=True, exist_ok=True)
        for s in srcs:
            o = objdir / (s.name + ".o")
            if force or _stale(o, [s] + hdrs):
                _run([hipcc, f"--offload-arch={GFX}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
                      "-I", CSRC / "include", "-I", CSRC / "hip", "-c", s, "-o", o])
            objs.append(o)
        _compile_atomic([hipcc, f"--offload-arch={GFX}", "-shared", "-fPIC", "-o", out, *objs,
                         f"-L{ROCM / 'lib'}", "-lamdhip64", f"-Wl,-rpath,{ROCM / 'lib'}"], out)
    return out


def build_cpp(force=False):
    mk = CSRC / "Makefile"
    if not mk.exists():
        return None
    BUILD.mkdir(exist_ok=True)
    _run(["make", "-C", CSRC, "-j8", f"BUILD={BUILD}"] + (["-B"] if force else []))
    # programs needed at run time (bench.py, perf --native, GPU tests of the C++ image client): keep
    # copies next to the shared objects, which is the directory that travels with the package
    for built, installed in (("tca_loadgen", "tca_loadgen"), ("image_client", "tca_image_client")):
        src = BUILD / "bin" / built
        if not src.exists():
            continue
        LIBDIR.mkdir(exist_ok=True)
        tmp = LIBDIR / f"{installed}.tmp{os.getpid()}"
        shutil.copy2(src, tmp)
        if shutil.which("strip"):
            subprocess.call(["strip", str(tmp)])
        os.replace(tmp, LIBDIR / installed)
    return BUILD


TARGETS = {"core": build_core, "pyext": build_pyext, "hip": build_hip, "cpp": build_cpp}


def build(targets=None, force=False):
    targets = list(targets or ["all"])
    if "all" in targets:
        targets = list(TARGETS)
    outs = {}
    for t in targets:
        outs[t] = TARGETS[t](force=force)
    return outs


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    force = "--force" in argv
    targets = [a for a in argv if not a.startswith("-")]
    outs = build(targets, force=force)
    for k, v in outs.items():
        print(f"[build] {k}: {v}")


if __name__ == "__main__":
    main()

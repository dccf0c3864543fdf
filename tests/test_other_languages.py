"""The non-Python, non-C++ clients.  The JavaScript demo RUNS against the in-repo server (node is in
the image; it falls back to node's own http2 module when @grpc/grpc-js is not installed).  There is no
JDK / Go / Scala toolchain, so those cannot be compiled here: Java sources get structural checks
(package = directory, balanced delimiters, every imported in-tree class exists)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA_ROOT = os.path.join(ROOT, "src", "java", "src", "main", "java")


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_javascript_demo_syntax_and_helpers():
    js = os.path.join(ROOT, "src", "grpc_generated", "javascript", "client.js")
    subprocess.run(["node", "--check", js], check=True, timeout=60)
    script = (
        "const m=require(%r);"
        "const r=m.buildRequest([1,2],[3,4]);"
        "if(r.model_name!=='simple'||r.inputs[1].contents.int_contents[0]!==3)process.exit(1);"
        "const b=Buffer.alloc(8);b.writeInt32LE(-5,0);b.writeInt32LE(7,4);"
        "const d=m.decodeInt32(b);if(d[0]!==-5||d[1]!==7)process.exit(2);" % js)
    subprocess.run(["node", "-e", script], check=True, timeout=60)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
@pytest.mark.parametrize("grpc_impl", ["native", "grpcio"])
def test_javascript_demo_runs_against_the_server(grpc_impl):
    """Health, metadata, one inference with typed contents and a refused call, over gRPC from node."""
    from triton_client_amd.server import Server

    js = os.path.join(ROOT, "src", "grpc_generated", "javascript", "client.js")
    with Server(models=["simple"], enable_http=False, grpc_impl=grpc_impl) as srv:
        done = subprocess.run(["node", js, srv.grpc_url], capture_output=True, text=True, timeout=60)
    assert done.returncode == 0, done.stdout[-1500:] + done.stderr[-1500:]
    lines = done.stdout.strip().splitlines()
    assert lines[0].startswith("model simple: inputs INPUT0, INPUT1") and lines[-1] == "PASS"
    assert "3 + 1 = 4    3 - 1 = 2" in lines


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_javascript_wire_coding_matches_protobuf():
    """The demo's hand-written request encoder against the real protobuf runtime."""
    from triton_client_amd.grpc import service_pb2 as pb

    js = os.path.join(ROOT, "src", "grpc_generated", "javascript", "client.js")
    script = ("const m=require(%r);const r=m.buildRequest([1,-2,300000],[4,5,6]);r.id='abc';"
              "process.stdout.write(m.encodeInferRequest(r).toString('hex'));" % js)
    out = subprocess.run(["node", "-e", script], check=True, timeout=60, capture_output=True, text=True).stdout
    msg = pb.ModelInferRequest.FromString(bytes.fromhex(out))
    assert msg.model_name == "simple" and msg.id == "abc" and [o.name for o in msg.outputs] == ["OUTPUT0", "OUTPUT1"]
    assert list(msg.inputs[0].contents.int_contents) == [1, -2, 300000] and list(msg.inputs[0].shape) == [1, 16]
    assert msg.inputs[1].name == "INPUT1" and msg.inputs[1].datatype == "INT32"


def _java_files():
    out = []
    for base, _, files in os.walk(JAVA_ROOT):
        out += [os.path.join(base, f) for f in files if f.endswith(".java")]
    return sorted(out)


def test_java_sources_are_structurally_sound():
    files = _java_files()
    assert len(files) >= 17
    classes = {os.path.relpath(f, JAVA_ROOT)[:-5].replace(os.sep, ".") for f in files}
    for f in files:
        text = open(f).read()
        rel = os.path.relpath(f, JAVA_ROOT)
        package = re.search(r"^package\s+([\w.]+);", text, re.M).group(1)
        assert package == os.path.dirname(rel).replace(os.sep, "."), rel
        assert re.search(r"\b(class|enum|interface)\s+%s\b" % os.path.basename(rel)[:-5], text), rel
        # strip strings, chars and comments, then delimiters must balance
        code = re.sub(r"'(\\.|[^'\\])'", "''", text)  # char literals first: '"' would confuse the string pass
        code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
        code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)
        code = re.sub(r"//[^\n]*", "", code)
        for open_c, close_c in ("{}", "()", "[]"):
            assert code.count(open_c) == code.count(close_c), f"{rel}: unbalanced {open_c}{close_c}"
        for imp in re.findall(r"^import\s+(tca\.[\w.]+);", text, re.M):
            assert imp in classes or imp.rsplit(".", 1)[0] in classes, f"{rel}: import {imp} does not exist"


def test_java_binary_protocol_matches_python_codec():
    """The Java packer's documented layout (4-byte LE length + UTF-8 bytes per BYTES element, LE scalars)
    is the one the Python codec produces: pin it with the Python side."""
    import numpy as np
    from triton_client_amd.utils import serialize_byte_tensor

    packed = serialize_byte_tensor(np.array(["ab", "", "xyz"], dtype=np.object_)).item()
    assert packed == b"\x02\x00\x00\x00ab\x00\x00\x00\x00\x03\x00\x00\x00xyz"
    text = open(os.path.join(JAVA_ROOT, "tca", "client", "BinaryProtocol.java")).read()
    assert "ByteOrder.LITTLE_ENDIAN" in text and "putInt(e.length)" in text


def test_java_api_build_script_options_and_refusals():
    """``src/java-api-bindings/scripts/install_dependencies_and_build.sh`` (the reference's script of that name builds
    JavaCPP bindings of the server's C API, ``src/java-api-bindings/scripts/install_dependencies_and_build.sh:27-41``;
    ours packages the client API): parses, documents its options, refuses unknown ones, and without a JDK says so
    instead of downloading one.  Parity of the build itself is unpinned here: no JDK in this image."""
    import shutil
    import subprocess

    script = os.path.join(ROOT, "src", "java-api-bindings", "scripts", "install_dependencies_and_build.sh")
    assert subprocess.run(["bash", "-n", script]).returncode == 0
    usage = subprocess.run([script, "--help"], capture_output=True, text=True)
    assert usage.returncode == 0
    for option in ("--build-home", "--maven-version", "--jar-install-path", "--offline", "--keep-build-dependencies"):
        assert option in usage.stdout
    bad = subprocess.run([script, "--no-such-option"], capture_output=True, text=True)
    assert bad.returncode == 2 and "unknown option" in bad.stderr
    if shutil.which("javac") is None:
        run = subprocess.run([script, "--offline"], capture_output=True, text=True)
        assert run.returncode == 3 and "JDK" in run.stderr

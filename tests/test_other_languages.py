"""Source-level checks of the non-Python, non-C++ clients.  There is no JDK / Go / Scala toolchain in
the development image, so these cannot be compiled here: the JavaScript demo is syntax-checked and
its pure helpers executed with node; Java sources get structural checks (package = directory,
balanced delimiters, every imported in-tree class exists)."""
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

"""Host codec: dtype tables, BYTES and BF16 wire forms (golden bytes)."""
import struct

import numpy as np
import pytest

import tritonclient.utils as utils
from triton_client_amd import _native


def test_dtype_tables_roundtrip():
    for name in ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP16", "FP32",
                 "FP64"]:
        assert utils.np_to_triton_dtype(utils.triton_to_np_dtype(name)) == name
    assert utils.triton_to_np_dtype("BF16") == np.float32
    assert utils.triton_to_np_dtype("BYTES") == np.object_
    assert utils.np_to_triton_dtype(np.object_) == "BYTES"
    assert utils.np_to_triton_dtype(np.dtype("S5")) == "BYTES"
    assert utils.np_to_triton_dtype(np.complex64) is None
    assert utils.triton_to_np_dtype("NOPE") is None


def test_bytes_golden():
    arr = np.array([[b"ab", b""], ["xyz", 7]], dtype=object)
    blob = utils.serialize_byte_tensor(arr).item()
    expect = b"".join(struct.pack("<I", len(s)) + s for s in [b"ab", b"", b"xyz", b"7"])
    assert blob == expect
    back = utils.deserialize_bytes_tensor(blob)
    assert back.dtype == np.object_ and list(back) == [b"ab", b"", b"xyz", b"7"]


def test_bytes_embedded_nul_and_np_bytes():
    arr = np.array([b"a\x00b", b"\x00"], dtype=object)
    assert list(utils.deserialize_bytes_tensor(utils.serialize_byte_tensor(arr).item())) == [b"a\x00b", b"\x00"]
    fixed = np.array([b"abc", b"de"], dtype=np.bytes_)
    assert list(utils.deserialize_bytes_tensor(utils.serialize_byte_tensor(fixed).item())) == [b"abc", b"de"]


def test_bytes_empty_and_errors():
    assert utils.serialize_byte_tensor(np.array([], dtype=object)).size == 0
    with pytest.raises(utils.InferenceServerException):
        utils.serialize_byte_tensor(np.zeros(3, dtype=np.int32))
    with pytest.raises(Exception):
        utils.deserialize_bytes_tensor(b"\x05\x00\x00\x00ab")  # length overruns


def test_serialized_byte_size():
    arr = np.array([b"ab", b"cde"], dtype=object)
    assert utils.serialized_byte_size(arr) == 5
    packed = utils.serialize_byte_tensor(arr)
    assert utils.serialized_byte_size(packed) == 5 + 8
    with pytest.raises(utils.InferenceServerException):
        utils.serialized_byte_size(np.zeros(2, dtype=np.int8))


def test_bf16_truncates_not_rounds():
    x = np.array([1.0, -2.5, 3.14159274, 1.00390625 + 2**-9], dtype=np.float32)
    blob = utils.serialize_bf16_tensor(x).item()
    ref = b"".join(struct.pack("<f", float(v))[2:4] for v in x)
    assert blob == ref
    back = utils.deserialize_bf16_tensor(blob)
    assert back.dtype == np.float32
    expect = np.frombuffer(b"".join(b"\x00\x00" + ref[i:i + 2] for i in range(0, len(ref), 2)), dtype=np.float32)
    np.testing.assert_array_equal(back, expect)
    with pytest.raises(utils.InferenceServerException):
        utils.serialize_bf16_tensor(np.zeros(2, dtype=np.float64))


def test_native_bytes_index_and_base64():
    lib = _native.core()
    blob = utils.serialize_byte_tensor(np.array([b"hello", b"", b"wor\x00ld"], dtype=object)).item()
    import ctypes

    offs = (ctypes.c_uint64 * 8)()
    lens = (ctypes.c_uint32 * 8)()
    n = lib.tca_bytes_index(blob, len(blob), -1, offs, lens, 8)
    assert n == 3 and list(lens[:3]) == [5, 0, 6] and list(offs[:3]) == [4, 13, 17]
    assert lib.tca_bytes_index(blob[:-1], len(blob) - 1, -1, offs, lens, 8) < 0
    raw = bytes(range(64))
    out = ctypes.create_string_buffer(128)
    m = lib.tca_base64_encode(raw, 64, out, 128)
    import base64

    assert out.raw[:m] == base64.b64encode(raw)
    dec = ctypes.create_string_buffer(128)
    k = lib.tca_base64_decode(out.raw[:m], m, dec, 128)
    assert dec.raw[:k] == raw


def test_exception_str():
    e = utils.InferenceServerException("boom", status="400", debug_details="d")
    assert str(e) == "[400] boom" and e.message() == "boom" and e.status() == "400" and e.debug_details() == "d"

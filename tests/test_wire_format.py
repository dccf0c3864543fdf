"""Byte-exact request/response bodies (no sockets) for both transports."""
import json
import struct

import numpy as np
import pytest

import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient
from tritonclient.utils import InferenceServerException


def _inputs(mod, binary=True):
    a = np.arange(16, dtype=np.int32).reshape(1, 16)
    i0 = mod.InferInput("INPUT0", [1, 16], "INT32")
    i1 = mod.InferInput("INPUT1", [1, 16], "INT32")
    if mod is httpclient:
        i0.set_data_from_numpy(a, binary_data=binary)
        i1.set_data_from_numpy(a, binary_data=binary)
    else:
        i0.set_data_from_numpy(a)
        i1.set_data_from_numpy(a)
    return a, [i0, i1]


def test_http_body_binary_layout():
    a, inputs = _inputs(httpclient)
    outs = [httpclient.InferRequestedOutput("OUTPUT0"), httpclient.InferRequestedOutput("OUTPUT1", binary_data=False,
                                                                                         class_count=3)]
    body, json_size = httpclient.InferenceServerClient.generate_request_body(
        inputs, outputs=outs, request_id="7", sequence_id=5, sequence_start=True, priority=2, timeout=100,
        parameters={"custom": "x"})
    assert json_size is not None
    head = json.loads(body[:json_size])
    assert body[json_size:] == a.tobytes() * 2
    assert head["id"] == "7"
    assert head["parameters"] == {"sequence_id": 5, "sequence_start": True, "sequence_end": False, "priority": 2,
                                  "timeout": 100, "custom": "x"}
    assert head["inputs"][0] == {"name": "INPUT0", "shape": [1, 16], "datatype": "INT32",
                                 "parameters": {"binary_data_size": 64}}
    assert head["outputs"][0] == {"name": "OUTPUT0", "parameters": {"binary_data": True}}
    assert head["outputs"][1] == {"name": "OUTPUT1", "parameters": {"classification": 3, "binary_data": False}}


def test_http_body_json_only_and_default_outputs():
    a, inputs = _inputs(httpclient, binary=False)
    body, json_size = httpclient.InferenceServerClient.generate_request_body(inputs)
    assert json_size is None
    head = json.loads(body)
    assert head["inputs"][0]["data"] == a.reshape(-1).tolist()
    assert head["parameters"] == {"binary_data_output": True}
    assert "outputs" not in head and "id" not in head


def test_http_reserved_parameters_and_validation():
    _, inputs = _inputs(httpclient)
    with pytest.raises(InferenceServerException, match="reserved parameter"):
        httpclient.InferenceServerClient.generate_request_body(inputs, parameters={"priority": 1})
    bad = httpclient.InferInput("X", [2], "INT32")
    with pytest.raises(InferenceServerException, match="unexpected datatype"):
        bad.set_data_from_numpy(np.zeros(2, dtype=np.float32))
    with pytest.raises(InferenceServerException, match="unexpected numpy array shape"):
        bad.set_data_from_numpy(np.zeros(3, dtype=np.int32))
    with pytest.raises(InferenceServerException, match="must be a numpy array"):
        bad.set_data_from_numpy([1, 2])
    bf = httpclient.InferInput("B", [2], "BF16")
    with pytest.raises(InferenceServerException, match="BF16 inputs must be sent as binary"):
        bf.set_data_from_numpy(np.zeros(2, dtype=np.float32), binary_data=False)
    by = httpclient.InferInput("S", [1], "BYTES")
    with pytest.raises(InferenceServerException, match="UTF-8"):
        by.set_data_from_numpy(np.array([b"\xff\xfe"], dtype=object), binary_data=False)


def test_http_shared_memory_parameters():
    i = httpclient.InferInput("INPUT0", [1, 16], "INT32")
    i.set_shared_memory("region", 64, offset=64)
    t = i._get_tensor()
    assert t["parameters"] == {"shared_memory_region": "region", "shared_memory_byte_size": 64,
                               "shared_memory_offset": 64}
    assert "data" not in t and i._get_binary_data() is None
    i.set_shared_memory("region", 64)
    assert "shared_memory_offset" not in i._get_tensor()["parameters"]
    i.set_data_from_numpy(np.zeros((1, 16), dtype=np.int32))
    assert i._get_tensor()["parameters"] == {"binary_data_size": 64}
    o = httpclient.InferRequestedOutput("OUTPUT0")
    o.set_shared_memory("r", 64)
    assert o._get_tensor()["parameters"] == {"binary_data": False, "shared_memory_region": "r",
                                             "shared_memory_byte_size": 64}
    o.unset_shared_memory()
    assert o._get_tensor()["parameters"] == {"binary_data": True}
    c = httpclient.InferRequestedOutput("OUTPUT0", class_count=2)
    with pytest.raises(InferenceServerException, match="classification"):
        c.set_shared_memory("r", 8)


def test_http_parse_response_body_mixed():
    out0 = np.arange(4, dtype=np.float32)
    strs = [b"ab", b"c"]
    blob1 = b"".join(struct.pack("<I", len(s)) + s for s in strs)
    head = {"model_name": "m", "model_version": "1", "outputs": [
        {"name": "A", "datatype": "FP32", "shape": [4], "parameters": {"binary_data_size": 16}},
        {"name": "J", "datatype": "INT32", "shape": [2], "data": [5, 6]},
        {"name": "S", "datatype": "BYTES", "shape": [2], "parameters": {"binary_data_size": len(blob1)}},
        {"name": "E", "datatype": "FP32", "shape": [0], "parameters": {"binary_data_size": 0}},
        {"name": "SHM", "datatype": "FP32", "shape": [4], "parameters": {"shared_memory_region": "r",
                                                                          "shared_memory_byte_size": 16}},
    ]}
    hj = json.dumps(head).encode()
    body = hj + out0.tobytes() + blob1
    res = httpclient.InferenceServerClient.parse_response_body(body, header_length=len(hj))
    np.testing.assert_array_equal(res.as_numpy("A"), out0)
    np.testing.assert_array_equal(res.as_numpy("J"), np.array([5, 6], dtype=np.int32))
    assert list(res.as_numpy("S")) == strs
    assert res.as_numpy("E").shape == (0,)
    assert res.as_numpy("missing") is None
    assert res.get_output("J")["data"] == [5, 6] and res.get_output("zzz") is None
    assert res.get_response()["model_name"] == "m"
    import gzip

    res2 = httpclient.InferenceServerClient.parse_response_body(gzip.compress(body), header_length=len(hj),
                                                                content_encoding="gzip")
    np.testing.assert_array_equal(res2.as_numpy("A"), out0)


def _varint(buf, pos):
    shift = value = 0
    while True:
        b = buf[pos]
        pos += 1
        value |= (b & 0x7F) << shift
        if not b & 0x80:
            return value, pos
        shift += 7


def _fields(buf):
    """Independent minimal protobuf reader: [(field_number, wire_type, value)]."""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise AssertionError(wt)
        out.append((num, wt, v))
    return out


def test_grpc_request_field_numbers():
    """Decode the serialized ModelInferRequest with a hand-rolled parser and
    check the field numbers of the public protocol."""
    from tritonclient.grpc._utils import _get_inference_request

    a, inputs = _inputs(grpcclient)
    outs = [grpcclient.InferRequestedOutput("OUTPUT0", class_count=2)]
    req = _get_inference_request("simple", inputs, "3", "rid", outs, 42, True, False, 1, 9, {"k": 1.5})
    top = _fields(req.SerializeToString())
    by_num = {}
    for num, wt, v in top:
        by_num.setdefault(num, []).append(v)
    assert by_num[1] == [b"simple"] and by_num[2] == [b"3"] and by_num[3] == [b"rid"]
    assert len(by_num[5]) == 2 and len(by_num[6]) == 1 and by_num[7] == [a.tobytes(), a.tobytes()]
    tensor = {n: v for n, _, v in _fields(by_num[5][0])}
    assert tensor[1] == b"INPUT0" and tensor[2] == b"INT32" and tensor[3] in (b"\x01\x10",)
    params = {}
    for entry in by_num[4]:
        kv = {n: v for n, _, v in _fields(entry)}
        params[kv[1]] = _fields(kv[2])
    assert params[b"sequence_id"] == [(2, 0, 42)]          # int64_param = 2
    assert params[b"sequence_start"] == [(1, 0, 1)]        # bool_param = 1
    assert params[b"priority"] == [(5, 0, 1)]              # uint64_param = 5
    assert params[b"timeout"] == [(2, 0, 9)]
    assert params[b"k"][0][0] == 4 and struct.unpack("<d", params[b"k"][0][2])[0] == 1.5  # double_param = 4
    out_t = {n: v for n, _, v in _fields(by_num[6][0])}
    assert out_t[1] == b"OUTPUT0"


def test_grpc_string_sequence_id_and_param_types():
    from tritonclient.grpc._utils import _get_inference_request

    _, inputs = _inputs(grpcclient)
    req = _get_inference_request("m", inputs, "", "", None, "uuid-1", False, True, 0, None,
                                 {"s": "v", "b": True, "i": 3})
    assert req.parameters["sequence_id"].string_param == "uuid-1"
    assert req.parameters["sequence_end"].bool_param is True
    assert req.parameters["b"].WhichOneof("parameter_choice") == "bool_param"
    assert req.parameters["i"].WhichOneof("parameter_choice") == "int64_param"
    assert req.parameters["s"].string_param == "v"
    with pytest.raises(InferenceServerException, match="not supported"):
        _get_inference_request("m", inputs, "", "", None, 0, False, False, 0, None, {"x": [1]})
    with pytest.raises(InferenceServerException, match="reserved"):
        _get_inference_request("m", inputs, "", "", None, 0, False, False, 0, None, {"sequence_id": 1})


def test_grpc_infer_result_positional_and_shm_mix():
    from tritonclient.grpc import service_pb2

    resp = service_pb2.ModelInferResponse(model_name="m")
    o = resp.outputs.add(name="SHM", datatype="FP32", shape=[2])
    o.parameters["shared_memory_region"].string_param = "r"
    resp.outputs.add(name="A", datatype="INT32", shape=[2])
    resp.outputs.add(name="S", datatype="BYTES", shape=[1])
    resp.raw_output_contents.append(np.array([1, 2], dtype=np.int32).tobytes())
    resp.raw_output_contents.append(struct.pack("<I", 2) + b"hi")
    res = grpcclient.InferResult(resp)
    np.testing.assert_array_equal(res.as_numpy("A"), [1, 2])
    assert list(res.as_numpy("S")) == [b"hi"]
    assert res.as_numpy("SHM") is None  # lives in the caller's region
    assert res.get_output("A", as_json=True)["name"] == "A"
    assert res.get_response(as_json=True)["model_name"] == "m"
    assert res.as_numpy("nope") is None


def test_protoc_subset_compiler():
    from triton_client_amd.grpc import _protoc

    files = _protoc.compile_protos({"a.proto": """
        syntax = "proto3"; package p; /* block */
        enum E { Z = 0; ONE = 1; }
        message M { message N { int32 v = 1; } repeated N items = 2; map<string, E> tags = 3;
                    oneof c { bool b = 4; string s = 5; } E e = 6; }
        service S { rpc Do(M) returns (stream M) {} }
    """})
    fd = files[0]
    m = fd.message_type[0]
    assert [f.name for f in m.field] == ["items", "tags", "b", "s", "e"]
    assert m.field[0].type_name == ".p.M.N" and m.field[1].type_name == ".p.M.TagsEntry"
    assert m.nested_type[1].options.map_entry and m.field[2].oneof_index == 0
    assert fd.service[0].method[0].server_streaming and not fd.service[0].method[0].client_streaming

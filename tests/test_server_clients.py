"""Typed client suite over {HTTP, gRPC} against the in-repo server: the
behaviours the reference's examples assert (they double as its integration
tests) plus the C++ gtest cases that have a Python equivalent."""
import json
import queue
import threading
import time
from functools import partial

import numpy as np
import pytest

from tritonclient.utils import InferenceServerException
import tritonclient.utils.shared_memory as shm


def _simple_inputs(mod, batch=1, binary=True):
    a = np.arange(16 * batch, dtype=np.int32).reshape(batch, 16)
    b = np.ones((batch, 16), dtype=np.int32)
    i0 = mod.InferInput("INPUT0", [batch, 16], "INT32")
    i1 = mod.InferInput("INPUT1", [batch, 16], "INT32")
    if mod.__name__.endswith("http"):
        i0.set_data_from_numpy(a, binary_data=binary)
        i1.set_data_from_numpy(b, binary_data=binary)
    else:
        i0.set_data_from_numpy(a)
        i1.set_data_from_numpy(b)
    return a, b, [i0, i1]


def test_health_and_metadata(client, proto):
    mod, c = client
    assert c.is_server_live() and c.is_server_ready()
    assert c.is_model_ready("simple") and c.is_model_ready("simple", "1")
    assert not c.is_model_ready("does_not_exist")
    if proto == "http":
        md = c.get_server_metadata()
        assert md["name"] == "triton" and "binary_tensor_data" in md["extensions"]
        mm = c.get_model_metadata("simple")
        assert [t["name"] for t in mm["inputs"]] == ["INPUT0", "INPUT1"] and mm["inputs"][0]["shape"] == [-1, 16]
        cfg = c.get_model_config("simple")
        assert cfg["max_batch_size"] == 8 and cfg["input"][0]["data_type"] == "TYPE_INT32"
    else:
        md = c.get_server_metadata()
        assert md.name == "triton"
        assert c.get_server_metadata(as_json=True)["name"] == "triton"
        mm = c.get_model_metadata("simple")
        assert [t.name for t in mm.inputs] == ["INPUT0", "INPUT1"] and list(mm.inputs[0].shape) == [-1, 16]
        cfg = c.get_model_config("simple").config
        assert cfg.max_batch_size == 8 and cfg.name == "simple"
        from tritonclient.grpc import model_config_pb2 as mc

        assert cfg.input[0].data_type == mc.TYPE_INT32
        assert c.get_model_config("simple", as_json=True)["config"]["name"] == "simple"
    with pytest.raises(InferenceServerException, match="Request for unknown model"):
        c.get_model_metadata("wrong_model_name")
    with pytest.raises(InferenceServerException, match="model version must be a string"):
        c.is_model_ready("simple", 1)


def test_simple_infer_variants(client, proto):
    mod, c = client
    a, b, inputs = _simple_inputs(mod)
    outs = [mod.InferRequestedOutput("OUTPUT0"), mod.InferRequestedOutput("OUTPUT1")]
    r = c.infer("simple", inputs, outputs=outs, request_id="abc")
    np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
    np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
    if proto == "http":
        assert r.get_response()["id"] == "abc"
    else:
        assert r.get_response().id == "abc"
    # no outputs requested -> all outputs come back
    r = c.infer("simple", inputs)
    np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
    np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
    # only one output
    r = c.infer("simple", inputs, outputs=[mod.InferRequestedOutput("OUTPUT1")])
    assert r.as_numpy("OUTPUT0") is None
    np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
    # batch > 1
    a4, b4, in4 = _simple_inputs(mod, batch=4)
    np.testing.assert_array_equal(c.infer("simple", in4).as_numpy("OUTPUT0"), a4 + b4)
    # headers + custom parameters are accepted
    c.infer("simple", inputs, headers={"my-header": "1"}, parameters={"flag": True, "n": 3, "s": "x"})


def test_http_json_and_mixed_tensors(server):
    import tritonclient.http as mod

    with mod.InferenceServerClient(server.http_url) as c:
        a, b, inputs = _simple_inputs(mod, binary=False)
        outs = [mod.InferRequestedOutput("OUTPUT0", binary_data=False), mod.InferRequestedOutput("OUTPUT1")]
        r = c.infer("simple", inputs, outputs=outs, query_params={"test_1": 1, "test_2": [2, 3]})
        assert "data" in r.get_output("OUTPUT0") and "data" not in r.get_output("OUTPUT1")
        np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
        np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
        for req_alg, resp_alg in (("gzip", "gzip"), ("deflate", "deflate"), ("gzip", None), (None, "deflate")):
            _, _, inputs = _simple_inputs(mod)
            r = c.infer("simple", inputs, request_compression_algorithm=req_alg,
                        response_compression_algorithm=resp_alg)
            np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
        with pytest.raises(InferenceServerException, match="Transfer-Encoding"):
            c.infer("simple", inputs, headers={"Transfer-Encoding": "chunked"})


def test_grpc_compression_and_typed_contents(server):
    import tritonclient.grpc as mod
    from tritonclient.grpc import service_pb2, service_pb2_grpc
    import grpc

    with mod.InferenceServerClient(server.grpc_url) as c:
        a, b, inputs = _simple_inputs(mod)
        for alg in ("gzip", "deflate", "bogus"):
            np.testing.assert_array_equal(c.infer("simple", inputs, compression_algorithm=alg).as_numpy("OUTPUT0"),
                                          a + b)
    # raw stub with typed contents (grpc_explicit_int_content_client intent)
    channel = grpc.insecure_channel(server.grpc_url)
    stub = service_pb2_grpc.GRPCInferenceServiceStub(channel)
    req = service_pb2.ModelInferRequest(model_name="simple")
    for name in ("INPUT0", "INPUT1"):
        t = req.inputs.add(name=name, datatype="INT32", shape=[1, 16])
        t.contents.int_contents.extend(range(16))
    resp = stub.ModelInfer(req)
    out0 = np.frombuffer(resp.raw_output_contents[0], dtype=np.int32)
    np.testing.assert_array_equal(out0, np.arange(16) * 2)
    req.raw_input_contents.append(np.arange(16, dtype=np.int32).tobytes())
    req.raw_input_contents.append(np.arange(16, dtype=np.int32).tobytes())
    with pytest.raises(grpc.RpcError) as ei:
        stub.ModelInfer(req)
    assert "contents field must not be specified when using raw_input_contents for 'INPUT0' for model 'simple'" \
        in ei.value.details()
    channel.close()


def test_error_reporting(client, proto):
    mod, c = client
    a, b, inputs = _simple_inputs(mod)
    with pytest.raises(InferenceServerException, match="Request for unknown model") as ei:
        c.infer("wrong_model_name", inputs)
    assert ei.value.status() in ("400", "StatusCode.INVALID_ARGUMENT")
    with pytest.raises(InferenceServerException, match="unexpected inference output"):
        c.infer("simple", inputs, outputs=[mod.InferRequestedOutput("NOPE")])
    with pytest.raises(InferenceServerException, match="expected 2 inputs but got 1"):
        c.infer("simple", inputs[:1])
    bad = mod.InferInput("INPUT0", [1, 8], "INT32")
    bad.set_data_from_numpy(np.zeros((1, 8), dtype=np.int32))
    with pytest.raises(InferenceServerException, match="unexpected shape"):
        c.infer("simple", [bad, inputs[1]])
    big = np.zeros((9, 16), dtype=np.int32)
    i0 = mod.InferInput("INPUT0", [9, 16], "INT32").set_data_from_numpy(big)
    i1 = mod.InferInput("INPUT1", [9, 16], "INT32").set_data_from_numpy(big)
    with pytest.raises(InferenceServerException, match="batch-size must be <= 8"):
        c.infer("simple", [i0, i1])
    with pytest.raises(InferenceServerException, match="model version must be a string"):
        c.infer("simple", inputs, model_version=1)


def test_string_models(client, proto):
    mod, c = client
    a = np.arange(16, dtype=np.int32).reshape(1, 16)
    in0 = np.array([str(x).encode() for x in a.reshape(-1)], dtype=object).reshape(1, 16)
    in1 = np.array([b"1"] * 16, dtype=object).reshape(1, 16)
    i0 = mod.InferInput("INPUT0", [1, 16], "BYTES").set_data_from_numpy(in0)
    i1 = mod.InferInput("INPUT1", [1, 16], "BYTES").set_data_from_numpy(in1)
    r = c.infer("simple_string", [i0, i1])
    out0 = r.as_numpy("OUTPUT0")
    assert [int(x) for x in out0.reshape(-1)] == list(range(1, 17))
    assert [int(x) for x in r.as_numpy("OUTPUT1").reshape(-1)] == list(range(-1, 15))
    # identity with embedded NULs and np.bytes_ input
    data = np.array([[b"a\x00b", b"", b"\x00\x00", b"xyz"]], dtype=object)
    ii = mod.InferInput("INPUT0", [1, 4], "BYTES").set_data_from_numpy(data)
    got = c.infer("simple_identity", [ii]).as_numpy("OUTPUT0")
    assert got.tolist() == data.tolist()
    fixed = np.array([[b"hello", b"wor"]], dtype=np.bytes_)
    ii = mod.InferInput("INPUT0", [1, 2], "BYTES").set_data_from_numpy(fixed)
    assert c.infer("simple_identity", [ii]).as_numpy("OUTPUT0").tolist() == [[b"hello", b"wor"]]


def test_versioned_model(client):
    mod, c = client
    a, b, inputs = _simple_inputs(mod)
    r1 = c.infer("onnx_int32_int32_int32", inputs, model_version="1")
    r2 = c.infer("onnx_int32_int32_int32", inputs, model_version="2")
    np.testing.assert_array_equal(r1.as_numpy("OUTPUT0"), a + b)
    np.testing.assert_array_equal(r2.as_numpy("OUTPUT0"), a - b)
    np.testing.assert_array_equal(r2.as_numpy("OUTPUT1"), a + b)
    with pytest.raises(InferenceServerException, match="unknown model"):
        c.infer("onnx_int32_int32_int32", inputs, model_version="9")


def _seq(mod, c, model, seq_id, values):
    out = []
    for i, v in enumerate(values):
        x = np.full((1, 1), v, dtype=np.int32)
        inp = mod.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(x)
        r = c.infer(model, [inp], sequence_id=seq_id, sequence_start=(i == 0), sequence_end=(i == len(values) - 1))
        out.append(int(r.as_numpy("OUTPUT")[0][0]))
    return out


def test_sequence_sync(client):
    mod, c = client
    values = [11, 7, 5, 3, 2, 0, 1]
    assert _seq(mod, c, "simple_sequence", 1000, [0] + values) == [1] + values
    assert _seq(mod, c, "simple_sequence", 1001, [100] + [-v for v in values]) == [101] + [-v for v in values]
    assert _seq(mod, c, "simple_sequence", "a-uuid", [20] + [-v for v in values]) == [21] + [-v for v in values]
    dyn = _seq(mod, c, "simple_dyna_sequence", 1000, [0] + values)
    assert dyn == [1] + values[:-1] + [values[-1] + 1000]
    sdyn = _seq(mod, c, "simple_string_dyna_sequence", "1002", [20] + [-v for v in values])
    expect, acc = [], 0
    for i, v in enumerate([20] + [-v for v in values]):
        acc = v if i == 0 else acc + v
        expect.append(acc)
    expect[-1] += 1002
    assert sdyn == expect
    inp = mod.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(np.zeros((1, 1), dtype=np.int32))
    with pytest.raises(InferenceServerException, match="correlation ID"):
        c.infer("simple_sequence", [inp])
    with pytest.raises(InferenceServerException, match="START"):
        c.infer("simple_sequence", [inp], sequence_id=777777)


def test_async_infer(client, proto):
    mod, c = client
    a, b, inputs = _simple_inputs(mod)
    if proto == "http":
        handles = [c.async_infer("simple", inputs) for _ in range(16)]
        for h in handles:
            np.testing.assert_array_equal(h.get_result().as_numpy("OUTPUT0"), a + b)
        bad = c.async_infer("wrong_model_name", inputs)
        with pytest.raises(InferenceServerException, match="unknown model"):
            bad.get_result()
    else:
        done = queue.Queue()

        def cb(result, error):
            done.put((result, error))

        for _ in range(16):
            c.async_infer("simple", inputs, cb)
        for _ in range(16):
            result, error = done.get(timeout=10)
            assert error is None
            np.testing.assert_array_equal(result.as_numpy("OUTPUT0"), a + b)
        c.async_infer("wrong_model_name", inputs, cb)
        result, error = done.get(timeout=10)
        assert result is None and "unknown model" in str(error)


def test_grpc_stream_sequences_and_decoupled(server):
    import tritonclient.grpc as mod

    results = queue.Queue()

    def cb(q, result, error):
        q.put(error if error is not None else result)

    with mod.InferenceServerClient(server.grpc_url) as c:
        c.start_stream(callback=partial(cb, results))
        with pytest.raises(InferenceServerException, match="single active stream"):
            c.start_stream(callback=partial(cb, results))
        values = [11, 7, 5, 3, 2, 0, 1]
        # two interleaved sequences on one stream
        for i, v in enumerate([0] + values):
            for sid, sign, first in ((2000, 1, 0), (2001, -1, 100)):
                val = first if i == 0 else sign * v
                inp = mod.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(np.full((1, 1), val, np.int32))
                c.async_stream_infer("simple_sequence", [inp], request_id=f"{sid}_{i}", sequence_id=sid,
                                     sequence_start=(i == 0), sequence_end=(i == len(values)))
        got = {2000: [], 2001: []}
        for _ in range(2 * (len(values) + 1)):
            r = results.get(timeout=10)
            assert not isinstance(r, Exception), r
            sid = int(r.get_response().id.split("_")[0])
            got[sid].append(int(r.as_numpy("OUTPUT")[0][0]))
        assert got[2000] == [1] + values and got[2001] == [101] + [-v for v in values]
        # decoupled: 1 request -> N responses
        n = 6
        i_in = mod.InferInput("IN", [n], "INT32").set_data_from_numpy(np.arange(100, 100 + n, dtype=np.int32))
        i_d = mod.InferInput("DELAY", [n], "UINT32").set_data_from_numpy(np.full(n, 5, dtype=np.uint32))
        i_w = mod.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(np.array([1], dtype=np.uint32))
        c.async_stream_infer("repeat_int32", [i_in, i_d, i_w], outputs=[mod.InferRequestedOutput("OUT")],
                             enable_empty_final_response=True)
        outs, finals = [], 0
        while finals == 0:
            r = results.get(timeout=10)
            assert not isinstance(r, Exception), r
            resp = r.get_response()
            if resp.parameters["triton_final_response"].bool_param:
                finals += 1
                assert len(resp.outputs) == 0
            else:
                outs.append(int(r.as_numpy("OUT")[0]))
        assert outs == list(range(100, 100 + n))
        # error on the stream is delivered through the callback, stream stays usable
        c.async_stream_infer("wrong_model_name", [i_in])
        err = results.get(timeout=10)
        assert isinstance(err, InferenceServerException) and "unknown model" in str(err)
        c.stop_stream()
        with pytest.raises(InferenceServerException, match="stream not available"):
            c.async_stream_infer("simple_sequence", [i_in])
    # decoupled models are refused on the unary path
    with mod.InferenceServerClient(server.grpc_url) as c:
        with pytest.raises(InferenceServerException, match="decoupled"):
            c.infer("repeat_int32", [i_in, i_d, i_w])


def test_system_shared_memory(client, proto):
    mod, c = client
    c.unregister_system_shared_memory()
    a = np.arange(16, dtype=np.int32)
    b = np.ones(16, dtype=np.int32)
    ip = shm.create_shared_memory_region(f"input_data_{proto}", f"/tca_in_{proto}", 128)
    op = shm.create_shared_memory_region(f"output_data_{proto}", f"/tca_out_{proto}", 128)
    try:
        shm.set_shared_memory_region(ip, [a])
        shm.set_shared_memory_region(ip, [b], offset=64)
        c.register_system_shared_memory(f"input_data_{proto}", f"/tca_in_{proto}", 128)
        c.register_system_shared_memory(f"output_data_{proto}", f"/tca_out_{proto}", 128)
        with pytest.raises(InferenceServerException, match="already in manager"):
            c.register_system_shared_memory(f"input_data_{proto}", f"/tca_in_{proto}", 128)
        status = c.get_system_shared_memory_status()
        if proto == "http":
            assert {r["name"] for r in status} == {f"input_data_{proto}", f"output_data_{proto}"}
        else:
            assert set(status.regions.keys()) == {f"input_data_{proto}", f"output_data_{proto}"}
            assert status.regions[f"input_data_{proto}"].byte_size == 128
        i0 = mod.InferInput("INPUT0", [1, 16], "INT32")
        i0.set_shared_memory(f"input_data_{proto}", 64)
        i1 = mod.InferInput("INPUT1", [1, 16], "INT32")
        i1.set_shared_memory(f"input_data_{proto}", 64, offset=64)
        o0 = mod.InferRequestedOutput("OUTPUT0")
        o0.set_shared_memory(f"output_data_{proto}", 64)
        o1 = mod.InferRequestedOutput("OUTPUT1")
        o1.set_shared_memory(f"output_data_{proto}", 64, offset=64)
        r = c.infer("simple", [i0, i1], outputs=[o0, o1])
        assert r.as_numpy("OUTPUT0") is None
        out0 = r.get_output("OUTPUT0")
        assert out0 is not None
        np.testing.assert_array_equal(shm.get_contents_as_numpy(op, np.int32, [1, 16]), (a + b).reshape(1, 16))
        np.testing.assert_array_equal(shm.get_contents_as_numpy(op, np.int32, [1, 16], offset=64),
                                      (a - b).reshape(1, 16))
        # reuse the same objects without shm (reuse_infer_objects_client intent)
        i0.set_data_from_numpy(a.reshape(1, 16))
        i1.set_data_from_numpy(b.reshape(1, 16))
        o0.unset_shared_memory()
        o1.unset_shared_memory()
        r = c.infer("simple", [i0, i1], outputs=[o0, o1])
        np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), (a + b).reshape(1, 16))
        # too-small output region is reported
        o0.set_shared_memory(f"output_data_{proto}", 8)
        with pytest.raises(InferenceServerException, match="should be at least 64 bytes"):
            c.infer("simple", [i0, i1], outputs=[o0])
        c.unregister_system_shared_memory(f"input_data_{proto}")
        i0.set_shared_memory(f"input_data_{proto}", 64)
        with pytest.raises(InferenceServerException, match="Unable to find shared memory region"):
            c.infer("simple", [i0, i1])
    finally:
        c.unregister_system_shared_memory()
        shm.destroy_shared_memory_region(ip)
        shm.destroy_shared_memory_region(op)
    assert f"/tca_in_{proto}" not in shm.mapped_shared_memory_regions()


def test_shm_string_tensors(client, proto):
    import tritonclient.utils as utils

    mod, c = client
    c.unregister_system_shared_memory()
    in0 = np.array([str(x).encode() for x in range(16)], dtype=object).reshape(1, 16)
    in1 = np.array([b"1"] * 16, dtype=object).reshape(1, 16)
    s0, s1 = utils.serialize_byte_tensor(in0), utils.serialize_byte_tensor(in1)
    n0, n1 = utils.serialized_byte_size(s0), utils.serialized_byte_size(s1)
    ip = shm.create_shared_memory_region("sin", f"/tca_sin_{proto}", n0 + n1)
    op = shm.create_shared_memory_region("sout", f"/tca_sout_{proto}", 512)
    try:
        shm.set_shared_memory_region(ip, [s0, s1])
        c.register_system_shared_memory("sin", f"/tca_sin_{proto}", n0 + n1)
        c.register_system_shared_memory("sout", f"/tca_sout_{proto}", 512)
        i0 = mod.InferInput("INPUT0", [1, 16], "BYTES").set_shared_memory("sin", n0)
        i1 = mod.InferInput("INPUT1", [1, 16], "BYTES").set_shared_memory("sin", n1, offset=n0)
        o0 = mod.InferRequestedOutput("OUTPUT0")
        o0.set_shared_memory("sout", 256)
        o1 = mod.InferRequestedOutput("OUTPUT1")
        o1.set_shared_memory("sout", 256, offset=256)
        c.infer("simple_string", [i0, i1], outputs=[o0, o1])
        out0 = shm.get_contents_as_numpy(op, np.object_, [1, 16])
        out1 = shm.get_contents_as_numpy(op, np.object_, [1, 16], offset=256)
        assert [int(x) for x in out0.reshape(-1)] == list(range(1, 17))
        assert [int(x) for x in out1.reshape(-1)] == list(range(-1, 15))
    finally:
        c.unregister_system_shared_memory()
        shm.destroy_shared_memory_region(ip)
        shm.destroy_shared_memory_region(op)


def test_model_control(client, proto):
    mod, c = client
    index = c.get_model_repository_index()
    names = [m["name"] for m in index] if proto == "http" else [m.name for m in index.models]
    assert "simple" in names and len(names) >= 8
    c.unload_model("simple")
    assert not c.is_model_ready("simple")
    c.load_model("simple")
    assert c.is_model_ready("simple")
    # config override (max_batch_size 8 -> 16)
    c.load_model("simple", config=json.dumps({"max_batch_size": 16, "backend": "python"}))
    cfg = c.get_model_config("simple")
    mbs = cfg["max_batch_size"] if proto == "http" else cfg.config.max_batch_size
    assert mbs == 16
    c.load_model("simple", config=json.dumps({"max_batch_size": 8}), files={"file:1/model.bin": b"\x00\x01binary"})
    with pytest.raises(InferenceServerException, match="failed to load"):
        c.load_model("wrong_model_name")
    with pytest.raises(InferenceServerException, match="failed to load"):
        c.load_model("simple", files={"file:1/model.bin": b"x"})  # files need a config
    c.load_model("simple")
    assert c.is_model_ready("simple")


def test_statistics_trace_log(client, proto):
    mod, c = client
    a, b, inputs = _simple_inputs(mod)
    before = c.get_inference_statistics("simple")
    c.infer("simple", inputs)
    after = c.get_inference_statistics("simple")
    if proto == "http":
        cnt = lambda s: s["model_stats"][0]["inference_stats"]["success"]["count"]  # noqa: E731
        assert after["model_stats"][0]["name"] == "simple"
        allstats = c.get_inference_statistics()
        assert len(allstats["model_stats"]) >= 8
    else:
        cnt = lambda s: s.model_stats[0].inference_stats.success.count  # noqa: E731
        assert after.model_stats[0].name == "simple"
        assert c.get_inference_statistics(as_json=True)["model_stats"]
    assert cnt(after) == cnt(before) + 1
    # trace settings: update, per-model override, clear
    ts = c.update_trace_settings(settings={"trace_rate": "1", "trace_level": ["TIMESTAMPS"]})
    get = (lambda t, k: t[k]) if proto == "http" else (lambda t, k: list(t.settings[k].value))
    assert get(ts, "trace_rate") == ["1"] and get(ts, "trace_level") == ["TIMESTAMPS"]
    ms = c.update_trace_settings(model_name="simple", settings={"trace_rate": "5"})
    assert get(ms, "trace_rate") == ["5"] and get(ms, "trace_level") == ["TIMESTAMPS"]
    assert get(c.get_trace_settings(), "trace_rate") == ["1"]
    cleared = c.update_trace_settings(model_name="simple", settings={"trace_rate": None})
    assert get(cleared, "trace_rate") == ["1"]
    c.update_trace_settings(settings={"trace_rate": "1000", "trace_level": ["OFF"]})
    with pytest.raises(InferenceServerException):
        c.update_trace_settings(settings={"no_such_setting": "1"})
    # log settings
    ls = c.update_log_settings({"log_verbose_level": 2, "log_info": False})
    if proto == "http":
        assert ls["log_verbose_level"] == 2 and ls["log_info"] is False
        assert c.get_log_settings()["log_verbose_level"] == 2
    else:
        assert ls.settings["log_verbose_level"].uint32_param == 2 and ls.settings["log_info"].bool_param is False
        assert c.get_log_settings(as_json=True)["settings"]["log_verbose_level"]["uint32_param"] == 2
    c.update_log_settings({"log_verbose_level": 0, "log_info": True})


def test_plugin_and_basic_auth(client, proto):
    mod, c = client
    from tritonclient.http.auth import BasicAuth

    seen = {}

    class Recorder(mod.InferenceServerClientPlugin):
        def __call__(self, request):
            request.headers["x-test"] = "1"
            seen.update(request.headers)

    c.register_plugin(Recorder())
    with pytest.raises(InferenceServerException, match="already registered"):
        c.register_plugin(Recorder())
    assert c.is_server_live()
    assert seen.get("x-test") == "1"
    c.unregister_plugin()
    with pytest.raises(InferenceServerException, match="No plugin"):
        c.unregister_plugin()
    c.register_plugin(BasicAuth("user", "pass"))
    assert c.plugin() is not None and c.is_server_live()
    c.unregister_plugin()


def test_client_timeout(server):
    import tritonclient.grpc as g
    import tritonclient.http as h

    x = np.zeros((1, 4), dtype=np.int32)
    with g.InferenceServerClient(server.grpc_url) as c:
        c.load_model("custom_identity_int32", config=json.dumps({"parameters": {"execute_delay_ms": {
            "string_value": "400"}}}))
        inp = g.InferInput("INPUT0", [1, 4], "INT32").set_data_from_numpy(x)
        with pytest.raises(InferenceServerException, match="Deadline Exceeded"):
            c.infer("custom_identity_int32", [inp], client_timeout=0.1)
        np.testing.assert_array_equal(c.infer("custom_identity_int32", [inp], client_timeout=5).as_numpy("OUTPUT0"),
                                      x)
    with h.InferenceServerClient(server.http_url, network_timeout=0.1) as c:
        inp = h.InferInput("INPUT0", [1, 4], "INT32").set_data_from_numpy(x)
        with pytest.raises(InferenceServerException, match="Deadline Exceeded"):
            c.infer("custom_identity_int32", [inp])
    with g.InferenceServerClient(server.grpc_url) as c:
        c.load_model("custom_identity_int32")


def test_grpc_cancel_and_keepalive(server):
    import tritonclient.grpc as g

    ka = g.KeepAliveOptions(keepalive_time_ms=2**31 - 1, keepalive_timeout_ms=20000,
                            keepalive_permit_without_calls=False, http2_max_pings_without_data=2)
    with g.InferenceServerClient(server.grpc_url, keepalive_options=ka) as c:
        assert c.is_server_live()
    with g.InferenceServerClient(server.grpc_url, channel_args=[("grpc.max_send_message_length", 1 << 20)]) as c:
        assert c.is_server_live()
        c.load_model("custom_identity_int32", config='{"parameters": {"execute_delay_ms": {"string_value": "300"}}}')
        done = queue.Queue()
        x = np.zeros((1, 4), dtype=np.int32)
        inp = g.InferInput("INPUT0", [1, 4], "INT32").set_data_from_numpy(x)
        ctx = c.async_infer("custom_identity_int32", [inp], lambda result, error: done.put((result, error)))
        ctx.cancel()
        result, error = done.get(timeout=5)
        assert result is None and error.status() == "StatusCode.CANCELLED"
        c.load_model("custom_identity_int32")


def test_memory_growth_new_client_per_request(server):
    """memory_growth_test intent: many short-lived clients must not leak sockets/threads."""
    import tritonclient.http as h

    before = threading.active_count()
    x = np.zeros((1, 16), dtype=np.int32)
    for _ in range(50):
        with h.InferenceServerClient(server.http_url) as c:
            inp = h.InferInput("INPUT0", [1, 16], "INT32").set_data_from_numpy(x)
            c.infer("custom_identity_int32", [inp])
    time.sleep(0.1)
    assert threading.active_count() <= before + 2


@pytest.mark.parametrize("impl", ["native", "grpcio"])
def test_both_grpc_fronts_serve_the_same_surface(impl):
    """The native HTTP/2 front and the stock Python gRPC server are
    interchangeable behind ``Server(grpc_impl=...)``."""
    import queue

    import tritonclient.grpc as grpcclient
    from triton_client_amd.server import Server
    from tritonclient.utils import InferenceServerException

    with Server(models=["simple", "simple_sequence", "repeat_int32"], enable_http=False, grpc_impl=impl) as srv:
        with grpcclient.InferenceServerClient(srv.grpc_url) as c:
            assert c.is_server_live() and c.is_model_ready("simple")
            assert c.get_model_config("simple").config.name == "simple"
            a = np.arange(16, dtype=np.int32).reshape(1, 16)
            b = np.full((1, 16), 3, dtype=np.int32)
            inputs = [grpcclient.InferInput("INPUT0", [1, 16], "INT32").set_data_from_numpy(a),
                      grpcclient.InferInput("INPUT1", [1, 16], "INT32").set_data_from_numpy(b)]
            r = c.infer("simple", inputs, compression_algorithm="gzip")
            np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
            with pytest.raises(InferenceServerException, match="unknown model"):
                c.infer("nope", inputs)
            # async + stream (decoupled: one request, five responses + the empty final one)
            done = queue.Queue()
            c.async_infer("simple", inputs, lambda result, error: done.put((result, error)))
            result, error = done.get(timeout=20)
            assert error is None
            np.testing.assert_array_equal(result.as_numpy("OUTPUT1"), a - b)
            c.start_stream(lambda result, error: done.put((result, error)))
            n = 5
            c.async_stream_infer("repeat_int32", [
                grpcclient.InferInput("IN", [n], "INT32").set_data_from_numpy(np.arange(n, dtype=np.int32)),
                grpcclient.InferInput("DELAY", [n], "UINT32").set_data_from_numpy(np.zeros(n, dtype=np.uint32)),
                grpcclient.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(np.zeros(1, dtype=np.uint32)),
            ], enable_empty_final_response=True)
            got = []
            while True:
                result, error = done.get(timeout=20)
                assert error is None
                if result.get_response().parameters["triton_final_response"].bool_param:
                    break
                got.append(int(result.as_numpy("OUT")[0]))
            c.stop_stream()
            assert got == list(range(n))


def test_large_wire_tensors_take_the_zero_copy_path():
    """Requests above the native front's zero-copy threshold: tensors stay in the
    transport buffer and are aliased by numpy; answers must be bit-identical,
    also for several inputs, BYTES data and a malformed message."""
    import tritonclient.grpc as g
    from triton_client_amd.server import Server

    with Server(models=["simple", "custom_identity_int32", "simple_identity"], grpc_impl="native") as srv:
        with g.InferenceServerClient(srv.grpc_url) as c:
            x = np.arange(200_000, dtype=np.int32).reshape(1, -1)  # 800 KB
            inp = g.InferInput("INPUT0", list(x.shape), "INT32").set_data_from_numpy(x)
            for _ in range(3):
                out = c.infer("custom_identity_int32", [inp]).as_numpy("OUTPUT0")
                np.testing.assert_array_equal(out, x)
            # BYTES payload above the threshold
            s = np.array([[b"x" * 300_000, b"tail"]], dtype=np.object_)
            sin = g.InferInput("INPUT0", [1, 2], "BYTES").set_data_from_numpy(s)
            got = c.infer("simple_identity", [sin]).as_numpy("OUTPUT0")
            assert got.shape == (1, 2) and got[0, 0] == s[0, 0] and got[0, 1] == b"tail"
            # wrong byte count is still reported cleanly
            bad = g.InferInput("INPUT0", [1, 16], "INT32")
            bad._raw_content = b"\0" * 200_000
            with pytest.raises(InferenceServerException, match="byte size"):
                c.infer("custom_identity_int32", [bad])


@pytest.mark.parametrize("impl", ["native", "asyncio"])
def test_both_http_fronts_serve_the_same_surface(impl):
    """The native HTTP/1.1 front and the asyncio one are interchangeable behind
    ``Server(http_impl=...)``: control endpoints, JSON and binary inference, compression in
    both directions, errors, asynchronous requests, large tensors."""
    import tritonclient.http as httpclient
    from triton_client_amd.server import Server
    from tritonclient.utils import InferenceServerException

    with Server(models=["simple", "simple_string", "custom_identity_int32"], enable_grpc=False, http_impl=impl,
                model_control="explicit") as srv:
        with httpclient.InferenceServerClient(srv.http_url, concurrency=4) as c:
            assert c.is_server_live() and c.is_server_ready()
            c.load_model("simple")
            c.load_model("custom_identity_int32")
            assert c.is_model_ready("simple") and not c.is_model_ready("simple_string")
            assert c.get_model_config("simple")["name"] == "simple"
            assert c.get_model_metadata("simple")["inputs"][0]["name"] == "INPUT0"
            assert {m["name"] for m in c.get_model_repository_index()} >= {"simple", "simple_string"}
            a = np.arange(16, dtype=np.int32).reshape(1, 16)
            b = np.full((1, 16), 3, dtype=np.int32)
            for binary in (True, False):
                inputs = [httpclient.InferInput("INPUT0", [1, 16], "INT32").set_data_from_numpy(a, binary_data=binary),
                          httpclient.InferInput("INPUT1", [1, 16], "INT32").set_data_from_numpy(b, binary_data=binary)]
                outputs = [httpclient.InferRequestedOutput("OUTPUT0", binary_data=binary),
                           httpclient.InferRequestedOutput("OUTPUT1", binary_data=not binary)]
                r = c.infer("simple", inputs, outputs=outputs, request_id="7")
                np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
                np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
                assert r.get_response()["id"] == "7"
            for algo in ("gzip", "deflate"):
                r = c.infer("simple", inputs, request_compression_algorithm=algo, response_compression_algorithm=algo)
                np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
            with pytest.raises(InferenceServerException, match="unknown model"):
                c.infer("nope", inputs)
            handles = [c.async_infer("simple", inputs) for _ in range(16)]
            for h in handles:
                np.testing.assert_array_equal(h.get_result().as_numpy("OUTPUT1"), a - b)
            x = np.arange(300_000, dtype=np.int32).reshape(1, -1)  # 1.2 MB: above the zero-copy threshold
            inp = httpclient.InferInput("INPUT0", list(x.shape), "INT32").set_data_from_numpy(x)
            np.testing.assert_array_equal(c.infer("custom_identity_int32", [inp]).as_numpy("OUTPUT0"), x)
            x = np.arange(3_000_000, dtype=np.int32).reshape(1, -1)  # 12 MB: the body buffer grows while it arrives
            inp = httpclient.InferInput("INPUT0", list(x.shape), "INT32").set_data_from_numpy(x)
            np.testing.assert_array_equal(c.infer("custom_identity_int32", [inp]).as_numpy("OUTPUT0"), x)
            stats = c.get_inference_statistics("simple")["model_stats"][0]
            assert stats["inference_count"] >= 20
            # the same counters in Prometheus text form
            port = int(srv.http_url.rsplit(":", 1)[1])
            out = _raw_http(port, b"GET /metrics HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n", expect_close=True)
            head, _, text = out.partition(b"\r\n\r\n")
            assert b"200 OK" in head and b"text/plain" in head.lower()
            metrics = {}
            for line in text.decode().splitlines():
                if line and not line.startswith("#"):
                    key, value = line.rsplit(" ", 1)
                    metrics[key] = int(value)
            assert metrics['nv_inference_request_success{model="simple",version="1"}'] >= 20
            assert metrics['nv_inference_count{model="simple",version="1"}'] == stats["inference_count"]
            native = metrics["tca_native_handled_total"] + metrics["tca_native_declined_total"]
            assert native > 0 if impl == "native" else native == 0  # only the native front offers it requests
            c.unload_model("simple")
            assert not c.is_model_ready("simple")


def _raw_http(port, payload, expect_close=False, timeout=5.0):
    """Sends raw bytes, returns everything received until the peer is idle / closes."""
    import socket

    s = socket.create_connection(("127.0.0.1", port), timeout=timeout)
    try:
        s.sendall(payload)
        s.settimeout(0.5 if not expect_close else timeout)
        data = b""
        while True:
            try:
                chunk = s.recv(65536)
            except socket.timeout:
                break
            if not chunk:
                break
            data += chunk
        return data
    finally:
        s.close()


@pytest.mark.parametrize("impl", ["native", "asyncio"])
def test_http_front_protocol_edges(impl):
    """Wire-level behaviour a well-behaved HTTP/1.1 server owes its clients."""
    import json

    from triton_client_amd.server import Server

    body = json.dumps({"inputs": [
        {"name": "INPUT0", "datatype": "INT32", "shape": [1, 16], "data": list(range(16))},
        {"name": "INPUT1", "datatype": "INT32", "shape": [1, 16], "data": [1] * 16}]}).encode()
    with Server(models=["simple"], enable_grpc=False, http_impl=impl) as srv:
        port = int(srv.http_url.rsplit(":", 1)[1])
        post = b"POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nContent-Length: %d\r\n\r\n" % len(body) + body
        # two requests back to back on one connection: two answers, in order
        out = _raw_http(port, post + b"GET /v2/health/live HTTP/1.1\r\nHost: x\r\n\r\n")
        assert out.count(b"HTTP/1.1 200 OK") == 2 and b'"OUTPUT0"' in out
        # Expect: 100-continue with the body held back
        import socket

        s = socket.create_connection(("127.0.0.1", port), timeout=5)
        s.sendall(b"POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nExpect: 100-continue\r\nContent-Length: %d\r\n\r\n"
                  % len(body))
        assert s.recv(1024).startswith(b"HTTP/1.1 100 Continue")
        s.sendall(body)
        assert b"200 OK" in s.recv(65536)
        s.close()
        # Connection: close is honoured
        out = _raw_http(port, b"GET /v2/health/ready HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n",
                        expect_close=True)
        assert b"200 OK" in out and b"connection: close" in out.lower()
        # unknown endpoint and malformed JSON are 4xx with a JSON error, the connection survives
        out = _raw_http(port,
                        b"GET /v2/nothing HTTP/1.1\r\nHost: x\r\n\r\nGET /v2/health/live HTTP/1.1\r\nHost: x\r\n\r\n")
        assert b"404" in out.split(b"\r\n", 1)[0] and out.count(b"HTTP/1.1 200 OK") == 1
        bad = b"{not json"
        out = _raw_http(port, b"POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nContent-Length: %d\r\n\r\n" % len(bad) + bad)
        assert b"400" in out.split(b"\r\n", 1)[0] and b'"error"' in out
        if impl == "native":
            # chunked request body (the asyncio front does not take those)
            half = len(body) // 2
            chunked = (b"POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nTransfer-Encoding: chunked\r\n\r\n" +
                       b"%x\r\n" % half + body[:half] + b"\r\n" + b"%x\r\n" % (len(body) - half) + body[half:] +
                       b"\r\n0\r\n\r\n")
            out = _raw_http(port, chunked)
            assert b"200 OK" in out and b'"OUTPUT1"' in out
            # garbage instead of a request line: 400 and the connection is closed
            out = _raw_http(port, b"NONSENSE\r\n\r\n", expect_close=True)
            assert out.startswith(b"HTTP/1.1 400")
            assert srv.http.stats()["requests"] >= 6


def test_malloc_arena_cap_is_applied_once_and_can_be_turned_off():
    """The server caps glibc's malloc arenas (message bodies of megabytes are allocated on reactor threads and
    freed on completer threads); ``TCA_MALLOC_ARENAS=0`` leaves the allocator alone."""
    import os
    import subprocess
    import sys

    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import ctypes\n"
        "calls = []\n"
        "import triton_client_amd.server as srv\n"
        "real = ctypes.CDLL\n"
        "class Fake:\n"
        "    def mallopt(self, what, value):\n"
        "        calls.append((what, value)); return 1\n"
        "ctypes.CDLL = lambda *a, **k: Fake()\n"
        "srv._limit_malloc_arenas(); srv._limit_malloc_arenas()\n"
        "print('CALLS', calls)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                        env={k: v for k, v in os.environ.items() if k not in ("TCA_MALLOC_ARENAS", "MALLOC_ARENA_MAX")})
    assert "CALLS [(-8, 8)]" in on.stdout, on.stdout + on.stderr
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, TCA_MALLOC_ARENAS="0"))
    assert "CALLS []" in off.stdout, off.stdout + off.stderr

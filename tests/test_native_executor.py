"""Native execution path: ordinary requests for models with a C++ kernel are answered by the
serving fronts without the interpreter; everything else takes the general path.  Both must be
indistinguishable to a client, so every case runs against a server with the path on and one
with it off and the answers are compared."""
import numpy as np
import pytest

import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient
import tritonclient.utils.shared_memory as shm
from tritonclient.utils import InferenceServerException


@pytest.fixture(scope="module")
def servers():
    from triton_client_amd.server import Server

    models = ["simple", "simple_int8", "simple_string", "custom_identity_int32", "repeat_int32", "simple_sequence",
              "simple_dyna_sequence"]
    on = Server(models=models, native_exec=True, model_control="explicit").start()
    off = Server(models=models, native_exec=False, model_control="explicit").start()
    for srv in (on, off):
        with httpclient.InferenceServerClient(srv.http_url) as c:
            for m in models:
                c.load_model(m)
    if on._native is None:
        on.stop()
        off.stop()
        pytest.skip("native extension not built")
    yield on, off
    on.stop()
    off.stop()


def _handled(srv):
    return srv._native.counters()["handled"]


def _inputs(mod, a, b, dtype="INT32", **kw):
    return [mod.InferInput("INPUT0", list(a.shape), dtype).set_data_from_numpy(a, **kw),
            mod.InferInput("INPUT1", list(b.shape), dtype).set_data_from_numpy(b, **kw)]


@pytest.mark.parametrize("batch", [1, 3, 8])
def test_grpc_wire_requests_run_natively_and_match(servers, batch):
    on, off = servers
    rng = np.random.default_rng(batch)
    a = rng.integers(-2**31, 2**31 - 1, size=(batch, 16), dtype=np.int64).astype(np.int32)  # wrap-around included
    b = rng.integers(-2**31, 2**31 - 1, size=(batch, 16), dtype=np.int64).astype(np.int32)
    before = _handled(on)
    answers = []
    for srv in (on, off):
        with grpcclient.InferenceServerClient(srv.grpc_url) as c:
            r = c.infer("simple", _inputs(grpcclient, a, b), request_id="abc")
            only1 = c.infer("simple", _inputs(grpcclient, a, b), outputs=[grpcclient.InferRequestedOutput("OUTPUT1")])
            answers.append((r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1"), r.get_response().id,
                            r.get_response().model_version, [o.name for o in only1.get_response().outputs],
                            only1.as_numpy("OUTPUT1")))
    assert _handled(on) == before + 2
    for x, y in zip(*answers):
        if isinstance(x, np.ndarray):
            np.testing.assert_array_equal(x, y)
        else:
            assert x == y
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(answers[0][0], a + b)
        np.testing.assert_array_equal(answers[0][1], a - b)


def test_int8_and_identity_kernels(servers):
    on, off = servers
    a = np.arange(-8, 8, dtype=np.int8).reshape(1, 16) * 15
    b = np.full((1, 16), 100, dtype=np.int8)
    before = _handled(on)
    outs = []
    for srv in (on, off):
        with grpcclient.InferenceServerClient(srv.grpc_url) as c:
            r = c.infer("simple_int8", _inputs(grpcclient, a, b, "INT8"))
            outs.append((r.as_numpy("OUTPUT0"), r.as_numpy("OUTPUT1")))
            x = np.arange(32, dtype=np.int32).reshape(1, 32)
            inp = grpcclient.InferInput("INPUT0", [1, 32], "INT32").set_data_from_numpy(x)
            np.testing.assert_array_equal(c.infer("custom_identity_int32", [inp]).as_numpy("OUTPUT0"), x)
            # a request parameter (here: the delay knob) sends the same model through Python
            slow = c.infer("custom_identity_int32", [inp], parameters={"execute_delay_ms": 1})
            np.testing.assert_array_equal(slow.as_numpy("OUTPUT0"), x)
    assert _handled(on) == before + 2  # both have kernels: add_sub on INT8, identity with a free dimension
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("binary_in,binary_out", [(True, True), (True, False), (False, True)])
def test_http_requests_match(servers, binary_in, binary_out):
    on, off = servers
    a = np.arange(32, dtype=np.int32).reshape(2, 16)
    b = np.full((2, 16), 7, dtype=np.int32)
    before = _handled(on)
    docs = []
    for srv in (on, off):
        with httpclient.InferenceServerClient(srv.http_url) as c:
            outputs = [httpclient.InferRequestedOutput("OUTPUT0", binary_data=binary_out),
                       httpclient.InferRequestedOutput("OUTPUT1", binary_data=binary_out)]
            r = c.infer("simple", _inputs(httpclient, a, b, binary_data=binary_in), outputs=outputs, request_id="r1")
            np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
            np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
            docs.append(r.get_response())
    assert docs[0] == docs[1]
    # only fully binary exchanges have a native form; JSON tensors go through Python
    assert _handled(on) == before + (1 if binary_in and binary_out else 0)


@pytest.mark.parametrize("protocol", ["http", "grpc"])
def test_system_shared_memory_requests_match(servers, protocol):
    on, off = servers
    mod = httpclient if protocol == "http" else grpcclient
    a = np.arange(16, dtype=np.int32).reshape(1, 16)
    b = np.full((1, 16), 5, dtype=np.int32)
    for tag, srv in (("on", on), ("off", off)):
        key_in, key_out = f"/ne_in_{protocol}_{tag}", f"/ne_out_{protocol}_{tag}"
        h_in = shm.create_shared_memory_region("ne_in", key_in, 128)
        h_out = shm.create_shared_memory_region("ne_out", key_out, 128)
        before = _handled(on)
        try:
            shm.set_shared_memory_region(h_in, [a, b])
            with mod.InferenceServerClient(srv.http_url if protocol == "http" else srv.grpc_url) as c:
                c.register_system_shared_memory("ne_in", key_in, 128)
                c.register_system_shared_memory("ne_out", key_out, 128)
                i0 = mod.InferInput("INPUT0", [1, 16], "INT32").set_shared_memory("ne_in", 64)
                i1 = mod.InferInput("INPUT1", [1, 16], "INT32").set_shared_memory("ne_in", 64, offset=64)
                o0 = mod.InferRequestedOutput("OUTPUT0")
                o0.set_shared_memory("ne_out", 64)
                o1 = mod.InferRequestedOutput("OUTPUT1")
                o1.set_shared_memory("ne_out", 64, offset=64)
                r = c.infer("simple", [i0, i1], outputs=[o0, o1])
                out0 = r.get_output("OUTPUT0")
                params = out0["parameters"] if isinstance(out0, dict) else {k: (v.string_param or v.int64_param)
                                                                              for k, v in out0.parameters.items()}
                assert params["shared_memory_region"] == "ne_out" and int(params["shared_memory_byte_size"]) == 64
                np.testing.assert_array_equal(shm.get_contents_as_numpy(h_out, np.int32, [1, 16]), a + b)
                np.testing.assert_array_equal(shm.get_contents_as_numpy(h_out, np.int32, [1, 16], offset=64), a - b)
                # an output window that is too small is an error from the general path, natively declined
                o_small = mod.InferRequestedOutput("OUTPUT0")
                o_small.set_shared_memory("ne_out", 32)
                with pytest.raises(InferenceServerException, match="should be at least 64 bytes"):
                    c.infer("simple", [i0, i1], outputs=[o_small])
                c.unregister_system_shared_memory("ne_in")
                with pytest.raises(InferenceServerException, match="Unable to find"):
                    c.infer("simple", [i0, i1], outputs=[o0, o1])
                c.unregister_system_shared_memory("ne_out")
            if tag == "on":
                assert _handled(on) == before + 1
        finally:
            shm.destroy_shared_memory_region(h_in)
            shm.destroy_shared_memory_region(h_out)


def test_errors_and_special_requests_take_the_general_path(servers):
    on, off = servers
    a = np.arange(16, dtype=np.int32).reshape(1, 16)
    b = np.ones((1, 16), dtype=np.int32)
    before = _handled(on)
    for srv in (on, off):
        with grpcclient.InferenceServerClient(srv.grpc_url) as c:
            with pytest.raises(InferenceServerException, match="unexpected shape"):
                bad = np.zeros((1, 15), dtype=np.int32)
                c.infer("simple", _inputs(grpcclient, bad, bad))
            with pytest.raises(InferenceServerException, match="batch-size must be <= 8"):
                big = np.zeros((9, 16), dtype=np.int32)
                c.infer("simple", _inputs(grpcclient, big, big))
            with pytest.raises(InferenceServerException, match="data-type"):
                f = np.zeros((1, 16), dtype=np.float32)
                c.infer("simple", _inputs(grpcclient, f, f, "FP32"))
            with pytest.raises(InferenceServerException, match="unexpected inference output"):
                c.infer("simple", _inputs(grpcclient, a, b), outputs=[grpcclient.InferRequestedOutput("NOPE")])
            with pytest.raises(InferenceServerException, match="version 7 is not found"):
                c.infer("simple", _inputs(grpcclient, a, b), model_version="7")
            # request parameters, classification and a compressed request: answered, by Python
            r = c.infer("simple", _inputs(grpcclient, a, b), parameters={"note": "x"})
            np.testing.assert_array_equal(r.as_numpy("OUTPUT0"), a + b)
            r = c.infer("simple", _inputs(grpcclient, a, b),
                        outputs=[grpcclient.InferRequestedOutput("OUTPUT0", class_count=2)])
            assert r.as_numpy("OUTPUT0").shape == (1, 2)
            r = c.infer("simple", _inputs(grpcclient, a, b), model_version="1")  # the current version: native again
            np.testing.assert_array_equal(r.as_numpy("OUTPUT1"), a - b)
    assert _handled(on) == before + 1


def test_statistics_count_native_requests_and_unload_stops_the_path(servers):
    on, off = servers
    a = np.arange(32, dtype=np.int32).reshape(2, 16)
    b = np.ones((2, 16), dtype=np.int32)
    with grpcclient.InferenceServerClient(on.grpc_url) as c:
        s0 = c.get_inference_statistics("simple").model_stats[0]
        for _ in range(5):
            c.infer("simple", _inputs(grpcclient, a, b))
        s1 = c.get_inference_statistics("simple").model_stats[0]
        assert s1.inference_count - s0.inference_count == 10 and s1.execution_count - s0.execution_count == 5
        assert s1.inference_stats.success.count - s0.inference_stats.success.count == 5
        assert s1.inference_stats.success.ns > s0.inference_stats.success.ns
        assert any(bs.batch_size == 2 and bs.compute_infer.count >= 5 for bs in s1.batch_stats)
        before = _handled(on)
        c.unload_model("simple")
        with pytest.raises(InferenceServerException, match="unknown model"):
            c.infer("simple", _inputs(grpcclient, a, b))
        c.load_model("simple")
        np.testing.assert_array_equal(c.infer("simple", _inputs(grpcclient, a, b)).as_numpy("OUTPUT0"), a + b)
        assert _handled(on) == before + 1
        # a load with fault-injection parameters must leave the native path
        c.load_model("simple", config='{"parameters": {"execute_delay_ms": {"string_value": "1"}}}')
        before = _handled(on)
        np.testing.assert_array_equal(c.infer("simple", _inputs(grpcclient, a, b)).as_numpy("OUTPUT0"), a + b)
        assert _handled(on) == before
        c.load_model("simple")


def _stream_collect(url, send, expected):
    """Opens a stream, runs ``send(client)``, returns the first ``expected`` (result, error) pairs."""
    import queue

    done = queue.Queue()
    got = []
    with grpcclient.InferenceServerClient(url) as c:
        c.start_stream(lambda result, error: done.put((result, error)))
        send(c)
        for _ in range(expected):
            got.append(done.get(timeout=20))
        c.stop_stream()
    return got


@pytest.mark.parametrize("final", [False, True])
def test_decoupled_stream_runs_natively_and_matches(servers, final):
    on, off = servers
    for srv in (on, off):
        with httpclient.InferenceServerClient(srv.http_url) as c:
            c.load_model("repeat_int32")
    values = np.array([7, -3, 2**31 - 1, 0, 42], dtype=np.int32)

    def send(c, delays=None, outputs=None):
        n = len(values)
        c.async_stream_infer("repeat_int32", [
            grpcclient.InferInput("IN", [n], "INT32").set_data_from_numpy(values),
            grpcclient.InferInput("DELAY", [n], "UINT32").set_data_from_numpy(
                np.zeros(n, dtype=np.uint32) if delays is None else delays),
            grpcclient.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(np.zeros(1, dtype=np.uint32)),
        ], request_id="rep", enable_empty_final_response=final, outputs=outputs)

    def summarise(pairs):
        rows = []
        for result, error in pairs:
            assert error is None, error
            resp = result.get_response()
            flag = resp.parameters["triton_final_response"].bool_param if "triton_final_response" in resp.parameters \
                else None
            rows.append((resp.id, resp.model_name, resp.model_version, [o.name for o in resp.outputs], flag,
                         None if not resp.outputs else int(result.as_numpy(resp.outputs[0].name)[0]),
                         None if len(resp.outputs) < 2 else int(result.as_numpy("IDX")[0])))
        return rows

    expected = len(values) + (1 if final else 0)
    before = _handled(on)
    a = summarise(_stream_collect(on.grpc_url, send, expected))
    b = summarise(_stream_collect(off.grpc_url, send, expected))
    assert a == b and _handled(on) == before + 1
    assert [r[5] for r in a[:len(values)]] == [int(v) for v in values]
    assert [r[6] for r in a[:len(values)]] == list(range(len(values)))
    if final:
        assert a[-1][3] == [] and a[-1][4] is True and all(r[4] is False for r in a[:-1])
    # a subset of the outputs
    only_idx = [grpcclient.InferRequestedOutput("IDX")]
    a = summarise(_stream_collect(on.grpc_url, lambda c: send(c, outputs=only_idx), expected))
    b = summarise(_stream_collect(off.grpc_url, lambda c: send(c, outputs=only_idx), expected))
    assert a == b and a[0][3] == ["IDX"]
    # responses that have to wait are produced by the general path
    before = _handled(on)
    delays = np.array([0, 2, 0, 1, 0], dtype=np.uint32)
    a = summarise(_stream_collect(on.grpc_url, lambda c: send(c, delays=delays), expected))
    assert _handled(on) == before and [r[5] for r in a[:len(values)]] == [int(v) for v in values]
    # the unary call keeps refusing a decoupled model
    with grpcclient.InferenceServerClient(on.grpc_url) as c:
        with pytest.raises(InferenceServerException, match="decoupled"):
            c.infer("repeat_int32", [
                grpcclient.InferInput("IN", [5], "INT32").set_data_from_numpy(values),
                grpcclient.InferInput("DELAY", [5], "UINT32").set_data_from_numpy(np.zeros(5, dtype=np.uint32)),
                grpcclient.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(np.zeros(1, dtype=np.uint32))])


def test_ordinary_models_over_streams(servers):
    """`simple` through async_stream_infer: native answers keep the order of the requests; the statistics count them."""
    on, off = servers
    a = np.arange(16, dtype=np.int32).reshape(1, 16)

    def send(c):
        for k in range(12):
            c.async_stream_infer("simple", _inputs(grpcclient, a + k, a), request_id=str(k),
                                 enable_empty_final_response=(k % 2 == 0))

    before = _handled(on)
    with grpcclient.InferenceServerClient(on.grpc_url) as c:
        s0 = c.get_inference_statistics("simple").model_stats[0].inference_count
    rows = []
    for srv in (on, off):
        pairs = _stream_collect(srv.grpc_url, send, 12)
        assert all(e is None for _, e in pairs)
        rows.append(sorted((int(r.get_response().id), r.as_numpy("OUTPUT0").tolist(),
                            r.get_response().parameters["triton_final_response"].bool_param
                            if "triton_final_response" in r.get_response().parameters else None) for r, _ in pairs))
        if srv is on:
            assert [int(r.get_response().id) for r, _ in pairs] == list(range(12))
    assert rows[0] == rows[1] and _handled(on) == before + 12
    assert rows[0][0][2] is True and rows[0][1][2] is None
    with grpcclient.InferenceServerClient(on.grpc_url) as c:
        assert c.get_inference_statistics("simple").model_stats[0].inference_count - s0 == 12


def test_a_sequence_stream_with_python_messages_keeps_its_order(servers):
    """One stream carrying a sequence whose messages alternate between ones the executor answers and ones only
    Python answers (a request parameter): a message answered natively is on the wire at once, one handed to
    Python runs later, so once a stream's message went to Python its later messages follow it there - the
    accumulator sees the values in request order and the responses come back in that order."""
    on, off = servers

    def send(c):
        for k in range(10):
            extra = {"parameters": {"note": "x"}} if k in (1, 4, 5) else {}
            c.async_stream_infer("simple_sequence", _sequence_request(grpcclient, 1 + k), sequence_id=4242,
                                 sequence_start=(k == 0), sequence_end=(k == 9), request_id=str(k), **extra)

    rows = []
    before = _handled(on)
    for srv in (on, off):
        pairs = _stream_collect(srv.grpc_url, send, 10)
        assert all(e is None for _, e in pairs)
        rows.append([(int(r.get_response().id), int(r.as_numpy("OUTPUT")[0, 0])) for r, _ in pairs])
    assert [i for i, _ in rows[0]] == list(range(10)) and rows[0] == rows[1]
    assert _handled(on) == before + 1  # the message ahead of the first Python one
    # a new stream starts on the native path again
    pairs = _stream_collect(on.grpc_url, lambda c: c.async_stream_infer(
        "simple_sequence", _sequence_request(grpcclient, 3), sequence_id=4243, sequence_start=True, sequence_end=True), 1)
    assert pairs[0][1] is None and _handled(on) == before + 2
    assert on._native.open_sequences("simple_sequence") == 0


def _sequence_request(mod, value):
    return [mod.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(np.array([[value]], dtype=np.int32))]


@pytest.mark.parametrize("protocol", ["grpc", "http"])
def test_stateful_models_run_natively_and_match(servers, protocol):
    """Sequences (numeric correlation ids) on `simple_sequence` / `simple_dyna_sequence`: start adds one,
    the dyna model adds the id on the last request, and both servers answer the same - errors included."""
    on, off = servers
    mod = grpcclient if protocol == "grpc" else httpclient
    before = _handled(on)
    transcripts = []
    for srv in (on, off):
        rows = []
        with mod.InferenceServerClient(srv.grpc_url if protocol == "grpc" else srv.http_url) as c:
            for model, seq in (("simple_sequence", 1001), ("simple_dyna_sequence", 1002)):
                for k, value in enumerate((10, 20, 2**31 - 1, -5)):
                    r = c.infer(model, _sequence_request(mod, value), sequence_id=seq, sequence_start=(k == 0),
                                sequence_end=(k == 3), request_id=f"{model}-{k}")
                    resp = r.get_response()
                    rows.append((resp.id if protocol == "grpc" else resp["id"], int(r.as_numpy("OUTPUT")[0, 0])))
                # a one-request sequence, and a sequence that is opened twice
                r = c.infer(model, _sequence_request(mod, 7), sequence_id=seq + 10, sequence_start=True, sequence_end=True)
                rows.append(int(r.as_numpy("OUTPUT")[0, 0]))
                for _ in range(2):
                    r = c.infer(model, _sequence_request(mod, 1), sequence_id=seq + 20, sequence_start=True)
                    rows.append(int(r.as_numpy("OUTPUT")[0, 0]))
                c.infer(model, _sequence_request(mod, 1), sequence_id=seq + 20, sequence_end=True)
                # no start flag, no id, an ended sequence: refused in the same words
                for kwargs in ({"sequence_id": seq + 30}, {}, {"sequence_id": seq, "sequence_end": True}):
                    with pytest.raises(InferenceServerException) as err:
                        c.infer(model, _sequence_request(mod, 1), **kwargs)
                    rows.append(str(err.value))
                # string correlation ids stay with the general path and have a namespace of their own
                r = c.infer(model, _sequence_request(mod, 3), sequence_id="77" if model.endswith("dyna_sequence") else "s",
                            sequence_start=True, sequence_end=True)
                rows.append(int(r.as_numpy("OUTPUT")[0, 0]))
        transcripts.append(rows)
    assert transcripts[0] == transcripts[1]
    rows = transcripts[0]
    assert [v for _, v in rows[:4]] == [11, 20, 2**31 - 1, -5]
    assert "START" in rows[7]
    assert _handled(on) - before == 2 * (4 + 1 + 3)  # everything but the refusals and the string ids
    assert on._native.open_sequences("simple_sequence") == 0 and on._native.open_sequences("simple_dyna_sequence") == 0
    dyna = transcripts[0][len(rows) // 2:]
    assert [v for _, v in dyna[:4]] == [11, 20, 2**31 - 1, np.int32(-5 + 1002)]


def test_sequences_over_streams_and_mixed_paths(servers):
    """A sequence whose requests alternate between the native path and the general one (here: a request
    parameter the native path does not know) stays one sequence; streams carry sequences too."""
    on, off = servers
    outs = []
    for srv in (on, off):
        with grpcclient.InferenceServerClient(srv.grpc_url) as c:
            got = []
            for k in range(4):
                extra = {"parameters": {"note": "x"}} if k % 2 else {}
                r = c.infer("simple_sequence", _sequence_request(grpcclient, k), sequence_id=555, sequence_start=(k == 0),
                            sequence_end=(k == 3), **extra)
                got.append(int(r.as_numpy("OUTPUT")[0, 0]))
            outs.append(got)

        def send(c):
            for k in range(6):
                c.async_stream_infer("simple_dyna_sequence", _sequence_request(grpcclient, 100 + k), sequence_id=9,
                                     sequence_start=(k == 0), sequence_end=(k == 5), request_id=str(k))

        pairs = _stream_collect(srv.grpc_url, send, 6)
        assert all(e is None for _, e in pairs)
        outs.append(sorted((int(r.get_response().id), int(r.as_numpy("OUTPUT")[0, 0])) for r, _ in pairs))
    assert outs[0] == outs[2] == [1, 1, 2, 3]
    assert outs[1] == outs[3] == [(0, 101), (1, 101), (2, 102), (3, 103), (4, 104), (5, 114)]
    assert on._native.open_sequences("simple_sequence") == 0

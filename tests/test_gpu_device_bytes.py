"""BYTES tensors in device shared memory served by the GPU kernels (index / decimal add-sub /
pack): the output regions hold exactly the bytes ``serialize_byte_tensor`` produces for the
expected strings, the client reads them back without copying the region's tail, and requests
that do not qualify (wire outputs, malformed input) still behave as on the host path."""

import threading

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module", params=["native", "interpreter"])
def served(request):
    """The same server twice: requests queued and answered by the C++ executor's batch thread,
    and (``TCA_NATIVE_DEVICE_BYTES=0``) by the Python batch thread.  Same kernels, same answers."""
    from triton_client_amd.server import spawn_server

    env = {"TCA_NATIVE_DEVICE_BYTES": "1" if request.param == "native" else "0"}
    srv = spawn_server(models=["simple_string", "simple_identity"], gpu=0, env=env)
    yield srv
    srv.stop()


def _regions(hipshm, client, names, nbytes):
    handles = {}
    for n in names:
        h = hipshm.create_shared_memory_region(n, nbytes, 0)
        client.register_cuda_shared_memory(n, hipshm.get_raw_handle(h), 0, nbytes)
        handles[n] = h
    return handles


def _cleanup(hipshm, client, handles):
    for n, h in handles.items():
        client.unregister_cuda_shared_memory(n)
        hipshm.destroy_shared_memory_region(h)


def _request(g, batch, tag, in_sizes, out_bytes):
    inputs = []
    for i, n in enumerate(in_sizes):
        t = g.InferInput(f"INPUT{i}", [batch, 16], "BYTES")
        t.set_shared_memory(f"{tag}_in{i}", n)
        inputs.append(t)
    outputs = []
    for i in range(2):
        o = g.InferRequestedOutput(f"OUTPUT{i}")
        o.set_shared_memory(f"{tag}_out{i}", out_bytes)
        outputs.append(o)
    return inputs, outputs


@pytest.mark.parametrize("batch", [1, 4])
def test_string_add_sub_golden_bytes(served, batch):
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import serialize_byte_tensor

    rng = np.random.default_rng(batch)
    a = rng.integers(-10 ** 9, 10 ** 9, size=(batch, 16))
    b = rng.integers(-10 ** 6, 10 ** 6, size=(batch, 16))
    a[0, 0], b[0, 0] = 0, 0
    sa = np.array([str(v).encode() for v in a.reshape(-1)], dtype=object).reshape(batch, 16)
    sb = np.array([(" +%d " % v if v >= 0 else str(v)).encode() for v in b.reshape(-1)],
                  dtype=object).reshape(batch, 16)
    blob_a, blob_b = serialize_byte_tensor(sa).item(), serialize_byte_tensor(sb).item()
    cap = 4096
    with g.InferenceServerClient(served.grpc_url) as c:
        tag = f"gold{batch}"
        handles = _regions(hipshm, c, [f"{tag}_in0", f"{tag}_in1", f"{tag}_out0", f"{tag}_out1"], cap)
        try:
            hipshm.set_shared_memory_region(handles[f"{tag}_in0"], [serialize_byte_tensor(sa)])
            hipshm.set_shared_memory_region(handles[f"{tag}_in1"], [serialize_byte_tensor(sb)])
            ins, outs = _request(g, batch, tag, [len(blob_a), len(blob_b)], cap)
            res = c.infer("simple_string", ins, outputs=outs)
            assert res.get_output("OUTPUT0").shape == [batch, 16]
            for name, expect in (("out0", a + b), ("out1", a - b)):
                want = serialize_byte_tensor(np.array([str(v).encode() for v in expect.reshape(-1)],
                                                      dtype=object)).item()
                raw = hipshm.get_contents_as_numpy(handles[f"{tag}_{name}"], np.uint8, [len(want)])
                assert raw.tobytes() == want  # byte for byte the reference wire form
                got = hipshm.get_contents_as_numpy(handles[f"{tag}_{name}"], np.object_, [batch, 16])
                assert [int(x) for x in got.reshape(-1)] == [int(v) for v in expect.reshape(-1)]
            # statistics count the request as a success
            stats = c.get_inference_statistics("simple_string", as_json=True)["model_stats"][0]
            assert int(stats["inference_stats"]["success"]["count"]) >= 1
        finally:
            _cleanup(hipshm, c, handles)


def test_http_requests_take_the_same_path(served):
    """The REST front: tensors in device regions, the same kernels, the same bytes; the response names the
    output regions.  A result that does not fit is refused in the general path's words."""
    import tritonclient.http as h
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import InferenceServerException, serialize_byte_tensor

    a = np.arange(-8, 8).reshape(1, 16) * 1234567
    b = np.arange(16).reshape(1, 16) - 5
    sa = np.array([str(v).encode() for v in a.reshape(-1)], dtype=object).reshape(1, 16)
    sb = np.array([str(v).encode() for v in b.reshape(-1)], dtype=object).reshape(1, 16)
    blob_a, blob_b = serialize_byte_tensor(sa).item(), serialize_byte_tensor(sb).item()
    with h.InferenceServerClient(served.http_url) as c:
        handles = _regions(hipshm, c, ["ht_in0", "ht_in1", "ht_out0", "ht_out1"], 2048)
        try:
            hipshm.set_shared_memory_region(handles["ht_in0"], [serialize_byte_tensor(sa)])
            hipshm.set_shared_memory_region(handles["ht_in1"], [serialize_byte_tensor(sb)])
            ins, outs = _request(h, 1, "ht", [len(blob_a), len(blob_b)], 2048)
            res = c.infer("simple_string", ins, outputs=outs, request_id="rest-1")
            body = res.get_response()
            assert body["id"] == "rest-1" and body["model_name"] == "simple_string"
            out0 = res.get_output("OUTPUT0")
            assert out0["shape"] == [1, 16] and out0["parameters"]["shared_memory_region"] == "ht_out0"
            for name, expect in (("out0", a + b), ("out1", a - b)):
                want = serialize_byte_tensor(np.array([str(v).encode() for v in expect.reshape(-1)],
                                                      dtype=object)).item()
                raw = hipshm.get_contents_as_numpy(handles[f"ht_{name}"], np.uint8, [len(want)])
                assert raw.tobytes() == want
            ins, outs = _request(h, 1, "ht", [len(blob_a), len(blob_b)], 16)
            with pytest.raises(InferenceServerException, match="too small|should be at least"):
                c.infer("simple_string", ins, outputs=outs)
        finally:
            _cleanup(hipshm, c, handles)


def test_client_reads_only_the_stream(served):
    """The device path leaves an index behind the result; the client fetches the index and then
    exactly the stream - not the rest of a large region."""
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    import triton_client_amd.utils.hip_shared_memory as impl
    from tritonclient.utils import serialize_byte_tensor

    vals = np.array([str(i).encode() for i in range(16)], dtype=object).reshape(1, 16)
    blob = serialize_byte_tensor(vals).item()
    cap = 1 << 20
    with g.InferenceServerClient(served.grpc_url) as c:
        handles = _regions(hipshm, c, ["big_in0", "big_in1", "big_out0", "big_out1"], cap)
        try:
            for n in ("big_in0", "big_in1"):
                hipshm.set_shared_memory_region(handles[n], [serialize_byte_tensor(vals)])
            ins, outs = _request(g, 1, "big", [len(blob), len(blob)], cap)
            c.infer("simple_string", ins, outputs=outs)
            reads = []
            orig = impl._read
            impl._read = lambda h, off, n: (reads.append(n), orig(h, off, n))[1]
            try:
                got = hipshm.get_contents_as_numpy(handles["big_out0"], np.object_, [1, 16])
            finally:
                impl._read = orig
            assert [int(x) for x in got.reshape(-1)] == [2 * i for i in range(16)]
            assert sum(reads) < 1024, reads
        finally:
            _cleanup(hipshm, c, handles)


def test_identity_keeps_nuls_and_empty_elements(served):
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import serialize_byte_tensor

    vals = np.array([b"", b"a\x00b", b"\x00" * 70, b"xyz" * 100] + [bytes([i]) * i for i in range(1, 13)], dtype=object)
    vals = vals.reshape(1, 16)
    blob = serialize_byte_tensor(vals).item()
    with g.InferenceServerClient(served.grpc_url) as c:
        handles = _regions(hipshm, c, ["id_in", "id_out"], 8192)
        try:
            hipshm.set_shared_memory_region(handles["id_in"], [serialize_byte_tensor(vals)])
            t = g.InferInput("INPUT0", [1, 16], "BYTES")
            t.set_shared_memory("id_in", len(blob))
            o = g.InferRequestedOutput("OUTPUT0")
            o.set_shared_memory("id_out", 8192)
            c.infer("simple_identity", [t], outputs=[o])
            raw = hipshm.get_contents_as_numpy(handles["id_out"], np.uint8, [len(blob)])
            assert raw.tobytes() == blob
            # wire output with a device input: the host path answers
            res = c.infer("simple_identity", [t])
            assert list(res.as_numpy("OUTPUT0").reshape(-1)) == list(vals.reshape(-1))
        finally:
            _cleanup(hipshm, c, handles)


def test_errors_match_the_host_path(served):
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import InferenceServerException, serialize_byte_tensor

    good = np.array([b"1"] * 16, dtype=object).reshape(1, 16)
    bad = good.copy()
    bad[0, 5] = b"12x"
    with g.InferenceServerClient(served.grpc_url) as c:
        handles = _regions(hipshm, c, ["e_in0", "e_in1", "e_out0", "e_out1"], 4096)
        try:
            hipshm.set_shared_memory_region(handles["e_in0"], [serialize_byte_tensor(good)])
            hipshm.set_shared_memory_region(handles["e_in1"], [serialize_byte_tensor(bad)])
            n_good, n_bad = len(serialize_byte_tensor(good).item()), len(serialize_byte_tensor(bad).item())
            ins, outs = _request(g, 1, "e", [n_good, n_bad], 4096)
            with pytest.raises(InferenceServerException, match="decimal integer strings"):
                c.infer("simple_string", ins, outputs=outs)
            # a stream that ends early
            ins, outs = _request(g, 1, "e", [n_good, n_bad - 3], 4096)
            with pytest.raises(InferenceServerException):
                c.infer("simple_string", ins, outputs=outs)
            # output range too small for the result
            hipshm.set_shared_memory_region(handles["e_in1"], [serialize_byte_tensor(good)])
            ins, outs = _request(g, 1, "e", [n_good, n_good], 16)
            with pytest.raises(InferenceServerException, match="too small|should be at least"):
                c.infer("simple_string", ins, outputs=outs)
            # and the model still serves
            ins, outs = _request(g, 1, "e", [n_good, n_good], 4096)
            c.infer("simple_string", ins, outputs=outs)
            got = hipshm.get_contents_as_numpy(handles["e_out0"], np.object_, [1, 16])
            assert [int(x) for x in got.reshape(-1)] == [2] * 16
        finally:
            _cleanup(hipshm, c, handles)


def test_concurrent_requests_share_launches(served):
    """32 clients at once: requests are answered correctly whatever batches they end up in."""
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import serialize_byte_tensor

    errors = []

    def worker(k):
        try:
            vals = np.array([str(k * 100 + i).encode() for i in range(16)], dtype=object).reshape(1, 16)
            blob = serialize_byte_tensor(vals).item()
            with g.InferenceServerClient(served.grpc_url) as c:
                tag = f"cc{k}"
                handles = _regions(hipshm, c, [f"{tag}_in0", f"{tag}_in1", f"{tag}_out0", f"{tag}_out1"], 1024)
                try:
                    hipshm.set_shared_memory_region(handles[f"{tag}_in0"], [serialize_byte_tensor(vals)])
                    hipshm.set_shared_memory_region(handles[f"{tag}_in1"], [serialize_byte_tensor(vals)])
                    ins, outs = _request(g, 1, tag, [len(blob), len(blob)], 1024)
                    for _ in range(50):
                        c.infer("simple_string", ins, outputs=outs)
                        got = hipshm.get_contents_as_numpy(handles[f"{tag}_out0"], np.object_, [1, 16])
                        assert [int(x) for x in got.reshape(-1)] == [2 * (k * 100 + i) for i in range(16)]
                        got = hipshm.get_contents_as_numpy(handles[f"{tag}_out1"], np.object_, [1, 16])
                        assert [int(x) for x in got.reshape(-1)] == [0] * 16
                finally:
                    _cleanup(hipshm, c, handles)
        except Exception as e:  # noqa: BLE001
            errors.append(f"{k}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(32)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def _native_handled(http_url):
    import urllib.request

    with urllib.request.urlopen(f"http://{http_url}/metrics", timeout=10) as r:
        for line in r.read().decode().splitlines():
            if line.startswith("tca_native_handled_total "):
                return int(line.rsplit(" ", 1)[1])
    raise AssertionError("the server reports no native execution counters")


def test_native_executor_answers_and_survives_unregistration():
    """Requests with all tensors in device regions never reach the interpreter (the executor's
    counter moves, the model statistics still count them), requests the device refuses come back
    through the general path with its error text, and a region can be unregistered while other
    callers keep the executor busy."""
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm
    from tritonclient.utils import InferenceServerException, serialize_byte_tensor

    from triton_client_amd.server import spawn_server

    vals = np.array([str(7 * i).encode() for i in range(16)], dtype=object).reshape(1, 16)
    blob = serialize_byte_tensor(vals).item()
    srv = spawn_server(models=["simple_string"], gpu=0)
    try:
        stop = threading.Event()
        errors = []

        def load(k):
            try:
                with g.InferenceServerClient(srv.grpc_url) as c:
                    tag = f"bg{k}"
                    handles = _regions(hipshm, c, [f"{tag}_in0", f"{tag}_in1", f"{tag}_out0", f"{tag}_out1"], 1024)
                    try:
                        for n in ("in0", "in1"):
                            hipshm.set_shared_memory_region(handles[f"{tag}_{n}"], [serialize_byte_tensor(vals)])
                        ins, outs = _request(g, 1, tag, [len(blob), len(blob)], 1024)
                        while not stop.is_set():
                            c.infer("simple_string", ins, outputs=outs)
                    finally:
                        _cleanup(hipshm, c, handles)
            except Exception as e:  # noqa: BLE001
                errors.append(f"{k}: {type(e).__name__}: {e}")

        threads = [threading.Thread(target=load, args=(k,)) for k in range(4)]
        for t in threads:
            t.start()
        try:
            with g.InferenceServerClient(srv.grpc_url) as c:
                before = _native_handled(srv.http_url)
                for round_ in range(20):  # register, use, unregister while the others keep going
                    tag = f"fg{round_}"
                    handles = _regions(hipshm, c, [f"{tag}_in0", f"{tag}_in1", f"{tag}_out0", f"{tag}_out1"], 1024)
                    try:
                        for n in ("in0", "in1"):
                            hipshm.set_shared_memory_region(handles[f"{tag}_{n}"], [serialize_byte_tensor(vals)])
                        ins, outs = _request(g, 1, tag, [len(blob), len(blob)], 1024)
                        res = c.infer("simple_string", ins, outputs=outs, request_id=f"r{round_}")
                        assert res.get_response().id == f"r{round_}"
                        assert res.get_output("OUTPUT1").shape == [1, 16]
                        got = hipshm.get_contents_as_numpy(handles[f"{tag}_out0"], np.object_, [1, 16])
                        assert [int(x) for x in got.reshape(-1)] == [14 * i for i in range(16)]
                        if round_ == 5:  # refused by the device -> answered by the general path
                            small_in, small_out = _request(g, 1, tag, [len(blob), len(blob)], 16)
                            with pytest.raises(InferenceServerException, match="too small|should be at least"):
                                c.infer("simple_string", small_in, outputs=small_out)
                    finally:
                        _cleanup(hipshm, c, handles)
                assert _native_handled(srv.http_url) - before >= 20
                stats = c.get_inference_statistics("simple_string", as_json=True)["model_stats"][0]
                assert int(stats["inference_stats"]["success"]["count"]) >= 20
        finally:
            stop.set()
            for t in threads:
                t.join()
        assert not errors, errors[:3]
    finally:
        srv.stop()

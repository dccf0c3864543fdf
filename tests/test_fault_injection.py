"""Server-side fault knobs and the client behaviour they are there to exercise:
``delay_ms`` (latency), ``fail_every_n`` (request errors) and
``drop_stream_after_n`` (a gRPC stream that dies mid-flight), set as model
parameters through a load-with-config override.  The reference has no fault
injection; its clients' handling of a dead stream is
``src/python/library/tritonclient/grpc/_infer_stream.py:118-167``."""

import json
import queue
import time

import numpy as np
import pytest


def _params(**kv):
    return json.dumps({"parameters": {k: {"string_value": str(v)} for k, v in kv.items()}})


@pytest.fixture(params=["native", "grpcio"])
def srv(request):
    from triton_client_amd.server import Server

    with Server(models=["simple", "custom_identity_int32"], grpc_impl=request.param) as s:
        yield s


def test_fail_every_n_and_delay(srv):
    import tritonclient.http as h
    from tritonclient.utils import InferenceServerException

    x = np.arange(4, dtype=np.int32).reshape(1, 4)
    with h.InferenceServerClient(srv.http_url) as c:
        c.load_model("custom_identity_int32", config=_params(fail_every_n=3, delay_ms=30))
        inp = h.InferInput("INPUT0", [1, 4], "INT32")
        inp.set_data_from_numpy(x)
        outcomes = []
        t0 = time.perf_counter()
        for _ in range(6):
            try:
                c.infer("custom_identity_int32", [inp])
                outcomes.append("ok")
            except InferenceServerException as e:
                assert "injected failure" in str(e)
                outcomes.append("fail")
        assert outcomes == ["ok", "ok", "fail", "ok", "ok", "fail"]
        assert time.perf_counter() - t0 >= 6 * 0.03
        # the failures are visible in the statistics
        stats = c.get_inference_statistics("custom_identity_int32")["model_stats"][0]["inference_stats"]
        assert int(stats["fail"]["count"]) == 2 and int(stats["success"]["count"]) == 4
        c.load_model("custom_identity_int32")  # back to the defaults
        c.infer("custom_identity_int32", [inp])


def test_stream_dropped_by_the_server(srv):
    """After the injected failure the stream reports an error to the callback and
    refuses further requests; a fresh stream works again."""
    import tritonclient.grpc as g
    from tritonclient.utils import InferenceServerException

    x = np.arange(4, dtype=np.int32).reshape(1, 4)
    with g.InferenceServerClient(srv.grpc_url) as c:
        c.load_model("custom_identity_int32", config=_params(drop_stream_after_n=2))
        got = queue.Queue()
        c.start_stream(callback=lambda result, error: got.put((result, error)))
        inp = g.InferInput("INPUT0", [1, 4], "INT32")
        inp.set_data_from_numpy(x)
        for i in range(2):
            c.async_stream_infer("custom_identity_int32", [inp], request_id=str(i))
        results, errors = [], []
        deadline = time.time() + 20
        while time.time() < deadline and not errors:
            try:
                result, error = got.get(timeout=0.5)
            except queue.Empty:
                continue
            (errors if error is not None else results).append(error if error is not None else result)
        assert len(results) == 2 and errors, "two answers, then the stream must fail"
        assert "injected stream failure" in str(errors[0]) or "UNAVAILABLE" in str(errors[0])
        # the dead stream refuses new work ...
        with pytest.raises(InferenceServerException):
            for _ in range(50):
                c.async_stream_infer("custom_identity_int32", [inp])
                time.sleep(0.02)
        c.stop_stream()
        # ... and the client recovers with a new stream once the knob is cleared
        c.load_model("custom_identity_int32")
        got2 = queue.Queue()
        c.start_stream(callback=lambda result, error: got2.put((result, error)))
        c.async_stream_infer("custom_identity_int32", [inp], request_id="again")
        result, error = got2.get(timeout=20)
        assert error is None
        np.testing.assert_array_equal(result.as_numpy("OUTPUT0"), x)
        c.stop_stream()

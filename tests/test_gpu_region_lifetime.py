"""Device regions versus in-flight GPU work on a real MI355X: unregistering a region while
requests that read / write it are queued or running, many distinct regions through one
server (captured graphs are bounded and evicted graphs are destroyed only after the stream
has passed them), and an ensemble whose input tensors are fresh on every request."""

import io
import os
import queue

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

B = 8
IN_BYTES, OUT_BYTES = B * 3 * 224 * 224 * 2, B * 1000 * 4


@pytest.fixture(scope="module")
def served():
    from triton_client_amd.server import spawn_server

    # a small graph cache so that LRU eviction happens within the test
    srv = spawn_server(models=["resnet50"], gpu=0, env={"TCA_RESNET_GRAPH_CACHE": "48"})
    yield srv
    srv.stop()


@pytest.fixture(scope="module")
def served_general():
    """The same server with the native graph path off: every request takes the general (Python) path."""
    from triton_client_amd.server import spawn_server

    srv = spawn_server(models=["resnet50"], gpu=0, env={"TCA_RESNET_GRAPH_CACHE": "48", "TCA_NATIVE_GRAPHS": "0"})
    yield srv
    srv.stop()


def _make_regions(hipshm, client, tag, n, x):
    regions = []
    for i in range(n):
        hin = hipshm.create_shared_memory_region(f"{tag}_in{i}", IN_BYTES, 0)
        hout = hipshm.create_shared_memory_region(f"{tag}_out{i}", OUT_BYTES, 0)
        hipshm.set_shared_memory_region(hin, [x])
        client.register_cuda_shared_memory(f"{tag}_in{i}", hipshm.get_raw_handle(hin), 0, IN_BYTES)
        client.register_cuda_shared_memory(f"{tag}_out{i}", hipshm.get_raw_handle(hout), 0, OUT_BYTES)
        regions.append((hin, hout))
    return regions


def _request(g, tag, i):
    inp = g.InferInput("input", [B, 3, 224, 224], "FP16")
    inp.set_shared_memory(f"{tag}_in{i}", IN_BYTES)
    out = g.InferRequestedOutput("output")
    out.set_shared_memory(f"{tag}_out{i}", OUT_BYTES)
    return [inp], [out]


def test_unregister_while_32_requests_are_in_flight(served):
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm

    x = np.random.default_rng(0).standard_normal((B, 3, 224, 224)).astype(np.float16)
    with g.InferenceServerClient(served.grpc_url) as c, g.InferenceServerClient(served.grpc_url) as ctl:
        regions = _make_regions(hipshm, c, "life", 32, x)
        # reference logits through a region that stays registered
        ins, outs = _request(g, "life", 0)
        c.infer("resnet50", ins, outputs=outs)
        ref = hipshm.get_contents_as_numpy(regions[0][1], np.float32, [B, 1000]).copy()
        for _ in range(3):  # every round: 32 requests go out, then everything is unregistered under them
            done = queue.Queue()
            for i in range(32):
                ins, outs = _request(g, "life", i)
                c.async_infer("resnet50", ins, lambda result, error: done.put(error), outputs=outs)
            ctl.unregister_cuda_shared_memory()  # all regions; waits for the requests that hold them
            errors = [done.get(timeout=60) for _ in range(32)]
            # a request either ran to completion or was refused because its region was already going away
            for e in errors:
                assert e is None or "Unable to find shared memory region" in str(e), e
            assert sum(e is None for e in errors) >= 1
            assert ctl.get_cuda_shared_memory_status(as_json=True).get("regions", {}) == {}
            # the server is alive and the regions can come back
            for i, (hin, hout) in enumerate(regions):
                c.register_cuda_shared_memory(f"life_in{i}", hipshm.get_raw_handle(hin), 0, IN_BYTES)
                c.register_cuda_shared_memory(f"life_out{i}", hipshm.get_raw_handle(hout), 0, OUT_BYTES)
        ins, outs = _request(g, "life", 5)
        c.infer("resnet50", ins, outputs=outs)
        np.testing.assert_allclose(hipshm.get_contents_as_numpy(regions[5][1], np.float32, [B, 1000]), ref,
                                   rtol=1e-2, atol=1e-2)
        c.unregister_cuda_shared_memory()
        for hin, hout in regions:
            hipshm.destroy_shared_memory_region(hin)
            hipshm.destroy_shared_memory_region(hout)


def test_600_distinct_regions_through_one_server(served):
    """Every region pair is a new (input address, output address): a new captured graph.  The
    cache keeps 48 of them; device memory in use must not grow with the number of regions seen
    beyond the regions themselves."""
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm

    n = int(os.environ.get("TCA_TEST_REGION_COUNT", "600"))
    x = np.random.default_rng(1).standard_normal((B, 3, 224, 224)).astype(np.float16)
    with g.InferenceServerClient(served.grpc_url) as c:
        regions = _make_regions(hipshm, c, "many", n, x)
        used = []
        ref = None
        for i in range(n):
            ins, outs = _request(g, "many", i)
            c.infer("resnet50", ins, outputs=outs)
            if i in (0, n // 2, n - 1):
                got = hipshm.get_contents_as_numpy(regions[i][1], np.float32, [B, 1000])
                ref = got.copy() if ref is None else ref
                np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)
            if i in (n // 3, n - 1):
                free, total = hipshm.device_memory_info(0)
                used.append(total - free)
        # second pass over regions whose graphs were evicted long ago: recaptured, same answers
        for i in (0, 1, 2):
            ins, outs = _request(g, "many", i)
            c.infer("resnet50", ins, outputs=outs)
            np.testing.assert_allclose(hipshm.get_contents_as_numpy(regions[i][1], np.float32, [B, 1000]), ref,
                                       rtol=1e-2, atol=1e-2)
        assert used[1] - used[0] < (256 << 20), f"device memory grew by {(used[1] - used[0]) >> 20} MiB"
        c.unregister_cuda_shared_memory()
        for hin, hout in regions:
            hipshm.destroy_shared_memory_region(hin)
            hipshm.destroy_shared_memory_region(hout)


def test_ensemble_requests_with_fresh_tensors_leave_memory_flat():
    """The ensemble produces a new input tensor for the classifier on every request; it must
    run through the classifier's static buffers, not capture (and keep) a graph per request."""
    import torch
    from PIL import Image

    import tritonclient.grpc as g
    from triton_client_amd.models.vision import _LOADED
    from triton_client_amd.server import Server

    rng = np.random.default_rng(2)
    blobs = []
    for _ in range(2):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8)).save(buf, format="PNG")
        blobs.append(buf.getvalue())
    data = np.array(blobs, dtype=np.object_).reshape(2, 1)
    with Server(models=["resnet50", "preprocess_resnet50_ensemble"], gpu=0) as srv:
        net = _LOADED["resnet50"]
        with g.InferenceServerClient(srv.grpc_url) as c:
            inp = g.InferInput("INPUT", [2, 1], "BYTES")
            inp.set_data_from_numpy(data)
            first = c.infer("preprocess_resnet50_ensemble", [inp]).as_numpy("OUTPUT")
            graphs, reserved = net.graph_count(), torch.cuda.memory_reserved()
            for _ in range(600):
                got = c.infer("preprocess_resnet50_ensemble", [inp]).as_numpy("OUTPUT")
            np.testing.assert_allclose(got, first, rtol=1e-2, atol=1e-2)
            assert net.graph_count() == graphs
            assert torch.cuda.memory_reserved() - reserved <= (8 << 20)


def _plan_counters(served):
    import urllib.request

    text = urllib.request.urlopen(f"http://{served.http_url}/metrics", timeout=5).read().decode()
    return {line.split()[0]: int(line.split()[1]) for line in text.splitlines() if line.startswith("tca_request_plan_")}


@pytest.mark.parametrize("path", ["native", "general"])
def test_repeated_messages_run_from_their_plan_on_fresh_contents(request, path):
    """The same raw request message again and again: every repeat still reads what is in the region NOW and
    writes its own answer, and a region that comes back under the same name with other memory behind it is
    addressed through the new mapping.  On the general path the repeats are queued from the stored plan of
    the message (counted); the native graph path parses every message and looks the regions up again."""
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm

    served = request.getfixturevalue("served" if path == "native" else "served_general")

    rng = np.random.default_rng(7)
    x1 = rng.standard_normal((B, 3, 224, 224)).astype(np.float16)
    x2 = rng.standard_normal((B, 3, 224, 224)).astype(np.float16)

    def read(handle):
        return hipshm.get_contents_as_numpy(handle, np.float32, [B, 1000]).copy()

    with g.InferenceServerClient(served.grpc_url) as c:
        def wire(x):  # reference: the same tensors in the message itself
            inp = g.InferInput("input", [B, 3, 224, 224], "FP16")
            inp.set_data_from_numpy(x)
            return c.infer("resnet50", [inp]).as_numpy("output")

        ref1, ref2 = wire(x1), wire(x2)
        assert np.abs(ref1 - ref2).max() > 1e-2
        (hin, hout), = _make_regions(hipshm, c, "plan", 1, x1)
        before = _plan_counters(served)
        ins, outs = _request(g, "plan", 0)
        for x, ref in ((x1, ref1), (x2, ref2), (x1, ref1), (x2, ref2)):
            hipshm.set_shared_memory_region(hin, [x])
            hipshm.set_shared_memory_region(hout, [np.full((B, 1000), -7.0, np.float32)])
            c.infer("resnet50", ins, outputs=outs)
            np.testing.assert_allclose(read(hout), ref, rtol=1e-2, atol=1e-2)
        mid = _plan_counters(served)
        if path == "general":
            assert mid["tca_request_plan_stored_total"] == before["tca_request_plan_stored_total"] + 1
            assert mid["tca_request_plan_hits_total"] == before["tca_request_plan_hits_total"] + 3
        else:
            assert mid == before  # none of the four reached the interpreter
        # the same names, other memory: the old plan must not be used (its views alias the old mapping)
        c.unregister_cuda_shared_memory("plan_in0")
        c.unregister_cuda_shared_memory("plan_out0")
        hipshm.set_shared_memory_region(hout, [np.full((B, 1000), -7.0, np.float32)])
        hin2 = hipshm.create_shared_memory_region("plan_in0_b", IN_BYTES, 0)
        hout2 = hipshm.create_shared_memory_region("plan_out0_b", OUT_BYTES, 0)
        hipshm.set_shared_memory_region(hin2, [x2])
        c.register_cuda_shared_memory("plan_in0", hipshm.get_raw_handle(hin2), 0, IN_BYTES)
        c.register_cuda_shared_memory("plan_out0", hipshm.get_raw_handle(hout2), 0, OUT_BYTES)
        for _ in range(2):
            c.infer("resnet50", ins, outputs=outs)
            np.testing.assert_allclose(read(hout2), ref2, rtol=1e-2, atol=1e-2)
        assert np.all(read(hout) == -7.0)  # nothing wrote through the old mapping
        after = _plan_counters(served)
        if path == "general":
            assert after["tca_request_plan_stored_total"] == mid["tca_request_plan_stored_total"] + 1
            assert after["tca_request_plan_hits_total"] == mid["tca_request_plan_hits_total"] + 1
        c.unregister_cuda_shared_memory("plan_in0")
        c.unregister_cuda_shared_memory("plan_out0")
        for h in (hin, hout, hin2, hout2):
            hipshm.destroy_shared_memory_region(h)


def test_repeated_rest_bodies_run_from_their_plan(served_general):
    """The REST front keeps plans too (small JSON bodies whose tensors are region references)."""
    served = served_general
    import tritonclient.http as h
    import tritonclient.utils.cuda_shared_memory as hipshm

    rng = np.random.default_rng(11)
    x1 = rng.standard_normal((B, 3, 224, 224)).astype(np.float16)
    x2 = rng.standard_normal((B, 3, 224, 224)).astype(np.float16)
    with h.InferenceServerClient(served.http_url) as c:
        hin = hipshm.create_shared_memory_region("rest_plan_in", IN_BYTES, 0)
        hout = hipshm.create_shared_memory_region("rest_plan_out", OUT_BYTES, 0)
        c.register_cuda_shared_memory("rest_plan_in", hipshm.get_raw_handle(hin), 0, IN_BYTES)
        c.register_cuda_shared_memory("rest_plan_out", hipshm.get_raw_handle(hout), 0, OUT_BYTES)
        inp = h.InferInput("input", [B, 3, 224, 224], "FP16")
        inp.set_shared_memory("rest_plan_in", IN_BYTES)
        out = h.InferRequestedOutput("output")
        out.set_shared_memory("rest_plan_out", OUT_BYTES)
        wire = h.InferInput("input", [B, 3, 224, 224], "FP16")
        before = _plan_counters(served)
        for x in (x1, x2, x1):
            wire.set_data_from_numpy(x)
            ref = c.infer("resnet50", [wire]).as_numpy("output")  # tensors in the body: never planned
            hipshm.set_shared_memory_region(hin, [x])
            result = c.infer("resnet50", [inp], outputs=[out])
            assert result.get_output("output")["parameters"]["shared_memory_region"] == "rest_plan_out"
            got = hipshm.get_contents_as_numpy(hout, np.float32, [B, 1000])
            np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)
        after = _plan_counters(served)
        assert after["tca_request_plan_stored_total"] == before["tca_request_plan_stored_total"] + 1
        assert after["tca_request_plan_hits_total"] == before["tca_request_plan_hits_total"] + 2
        c.unregister_cuda_shared_memory("rest_plan_in")
        c.unregister_cuda_shared_memory("rest_plan_out")
        hipshm.destroy_shared_memory_region(hin)
        hipshm.destroy_shared_memory_region(hout)

"""ResNet-50 served by the native executor from captured graphs (requests with both tensors in HIP shared
memory never reach the interpreter): distinct inputs in distinct regions come back as distinct, correct
outputs whatever the request batch sizes and however they were merged and padded; nothing is captured
online; the counters and the statistics see the requests; the general path (``TCA_NATIVE_GRAPHS=0``)
agrees.  Reference for every output: a forward of the same input by a model instance in this process
(same seeded weights), itself checked against fp32 in ``test_gpu_model.py``."""

import queue
import urllib.request

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

ROW_IN, ROW_OUT = 3 * 224 * 224 * 2, 1000 * 4
SLOTS = 64
# request batch sizes of the slots: 1 (perf_analyzer's default), odd ones, the captured sizes and beyond 8
BATCHES = [(1, 3, 8, 5, 2, 16, 7, 4)[i % 8] for i in range(SLOTS)]


def _inputs():
    """One distinct fp16 input per slot: smooth patterns of different orientation, frequency and offset, so
    that the logits of different slots differ by far more than fp16 rounding."""
    yy, xx = np.mgrid[0:224, 0:224].astype(np.float32) / 224.0
    out = []
    for i, b in enumerate(BATCHES):
        rows = []
        for j in range(b):
            k = 17 * i + j
            planes = [np.sin((1 + k % 5) * 3.1 * xx + (k % 7) * yy + c) * (0.4 + 0.1 * (k % 4)) + 0.15 * ((k + c) % 9 - 4)
                      for c in range(3)]
            rows.append(np.stack(planes))
        out.append(np.stack(rows).astype(np.float16))
    return out


@pytest.fixture(scope="module")
def reference():
    import torch

    from triton_client_amd.models import MODEL_ZOO

    xs = _inputs()
    m = MODEL_ZOO["resnet50"]()
    m.load(0)
    refs = []
    for x in xs:
        out = m.run(torch.from_numpy(x).cuda())
        m.stream.synchronize()
        refs.append(out.float().cpu().numpy())
    m.unload()
    return xs, refs


def _serve_and_check(env, xs, refs, protocol):
    import tritonclient.grpc as g
    import tritonclient.http as h
    import tritonclient.utils.cuda_shared_memory as hipshm

    from triton_client_amd.server import spawn_server

    srv = spawn_server(models=["resnet50"], gpu=0, env=env)
    regions = []
    try:
        mod = g if protocol == "grpc" else h
        url = srv.grpc_url if protocol == "grpc" else srv.http_url
        kwargs = {} if protocol == "grpc" else {"concurrency": 16}
        with mod.InferenceServerClient(url, **kwargs) as c:
            for i, x in enumerate(xs):
                b = BATCHES[i]
                hin = hipshm.create_shared_memory_region(f"ng_in{i}", b * ROW_IN, 0)
                hout = hipshm.create_shared_memory_region(f"ng_out{i}", b * ROW_OUT, 0)
                hipshm.set_shared_memory_region(hin, [x])
                c.register_cuda_shared_memory(f"ng_in{i}", hipshm.get_raw_handle(hin), 0, b * ROW_IN)
                c.register_cuda_shared_memory(f"ng_out{i}", hipshm.get_raw_handle(hout), 0, b * ROW_OUT)
                regions.append((hin, hout))

            def request(i):
                b = BATCHES[i]
                inp = mod.InferInput("input", [b, 3, 224, 224], "FP16")
                inp.set_shared_memory(f"ng_in{i}", b * ROW_IN)
                out = mod.InferRequestedOutput("output")
                out.set_shared_memory(f"ng_out{i}", b * ROW_OUT)
                return [inp], [out]

            for _ in range(3):  # three rounds of all 64 at once: merged and padded differently every time
                if protocol == "grpc":
                    done = queue.Queue()
                    for i in range(SLOTS):
                        ins, outs = request(i)
                        c.async_infer("resnet50", ins, lambda result, error: done.put(error), outputs=outs)
                    errors = [done.get(timeout=120) for _ in range(SLOTS)]
                    assert all(e is None for e in errors), errors
                else:
                    pending = [c.async_infer("resnet50", *request(i)[:1], outputs=request(i)[1]) for i in range(SLOTS)]
                    for p in pending:
                        p.get_result(timeout=120)
            outs = [np.array(hipshm.get_contents_as_numpy(regions[i][1], np.float32, [BATCHES[i], 1000]), copy=True)
                    for i in range(SLOTS)]
            stats = c.get_inference_statistics("resnet50") if protocol == "http" else \
                c.get_inference_statistics("resnet50", as_json=True)
            c.unregister_cuda_shared_memory()
        with urllib.request.urlopen(f"http://{srv.http_url}/metrics", timeout=30) as resp:
            metrics = resp.read().decode()
    finally:
        srv.stop()
        for hin, hout in regions:
            hipshm.destroy_shared_memory_region(hin)
            hipshm.destroy_shared_memory_region(hout)
    scale = max(float(np.abs(r).max()) for r in refs)
    worst = max(float(np.abs(o - r).max()) for o, r in zip(outs, refs))
    assert worst < 2e-2 * scale, (worst, scale)  # other batch sizes run other tile variants: fp16 rounding only
    # distinct slots stayed distinct: the closest pair of first rows is far apart compared with that error
    firsts = np.stack([o[0] for o in outs])
    apart = min(float(np.abs(firsts[i] - firsts[j]).max()) for i in range(SLOTS) for j in range(i + 1, SLOTS))
    assert apart > 4 * worst, (apart, worst)
    for o, r in zip(outs, refs):
        assert (o.argmax(1) == r.argmax(1)).mean() >= 0.9
    return metrics, stats


def _metric(text, name):
    for line in text.splitlines():
        if line.startswith(name):
            return float(line.rsplit(" ", 1)[1])
    return None


@pytest.mark.parametrize("protocol", ["grpc", "http"])
def test_distinct_inputs_any_batch_size_native(reference, protocol):
    xs, refs = reference
    metrics, stats = _serve_and_check({}, xs, refs, protocol)
    batches = _metric(metrics, 'tca_native_graph_batches_total{model="resnet50"}')
    requests = _metric(metrics, 'tca_native_graph_requests_total{model="resnet50"}')
    images = _metric(metrics, 'tca_native_graph_images_total{model="resnet50"}')
    assert requests == 3 * SLOTS and images == 3 * sum(BATCHES) and 1 <= batches < requests  # merged, all native
    assert _metric(metrics, 'tca_native_graph_padded_images_total{model="resnet50"}') is not None
    assert _metric(metrics, 'tca_graph_captures_online_total{model="resnet50"}') == 0  # every size served from load-time graphs
    row = stats["model_stats"][0]
    assert int(row["inference_count"]) == 3 * sum(BATCHES)


def test_general_path_agrees(reference):
    xs, refs = reference
    metrics, _ = _serve_and_check({"TCA_NATIVE_GRAPHS": "0"}, xs, refs, "grpc")
    assert _metric(metrics, 'tca_native_graph_batches_total{model="resnet50"}') is None


@pytest.mark.parametrize("protocol", ["grpc", "http"])
def test_tensors_inside_the_messages_are_served_natively(reference, protocol):
    """Requests that carry the input inside the message and get the output rows back inside the response,
    the two mixed forms (input in a device region / output in the response, input in the message / output in
    a device region) and requests with both tensors in system shared memory run from the same graphs:
    staged through page-locked memory, no interpreter, same answers."""
    import tritonclient.grpc as g
    import tritonclient.http as h
    import tritonclient.utils.cuda_shared_memory as hipshm

    from triton_client_amd.server import spawn_server

    import tritonclient.utils.shared_memory as sysshm

    xs, refs = reference
    n = 24
    srv = spawn_server(models=["resnet50"], gpu=0)
    regions, sys_regions = [], {}
    try:
        mod = g if protocol == "grpc" else h
        url = srv.grpc_url if protocol == "grpc" else srv.http_url
        kwargs = {} if protocol == "grpc" else {"concurrency": 16}
        with mod.InferenceServerClient(url, **kwargs) as c:
            for i in range(n):
                b = BATCHES[i]
                hin = hipshm.create_shared_memory_region(f"nw_in{i}", b * ROW_IN, 0)
                hout = hipshm.create_shared_memory_region(f"nw_out{i}", b * ROW_OUT, 0)
                hipshm.set_shared_memory_region(hin, [xs[i]])
                c.register_cuda_shared_memory(f"nw_in{i}", hipshm.get_raw_handle(hin), 0, b * ROW_IN)
                c.register_cuda_shared_memory(f"nw_out{i}", hipshm.get_raw_handle(hout), 0, b * ROW_OUT)
                regions.append((hin, hout))
                if i % 4 == 3:
                    hs = sysshm.create_shared_memory_region(f"nw_sys{i}", f"/tca_nw_sys{i}", b * (ROW_IN + ROW_OUT))
                    sysshm.set_shared_memory_region(hs, [xs[i]])
                    c.register_system_shared_memory(f"nw_sys{i}", f"/tca_nw_sys{i}", b * (ROW_IN + ROW_OUT))
                    sys_regions[i] = hs

            def request(i):
                # 0: both inside the messages, 1: device region in / wire out, 2: wire in / device region out,
                # 3: both in system shared memory
                b, form = BATCHES[i], i % 4
                inp = mod.InferInput("input", [b, 3, 224, 224], "FP16")
                if form == 1:
                    inp.set_shared_memory(f"nw_in{i}", b * ROW_IN)
                elif form == 3:
                    inp.set_shared_memory(f"nw_sys{i}", b * ROW_IN)
                else:
                    inp.set_data_from_numpy(xs[i])
                out = mod.InferRequestedOutput("output")
                if form == 2:
                    out.set_shared_memory(f"nw_out{i}", b * ROW_OUT)
                elif form == 3:
                    out.set_shared_memory(f"nw_sys{i}", b * ROW_OUT, offset=b * ROW_IN)
                return [inp], [out]

            results = [None] * n
            for _ in range(2):
                if protocol == "grpc":
                    done = queue.Queue()
                    for i in range(n):
                        ins, outs = request(i)
                        c.async_infer("resnet50", ins, lambda result, error, i=i: done.put((i, result, error)), outputs=outs)
                    for _ in range(n):
                        i, result, error = done.get(timeout=120)
                        assert error is None, error
                        results[i] = result
                else:
                    pending = [(i, c.async_infer("resnet50", request(i)[0], outputs=request(i)[1])) for i in range(n)]
                    for i, p in pending:
                        results[i] = p.get_result(timeout=120)
            outs = []
            for i in range(n):
                if i % 4 == 2:
                    outs.append(np.array(hipshm.get_contents_as_numpy(regions[i][1], np.float32, [BATCHES[i], 1000]), copy=True))
                elif i % 4 == 3:
                    outs.append(np.array(sysshm.get_contents_as_numpy(sys_regions[i], np.float32, [BATCHES[i], 1000],
                                                                      offset=BATCHES[i] * ROW_IN), copy=True))
                else:
                    outs.append(results[i].as_numpy("output"))
            c.unregister_cuda_shared_memory()
            c.unregister_system_shared_memory()
        with urllib.request.urlopen(f"http://{srv.http_url}/metrics", timeout=30) as resp:
            metrics = resp.read().decode()
    finally:
        srv.stop()
        for hin, hout in regions:
            hipshm.destroy_shared_memory_region(hin)
            hipshm.destroy_shared_memory_region(hout)
        for hs in sys_regions.values():
            sysshm.destroy_shared_memory_region(hs)
    scale = max(float(np.abs(r).max()) for r in refs[:n])
    worst = max(float(np.abs(o - r).max()) for o, r in zip(outs, refs[:n]))
    assert all(o.shape == r.shape for o, r in zip(outs, refs[:n])) and worst < 2e-2 * scale, (worst, scale)
    assert _metric(metrics, 'tca_native_graph_requests_total{model="resnet50"}') == 2 * n  # none took the general path
    assert _metric(metrics, 'tca_native_graph_wire_requests_total{model="resnet50"}') == 2 * n
    assert _metric(metrics, 'tca_native_graph_arena_full_total{model="resnet50"}') == 0
    assert _metric(metrics, 'tca_graph_captures_online_total{model="resnet50"}') == 0

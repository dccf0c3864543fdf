"""``resnet50_sharded`` served from one process with two replica processes (all three on GPU 0,
real hipIpc mappings, flags polled by the GPU): a batch that arrives in device shared memory is
split, classified by the three replicas and gathered into the caller's output region; the
logits equal those of the plain single-GPU model for the same batch."""

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.fixture(scope="module")
def served():
    from triton_client_amd.server import spawn_server

    srv = spawn_server(models=["resnet50", "resnet50_sharded"], gpu=0,
                       env={"TCA_SHARD_DEVICES": "0,0", "TCA_SHARD_PER_GPU_BATCH": "32",
                            "TCA_RESNET_MAX_BATCH": "32"}, startup_timeout=600)
    yield srv
    srv.stop()


@pytest.mark.parametrize("batch", [48, 96, 50])
def test_sharded_logits_equal_single_gpu(served, batch):
    import tritonclient.grpc as g
    import tritonclient.utils.cuda_shared_memory as hipshm

    in_bytes, out_bytes = batch * 3 * 224 * 224 * 2, batch * 1000 * 4
    x = np.random.default_rng(batch).standard_normal((batch, 3, 224, 224)).astype(np.float16)
    hin = hipshm.create_shared_memory_region("sh_in", in_bytes, 0)
    hout = hipshm.create_shared_memory_region("sh_out", out_bytes, 0)
    try:
        hipshm.set_shared_memory_region(hin, [x])
        with g.InferenceServerClient(served.grpc_url) as c:
            c.register_cuda_shared_memory("sh_in", hipshm.get_raw_handle(hin), 0, in_bytes)
            c.register_cuda_shared_memory("sh_out", hipshm.get_raw_handle(hout), 0, out_bytes)
            inp = g.InferInput("input", [batch, 3, 224, 224], "FP16")
            inp.set_shared_memory("sh_in", in_bytes)
            out = g.InferRequestedOutput("output")
            out.set_shared_memory("sh_out", out_bytes)
            meta = c.get_model_config("resnet50_sharded", as_json=True)["config"]
            assert int(meta["max_batch_size"]) == 96
            for _ in range(3):  # both slots, buffers reused
                hipshm.set_shared_memory_region(hout, [np.zeros((batch, 1000), np.float32)])
                c.infer("resnet50_sharded", [inp], outputs=[out])
                sharded = hipshm.get_contents_as_numpy(hout, np.float32, [batch, 1000]).copy()
                assert np.isfinite(sharded).all() and np.abs(sharded).max() > 0
            # single-GPU reference: the plain model on the same rows, in the chunks the group
            # forms (same batch -> same kernels -> the same bits, whichever replica ran them)
            ref = np.empty_like(sharded)
            start = 0
            for r in range(3):
                n = batch // 3 + (1 if r < batch % 3 else 0)
                res = c.infer("resnet50", [g.InferInput("input", [n, 3, 224, 224], "FP16").set_data_from_numpy(
                    x[start:start + n])])
                ref[start:start + n] = res.as_numpy("output")
                start += n
            np.testing.assert_allclose(sharded, ref, rtol=1e-3, atol=1e-3)
            c.unregister_cuda_shared_memory()
    finally:
        hipshm.destroy_shared_memory_region(hin)
        hipshm.destroy_shared_memory_region(hout)

"""ResNet-50 on the GPU: the fused forward (MFMA 1x1 convolutions with bias /
residual / ReLU epilogues + one-pass epilogues behind library convolutions)
against the plain library forward of the same weights and an fp32 reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from triton_client_amd.models import MODEL_ZOO

    m = MODEL_ZOO["resnet50"]()
    m.load(0)
    yield m
    m.unload()


@pytest.mark.parametrize("batch", [1, 8, 24])
def test_fused_forward_matches_library_and_fp32(model, batch):
    import torch

    x = torch.randn(batch, 3, 224, 224, device="cuda", dtype=torch.float16)
    with torch.no_grad(), torch.cuda.stream(model.stream):
        model.fused = True
        fused = model._forward_eager(x)
        model.fused = False
        plain = model._forward_eager(x)
        model.fused = True
        ref = model._net.float()(x.float().contiguous(memory_format=torch.channels_last))
        model._net.half()
    model.stream.synchronize()
    assert fused.shape == (batch, 1000) and torch.isfinite(fused).all()
    scale = ref.abs().max().item()
    # both fp16 pipelines must sit equally close to the fp32 reference
    err_fused = (fused - ref).abs().max().item() / scale
    err_plain = (plain - ref).abs().max().item() / scale
    assert err_fused < 2e-2, (err_fused, err_plain)
    assert err_fused < 2.0 * err_plain + 2e-3, (err_fused, err_plain)
    assert (fused.argmax(1) == ref.argmax(1)).float().mean().item() >= 0.75


@pytest.mark.parametrize("batch", [88, 128])
@pytest.mark.parametrize("tuning", ["1", "0"])
def test_large_batches_tuned_and_untuned_against_fp32(batch, tuning, monkeypatch):
    """The sizes the serving headline replays (merged batches of 88 .. 128 on the tuned dispatch table: 256-wide
    tiles on 16 wavefronts, fused and attached blocks, the row-organised stage-1 kernel) and the same sizes
    with the table off, as captured graphs on BOTH execution instances at once, against the fp32 network:
    relative error below 2e-2, within 2x of the library forward, and the two instances bit-identical."""
    import torch

    from triton_client_amd.models import MODEL_ZOO
    from triton_client_amd.ops import conv

    monkeypatch.setenv("TCA_CONV_TUNING", tuning)
    monkeypatch.setattr(conv, "_tuning", None)
    m = MODEL_ZOO["resnet50"]()
    m.load(0)
    try:
        gen = torch.Generator(device="cuda").manual_seed(batch)
        x = torch.randn(batch, 3, 224, 224, device="cuda", generator=gen).half()
        outs = [torch.empty(batch, 1000, device="cuda", dtype=torch.float32) for _ in range(m.instance_count)]
        for _ in range(2):  # capture, then replay: both instances enqueue before either is waited for
            for inst, out in enumerate(outs):
                m.run(x, out, inst, stable=True)
        for inst in range(m.instance_count):
            m.stream_of(inst).synchronize()
        with torch.no_grad(), torch.cuda.stream(m.stream):
            m.fused = False
            plain = m._forward_eager(x)
            m.fused = True
            ref = m._net.float()(x.float().contiguous(memory_format=torch.channels_last))
            m._net.half()
        m.stream.synchronize()
        scale = ref.abs().max().item()
        err = (outs[0] - ref).abs().max().item() / scale
        err_plain = (plain - ref).abs().max().item() / scale
        assert err < 2e-2 and err < 2.0 * err_plain + 2e-3, (err, err_plain)
        assert (outs[0].argmax(1) == ref.argmax(1)).float().mean().item() >= 0.9
        for other in outs[1:]:
            assert torch.equal(other, outs[0])
    finally:
        m.unload()
        monkeypatch.setattr(conv, "_tuning", None)


def test_graph_replay_equals_eager(model):
    import torch

    x = torch.randn(8, 3, 224, 224, device="cuda", dtype=torch.float16)
    out = torch.empty(8, 1000, device="cuda", dtype=torch.float32)
    model.run(x, out, stable=True)
    model.run(x, out, stable=True)  # replay
    model.stream.synchronize()
    with torch.no_grad(), torch.cuda.stream(model.stream):
        eager = model._forward_eager(x)
    model.stream.synchronize()
    # library kernels that split K accumulate with atomics, so two runs may differ by an fp16 ulp
    np.testing.assert_allclose(out.cpu().numpy(), eager.cpu().numpy(), rtol=1e-2, atol=1e-2)


def test_graphs_of_one_instance_share_their_activation_memory(model):
    """Every (batch, input buffer, output buffer) gets its own captured graph (zero-copy requests
    run straight out of the client's region), but the graphs of one execution instance replay one
    after the other on one stream: their activations must come from one pool, so that capturing
    for new buffers does not grow the footprint - and interleaved replays stay correct."""
    import torch

    xs = [torch.randn(8, 3, 224, 224, device="cuda", dtype=torch.float16) for _ in range(6)]
    outs = [torch.empty(8, 1000, device="cuda", dtype=torch.float32) for _ in range(6)]
    big_x = torch.randn(32, 3, 224, 224, device="cuda", dtype=torch.float16)
    big_out = torch.empty(32, 1000, device="cuda", dtype=torch.float32)
    model.run(big_x, big_out, stable=True)  # the largest batch sizes the pool
    model.run(xs[0], outs[0], stable=True)
    model.stream.synchronize()
    before = torch.cuda.memory_reserved()
    for x, out in zip(xs[1:], outs[1:]):
        model.run(x, out, stable=True)
    model.stream.synchronize()
    grown = torch.cuda.memory_reserved() - before
    assert grown <= 8 << 20, f"five more batch-8 graphs reserved {grown >> 20} MiB"
    # replay in another order than captured, the large graph in between
    for i in (3, 0, 5):
        model.run(xs[i], outs[i], stable=True)
        model.run(big_x, big_out, stable=True)
    model.stream.synchronize()
    with torch.no_grad(), torch.cuda.stream(model.stream):
        eager = [model._forward_eager(xs[i]) for i in (3, 0, 5)]
        eager_big = model._forward_eager(big_x)
    model.stream.synchronize()
    for i, ref in zip((3, 0, 5), eager):
        np.testing.assert_allclose(outs[i].cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(big_out.cpu().numpy(), eager_big.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
def test_gpu_ensemble_matches_client_side_preprocessing():
    """Encoded images through ``preprocess_resnet50_ensemble`` (host decode, HIP
    preprocessing, ResNet-50) give the logits of the same images preprocessed on
    the client with the Pillow pipeline and sent to ``resnet50`` directly."""
    import io

    import numpy as np
    from PIL import Image

    import tritonclient.grpc as g
    from triton_client_amd.ops.preprocess_ref import reference_preprocess
    from triton_client_amd.server import Server

    rng = np.random.default_rng(7)
    pictures = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in ((300, 400), (224, 224), (97, 131))]
    blobs = []
    for pic in pictures:
        buf = io.BytesIO()
        Image.fromarray(pic).save(buf, format="PNG")  # lossless: both paths see the same pixels
        blobs.append(buf.getvalue())
    with Server(models=["resnet50", "preprocess_resnet50_ensemble"], gpu=0) as srv:
        with g.InferenceServerClient(srv.grpc_url) as c:
            assert c.is_model_ready("preprocess_resnet50_ensemble")
            cfg = c.get_model_config("preprocess_resnet50_ensemble", as_json=True)["config"]
            assert cfg["platform"] == "ensemble" and len(cfg["ensemble_scheduling"]["step"]) == 2
            raw = np.array(blobs, dtype=np.object_).reshape(3, 1)
            inp = g.InferInput("INPUT", [3, 1], "BYTES").set_data_from_numpy(raw)
            got = c.infer("preprocess_resnet50_ensemble", [inp]).as_numpy("OUTPUT")
            x = np.stack([reference_preprocess(p, 224, 224, 3, "NCHW", np.float16, "INCEPTION",
                                               "pil") for p in pictures])
            direct = g.InferInput("input", [3, 3, 224, 224], "FP16").set_data_from_numpy(x)
            ref = c.infer("resnet50", [direct]).as_numpy("output")
            assert got.shape == ref.shape == (3, 1000)
            np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2)
            assert (got.argmax(axis=1) == ref.argmax(axis=1)).all()
            # classification extension on the ensemble output
            out = g.InferRequestedOutput("OUTPUT", class_count=2)
            top = c.infer("preprocess_resnet50_ensemble", [inp], outputs=[out]).as_numpy("OUTPUT")
            assert top.shape == (3, 2) and top[0, 0].decode().split(":")[1] == str(int(ref[0].argmax()))
            # a broken file is a clean request error
            bad = g.InferInput("INPUT", [1, 1], "BYTES").set_data_from_numpy(np.array([[b"not an image"]],
                                                                                      dtype=np.object_))
            with pytest.raises(Exception, match="unable to decode image"):
                c.infer("preprocess_resnet50_ensemble", [bad])

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


def test_graph_replay_equals_eager(model):
    import torch

    x = torch.randn(8, 3, 224, 224, device="cuda", dtype=torch.float16)
    out = torch.empty(8, 1000, device="cuda", dtype=torch.float32)
    model.run(x, out)
    model.run(x, out)  # replay
    model.stream.synchronize()
    with torch.no_grad(), torch.cuda.stream(model.stream):
        eager = model._forward_eager(x)
    model.stream.synchronize()
    # library kernels that split K accumulate with atomics, so two runs may differ by an fp16 ulp
    np.testing.assert_allclose(out.cpu().numpy(), eager.cpu().numpy(), rtol=1e-2, atol=1e-2)

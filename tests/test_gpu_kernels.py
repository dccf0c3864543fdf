"""Numerics of the hand-written HIP kernels against plain fp32 / Pillow /
numpy references."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    return torch


@pytest.mark.parametrize("src_hw,channels", [((375, 500), 3), ((224, 224), 3), ((100, 130), 3), ((600, 450), 1),
                                             ((300, 300), 4)])
@pytest.mark.parametrize("layout", ["NCHW", "NHWC"])
def test_preprocess_matches_pillow_bit_exact(torch, src_hw, channels, layout):
    from triton_client_amd.ops.preprocess import Preprocessor, reference_preprocess

    rng = np.random.default_rng(hash((src_hw, channels)) & 0xFFFF)
    shape = src_hw + ((channels,) if channels > 1 else ())
    imgs = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(3)]
    pre = Preprocessor(224, 224, 3, layout, "FP32", scaling="NONE", resize="pil", device=0)
    out_shape = (3, 3, 224, 224) if layout == "NCHW" else (3, 224, 224, 3)
    out = torch.zeros(out_shape, device="cuda:0", dtype=torch.float32)
    pre(imgs, out)
    got = out.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = reference_preprocess(im, 224, 224, 3, layout, np.float32, "NONE", "pil")
        np.testing.assert_array_equal(got[i], ref)
    pre.close()


@pytest.mark.parametrize("dtype,np_dtype,scaling", [("FP16", np.float16, "INCEPTION"), ("FP32", np.float32, "VGG"),
                                                    ("INT8", np.int8, "VGG"), ("UINT8", np.uint8, "NONE"),
                                                    ("BF16", None, "INCEPTION"),
                                                    ("FP32", np.float32, ((0.485, 0.456, 0.406),
                                                                          (0.229, 0.224, 0.225)))])
def test_preprocess_scaling_and_cast(torch, dtype, np_dtype, scaling):
    from triton_client_amd.ops.preprocess import Preprocessor, reference_preprocess

    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, size=(333, 517, 3), dtype=np.uint8) for _ in range(2)]
    pre = Preprocessor(299, 299, 3, "NCHW", dtype, scaling=scaling, resize="pil", device=0)
    tdt = {"FP16": torch.float16, "FP32": torch.float32, "INT8": torch.int8, "UINT8": torch.uint8,
           "BF16": torch.bfloat16}[dtype]
    out = torch.zeros((2, 3, 299, 299), device="cuda:0", dtype=tdt)
    pre(imgs, out)
    got = out.float().cpu().numpy()
    for i, im in enumerate(imgs):
        ref32 = reference_preprocess(im, 299, 299, 3, "NCHW", np.float32, scaling, "pil")
        if dtype == "BF16":
            ref = torch.from_numpy(ref32).to(torch.bfloat16).float().numpy()
            np.testing.assert_allclose(got[i], ref, rtol=0, atol=1e-2)
        elif np_dtype in (np.int8, np.uint8):
            ref = reference_preprocess(im, 299, 299, 3, "NCHW", np_dtype, scaling, "pil").astype(np.float32)
            assert np.abs(got[i] - ref).max() <= 1  # rint ties may differ by one step
        else:
            ref = ref32.astype(np_dtype).astype(np.float32)
            np.testing.assert_allclose(got[i], ref, rtol=1e-3, atol=1e-3)
    pre.close()


def test_preprocess_gray_bgr_cv_and_big_downscale(torch):
    from triton_client_amd.ops.preprocess import Preprocessor, reference_preprocess

    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    # RGB -> L output, BGR input order
    pre = Preprocessor(224, 224, 1, "NCHW", "FP32", scaling="VGG", resize="pil", device=0)
    out = torch.zeros((1, 1, 224, 224), device="cuda:0")
    pre([im], out, bgr=True)
    ref = reference_preprocess(im, 224, 224, 1, "NCHW", np.float32, "VGG", "pil", bgr=True)
    np.testing.assert_array_equal(out.cpu().numpy()[0], ref)
    pre.close()
    # OpenCV-style 2-tap tables
    pre = Preprocessor(224, 224, 3, "NHWC", "FP32", scaling="NONE", resize="cv", device=0)
    out = torch.zeros((1, 224, 224, 3), device="cuda:0")
    pre([im], out)
    ref = reference_preprocess(im, 224, 224, 3, "NHWC", np.float32, "NONE", "cv")
    np.testing.assert_array_equal(out.cpu().numpy()[0], ref)
    pre.close()
    # 4000x3000 -> 224: the tile height shrinks so the LDS tile still fits
    big = rng.integers(0, 256, size=(3000, 4000, 3), dtype=np.uint8)
    pre = Preprocessor(224, 224, 3, "NCHW", "FP32", scaling="NONE", resize="pil", device=0, max_image_bytes=64 << 20)
    out = torch.zeros((1, 3, 224, 224), device="cuda:0")
    pre([big], out)
    ref = reference_preprocess(big, 224, 224, 3, "NCHW", np.float32, "NONE", "pil")
    np.testing.assert_array_equal(out.cpu().numpy()[0], ref)
    # upscale
    small = rng.integers(0, 256, size=(60, 90, 3), dtype=np.uint8)
    pre([small], out)
    ref = reference_preprocess(small, 224, 224, 3, "NCHW", np.float32, "NONE", "pil")
    np.testing.assert_array_equal(out.cpu().numpy()[0], ref)
    pre.close()


def test_bytes_pack_index_gather(torch):
    import tritonclient.utils as utils
    from triton_client_amd.ops import codec

    rng = np.random.default_rng(7)
    items = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 200, size=3000)]
    items[5] = b""
    items[6] = b""
    items += [bytes(rng.integers(0, 256, size=100000, dtype=np.uint8))]  # one element larger than the LDS chunk
    wire = utils.serialize_byte_tensor(np.array(items, dtype=object)).item()
    payload = np.frombuffer(b"".join(items), dtype=np.uint8)
    offs = np.zeros(len(items) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b in items])
    d_payload = torch.from_numpy(payload.copy()).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    packed = codec.bytes_pack(d_payload, d_offs)
    assert bytes(packed.cpu().numpy()) == wire
    lens = torch.from_numpy(np.diff(offs).astype(np.int32)).cuda()
    packed2 = codec.bytes_pack_from_lengths(d_payload, lens)
    assert bytes(packed2.cpu().numpy()) == wire
    scan = codec.lengths_to_offsets(lens, add_each=4).cpu().numpy()
    np.testing.assert_array_equal(scan, np.concatenate([[0], np.cumsum(np.diff(offs) + 4)]))
    # index (unpack) on the GPU
    o, l = codec.bytes_index(packed, len(items))
    o, l = o.cpu().numpy(), l.cpu().numpy()
    np.testing.assert_array_equal(l, np.diff(offs))
    np.testing.assert_array_equal(o, offs[:-1] + 4 * (np.arange(len(items)) + 1))
    # fewer elements than the stream holds
    o2, l2 = codec.bytes_index(packed, 10)
    assert len(o2) == 10
    # gather into fixed pitch rows
    rows = codec.bytes_gather(packed, torch.from_numpy(o[:50]).cuda(), torch.from_numpy(l[:50]).cuda(), 64)
    rows = rows.cpu().numpy()
    for i in range(50):
        exp = items[i][:64].ljust(64, b"\0")
        assert bytes(rows[i]) == exp
    # malformed stream
    bad = torch.from_numpy(np.frombuffer(struct.pack("<I", 100) + b"abc", dtype=np.uint8).copy()).cuda()
    with pytest.raises(ValueError):
        codec.bytes_index(bad, 5)


def _host_index(wire, max_elems):
    offs, lens, pos = [], [], 0
    while pos < len(wire) and len(offs) < max_elems:
        if pos + 4 > len(wire):
            raise ValueError("truncated prefix")
        n = struct.unpack_from("<I", wire, pos)[0]
        if n > len(wire) - pos - 4:
            raise ValueError("truncated payload")
        offs.append(pos + 4)
        lens.append(n)
        pos += 4 + n
    return np.asarray(offs, dtype=np.int64), np.asarray(lens, dtype=np.int64)


@pytest.mark.parametrize("case", ["short", "empty", "mixed", "huge", "boundary", "unaligned"])
def test_bytes_index_chunk_parallel(torch, case):
    """The chunk-parallel index (pointer jumping per 4 KiB chunk, one lookup per chunk,
    per-chunk emit) against a host walk and the single-workgroup kernel."""
    from triton_client_amd.ops import codec

    rng = np.random.default_rng(11)
    if case == "short":
        sizes = rng.integers(1, 12, size=40000)
    elif case == "empty":
        sizes = np.zeros(5000, dtype=np.int64)  # 1024 prefixes per chunk: the longest path pointer jumping sees
    elif case == "mixed":
        sizes = rng.integers(0, 3000, size=800)
    elif case == "huge":
        sizes = np.array([3, 70000, 0, 5, 300000, 17, 1])  # elements that skip whole chunks
    elif case == "boundary":
        sizes = np.array([4092, 4092, 0, 4088, 1] + [60] * 64)  # prefixes exactly at / straddling chunk edges
    else:
        sizes = rng.integers(0, 50, size=20000)
    wire = b"".join(struct.pack("<I", int(n)) + bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in sizes)
    lead = 3 if case == "unaligned" else 0
    dev = torch.from_numpy(np.frombuffer(b"x" * lead + wire, dtype=np.uint8).copy()).cuda()[lead:]
    exp_o, exp_l = _host_index(wire, len(sizes))
    o, l = codec.bytes_index(dev, len(sizes), parallel=True)
    np.testing.assert_array_equal(o.cpu().numpy(), exp_o)
    np.testing.assert_array_equal(l.cpu().numpy(), exp_l)
    o1, l1 = codec.bytes_index(dev, len(sizes), parallel=False)
    assert torch.equal(o, o1) and torch.equal(l, l1)
    # fewer elements than the stream holds, more than it holds
    k = max(1, len(sizes) // 3)
    o2, l2 = codec.bytes_index(dev, k, parallel=True)
    np.testing.assert_array_equal(o2.cpu().numpy(), exp_o[:k])
    np.testing.assert_array_equal(l2.cpu().numpy(), exp_l[:k])
    o3, _ = codec.bytes_index(dev, len(sizes) + 100, parallel=True)
    assert len(o3) == len(sizes)
    # malformed: last payload cut short, then a prefix cut short; a cut beyond the requested elements is ignored
    for cut in (1, int(sizes[-1]) + 2):
        with pytest.raises(ValueError):
            codec.bytes_index(dev[: len(wire) - cut], len(sizes), parallel=True)
    o4, _ = codec.bytes_index(dev[: len(wire) - 1], len(sizes) - 1, parallel=True)
    np.testing.assert_array_equal(o4.cpu().numpy(), exp_o[:-1])


def test_bf16_kernels(torch):
    import tritonclient.utils as utils
    from triton_client_amd.ops import codec

    x = torch.randn(1 << 20, device="cuda") * 100
    x[:4] = torch.tensor([0.0, -0.0, float("inf"), 1.00390625 + 2 ** -9])
    trunc = codec.fp32_to_bf16(x).cpu().numpy().view(np.uint16)
    ref = np.frombuffer(utils.serialize_bf16_tensor(x.cpu().numpy()).item(), dtype=np.uint16)
    np.testing.assert_array_equal(trunc, ref)
    rne = codec.fp32_to_bf16(x, round_nearest=True)
    np.testing.assert_array_equal(rne.cpu().numpy().view(np.uint16),
                                  x.to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16))
    back = codec.bf16_to_fp32(rne)
    assert torch.equal(back, x.to(torch.bfloat16).float())
    odd = torch.randn(1027, device="cuda")
    assert torch.equal(codec.bf16_to_fp32(codec.fp32_to_bf16(odd, True)), odd.to(torch.bfloat16).float())


@pytest.mark.parametrize("dt", ["float32", "float16", "bfloat16"])
def test_topk_and_classification_strings(torch, dt):
    from triton_client_amd.ops import classification

    tdt = getattr(torch, dt)
    logits = (torch.randn(16, 1000, device="cuda") * 5).to(tdt)
    logits[3, 10] = logits[3, 20] = 50.0  # tie -> lowest index first
    vals, idx = classification.topk(logits, 5)
    ref = logits.float().cpu().numpy()
    order = np.argsort(-ref, axis=1, kind="stable")[:, :5]
    np.testing.assert_array_equal(idx.cpu().numpy(), order)
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(ref, order, axis=1))
    labels = [f"label {i}" for i in range(1000)]
    table = classification.LabelTable(labels, logits.device)
    buf, n = classification.classification_bytes(vals, idx, table)
    import tritonclient.utils as utils

    got = utils.deserialize_bytes_tensor(bytes(buf[:n].cpu().numpy()))
    exp = classification.topk_strings(logits, 5, labels).reshape(-1)
    assert len(got) == 80
    for g, e in zip(got, exp):
        gs, gi, gl = g.decode().split(":")
        es, ei, el = e.decode().split(":")
        assert (gi, gl) == (ei, el) and abs(float(gs) - float(es)) <= 1.1e-6, (g, e)


@pytest.mark.parametrize("rows,k,with_labels", [(128, 5, True), (128, 10, False), (3, 1000, True)])
def test_classification_payload_many_entries(torch, rows, k, with_labels):
    """More entries than one formatting pass holds (chunks of 512): element
    order, offsets and the running size must match a serial host encoder."""
    import tritonclient.utils as utils
    from triton_client_amd.ops import classification

    logits = torch.randn(rows, 1000, device="cuda") * 8
    labels = [f"class-{i}" * (1 + i % 3) for i in range(1000)] if with_labels else None
    table = classification.LabelTable(labels, logits.device) if with_labels else None
    blob, shape = classification.classification_payload(logits, k, table)
    assert shape == [rows, k]
    got = utils.deserialize_bytes_tensor(blob)
    exp = classification.topk_strings(logits, k, labels).reshape(-1)
    assert len(got) == rows * k
    for g, e in zip(got, exp):
        gp, ep = g.decode().split(":"), e.decode().split(":")
        assert gp[1:] == ep[1:] and abs(float(gp[0]) - float(ep[0])) <= 1.1e-6, (g, e)
    # destination too small: size is still reported, nothing is written past the end
    vals, idx = classification.topk(logits, k)
    small = torch.zeros(100, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        classification.classification_bytes(vals, idx, table, out=small)


def _nhwc(t):
    import torch as _torch

    return t.contiguous(memory_format=_torch.channels_last)


@pytest.mark.parametrize("variant", ["auto", "128x128", "128x64", "64x64", "128x128w8", "256x128w8", "128x64w8",
                                     "64x64w8", "128x256w8", "256x128w16", "256x256w16", "128x256w16"])
@pytest.mark.parametrize("shape", [
    # batch, H, W, Cin, Cout, stride, residual, relu
    (2, 56, 56, 64, 256, 1, True, True),
    (3, 28, 28, 512, 128, 1, False, True),
    (2, 56, 56, 256, 512, 2, False, False),
    (5, 7, 7, 2048, 512, 1, False, True),   # M = 245: ragged last tile
    (1, 14, 14, 1024, 2048, 2, True, True),
])
def test_conv1x1_mfma_matches_fp32_reference(torch, shape, variant):
    from triton_client_amd.ops import conv

    b, h, w, cin, cout, stride, with_res, relu = shape
    if "x256w" in variant and cout % 256:
        pytest.skip("tile wider than the layer")
    gen = torch.Generator(device="cuda").manual_seed(cin + cout + h)
    x = _nhwc(torch.randn(b, cin, h, w, device="cuda", generator=gen).half())
    wt = (torch.randn(cout, cin, device="cuda", generator=gen) / cin ** 0.5).half()
    bias = torch.randn(cout, device="cuda", generator=gen).half()
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = _nhwc(torch.randn(b, cout, oh, ow, device="cuda", generator=gen).half()) if with_res else None
    got = conv.conv1x1(x, wt, bias, res, relu=relu, stride=stride, variant=variant)
    ref = torch.nn.functional.conv2d(x.float(), wt.float().reshape(cout, cin, 1, 1), bias.float(), stride=stride)
    if with_res:
        ref = ref + res.float()
    if relu:
        ref = ref.clamp_min(0)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(got.float(), ref, rtol=4e-3, atol=4e-3)


def test_conv1x1_exact_integer_operand_maps(torch):
    """Small integers make every product and sum exact in fp16/fp32, so a wrong
    MFMA lane map (row/column swap, k permutation) cannot hide in rounding; the
    weight matrix is asymmetric on purpose."""
    from triton_client_amd.ops import conv

    b, h, w, cin, cout = 1, 12, 12, 64, 128
    x = _nhwc(torch.randint(-3, 4, (b, cin, h, w), device="cuda").half())
    wt = torch.randint(-2, 3, (cout, cin), device="cuda").half()
    wt[5, 7] = 9.0
    bias = torch.arange(cout, device="cuda").half()
    for staging in ("lds2", "lds3", "reg"):
        for variant in ("128x128", "128x64", "64x64", "128x128w8", "256x128w8", "128x64w8", "64x64w8", "256x128w16"):
            if staging == "reg" and variant.endswith("w16"):
                continue  # the 16-wavefront tiles are DMA-staged only
            got = conv.conv1x1(x, wt, bias, None, relu=False, variant=variant, staging=staging)
            ref = torch.nn.functional.conv2d(x.float(), wt.float().reshape(cout, cin, 1, 1), bias.float())
            assert torch.equal(got.float(), ref), (variant, staging)


@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False)])
def test_bias_act_inplace(torch, with_res, relu):
    from triton_client_amd.ops import conv

    y = _nhwc(torch.randn(4, 192, 17, 13, device="cuda").half())
    bias = torch.randn(192, device="cuda").half()
    res = _nhwc(torch.randn(4, 192, 17, 13, device="cuda").half()) if with_res else None
    ref = y.float() + bias.float().reshape(1, -1, 1, 1)
    if with_res:
        ref = ref + res.float()
    if relu:
        ref = ref.clamp_min(0)
    out = conv.bias_act_(y, bias, res, relu=relu)
    assert out is y
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("variant", ["auto", "128x64", "64x64"])
@pytest.mark.parametrize("shape", [
    # batch, H, W, Cin, Cout, stride, relu
    (2, 56, 56, 64, 64, 1, True),
    (3, 56, 56, 128, 128, 2, True),
    (2, 14, 14, 256, 256, 1, False),
    (5, 7, 7, 512, 512, 1, True),      # M = 245, every border pixel pads
    (1, 15, 13, 64, 128, 2, True),     # odd sizes
])
def test_conv3x3_implicit_gemm_matches_fp32_reference(torch, shape, variant):
    from triton_client_amd.ops import conv

    b, h, w, cin, cout, stride, relu = shape
    gen = torch.Generator(device="cuda").manual_seed(cin * 3 + cout + h)
    x = _nhwc(torch.randn(b, cin, h, w, device="cuda", generator=gen).half())
    wt = _nhwc((torch.randn(cout, cin, 3, 3, device="cuda", generator=gen) / (9 * cin) ** 0.5).half())
    bias = torch.randn(cout, device="cuda", generator=gen).half()
    got = conv.conv2d(x, wt, bias, None, relu=relu, stride=stride, padding=1, variant=variant)
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=stride, padding=1)
    if relu:
        ref = ref.clamp_min(0)
    assert got.shape == ref.shape
    torch.testing.assert_close(got.float(), ref, rtol=4e-3, atol=4e-3)


def test_conv3x3_exact_integer_taps(torch):
    """Integer data: a wrong tap order or padding rule changes exact sums."""
    from triton_client_amd.ops import conv

    x = _nhwc(torch.randint(-2, 3, (2, 64, 9, 11), device="cuda").half())
    wt = _nhwc(torch.randint(-1, 2, (64, 64, 3, 3), device="cuda").half())
    wt[3, 5, 0, 2] = 4.0  # asymmetric in (kh, kw)
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), None, stride=1, padding=1)
    for staging in ("lds2", "lds3", "reg"):
        for variant in ("128x64", "64x64", "128x64w8", "64x64w8"):
            got = conv.conv2d(x, wt, None, None, relu=False, stride=1, padding=1, variant=variant, staging=staging)
            assert torch.equal(got.float(), ref), (variant, staging)
    # stride 2, two K steps per tap, 128 output channels
    x = _nhwc(torch.randint(-2, 3, (3, 128, 10, 7), device="cuda").half())
    wt = _nhwc(torch.randint(-1, 2, (128, 128, 3, 3), device="cuda").half())
    wt[7, 70, 2, 0] = 3.0
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), None, stride=2, padding=1)
    for staging in ("lds2", "lds3", "reg"):
        for variant in ("128x128", "128x64", "64x64", "128x128w8", "256x128w8", "128x64w8", "64x64w8", "256x128w16"):
            if staging == "reg" and variant.endswith("w16"):
                continue
            got = conv.conv2d(x, wt, None, None, relu=False, stride=2, padding=1, variant=variant, staging=staging)
            assert torch.equal(got.float(), ref), (variant, staging)


def test_conv3x3_sixteen_wave_tiles_exact(torch):
    """The 16-wavefront tiles (four wave rows) on integer data: 3x3, ragged last tile, 256 output channels."""
    from triton_client_amd.ops import conv

    x = _nhwc(torch.randint(-2, 3, (3, 128, 14, 14), device="cuda").half())
    wt = _nhwc(torch.randint(-1, 2, (256, 128, 3, 3), device="cuda").half())
    wt[200, 70, 2, 0] = 3.0
    bias = torch.arange(256, device="cuda").half()
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=1, padding=1)
    for staging in ("lds2", "lds3"):
        for variant in ("256x128w16", "256x256w16", "128x256w16"):
            got = conv.conv2d(x, wt, bias, None, relu=False, stride=1, padding=1, variant=variant, staging=staging)
            assert torch.equal(got.float(), ref), (variant, staging)


@pytest.mark.parametrize("shape", [(3, 128, 14, 14, 256), (2, 256, 14, 14, 256), (1, 64, 28, 28, 128),
                                   (5, 64, 7, 7, 256)])
def test_conv3x3_plain_halo_form_exact(torch, shape):
    """Halo form of the plain 3x3 convolution (input rows in LDS once, the K loop streams weights): integer data,
    image borders inside a tile (several images per tile), ragged last tile, one to four 64-channel slices."""
    from triton_client_amd.ops import conv

    b, cin, h, w, cout = shape
    gen = torch.Generator(device="cuda").manual_seed(b * 1000 + cin + cout)
    x = _nhwc(torch.randint(-2, 3, (b, cin, h, w), device="cuda", generator=gen).half())
    wt = _nhwc(torch.randint(-1, 2, (cout, cin, 3, 3), device="cuda", generator=gen).half())
    wt[cout - 3, cin - 5, 2, 0] = 3.0
    bias = torch.arange(cout, device="cuda").half()
    res = _nhwc(torch.randint(-4, 5, (b, cout, h, w), device="cuda", generator=gen).half())
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=1, padding=1)
    ref = (ref + res.float()).clamp_min(0)
    for variant in ("128x256halo", "128x128halo"):
        if variant == "128x256halo" and cout % 256:
            continue
        got = conv.conv2d(x, wt, bias, res, relu=True, stride=1, padding=1, variant=variant)
        assert torch.equal(got.float(), ref), variant
    with pytest.raises(RuntimeError):  # stride 2 has no halo form
        conv.conv2d(x, wt, bias, None, relu=True, stride=2, padding=1, variant="128x128halo")


@pytest.mark.parametrize("staging", ["lds2", "lds3", "reg"])
def test_conv_split_k_exact_and_repeatable(torch, staging):
    """Split-K (several blocks per output tile, partials summed by the block that arrives
    last): integer data must give the exact sums for every split count, tile variant and
    staging path; launching again must work (the arrival counters are back at zero); the
    result must not depend on which block arrives last."""
    from triton_client_amd.ops import conv

    # 3x3, stride 2, two K steps per tap (18 K steps): runs that start in the middle of a tap
    x = _nhwc(torch.randint(-2, 3, (3, 128, 10, 7), device="cuda").half())
    wt = _nhwc(torch.randint(-1, 2, (128, 128, 3, 3), device="cuda").half())
    wt[7, 70, 2, 0] = 3.0
    bias = torch.randint(-3, 4, (128,), device="cuda").half()
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=2, padding=1)
    res = _nhwc(torch.randint(-2, 3, tuple(ref.shape), device="cuda").half())
    ref = (ref + res.float()).clamp_min(0)
    for variant in ("64x64", "64x64w8", "128x64", "128x128w8"):
        for splits in (0, 2, 3, 5, 18, 64):  # 64 > K steps: clamped
            for _ in range(2):
                got = conv.conv2d(x, wt, bias, res, relu=True, stride=2, padding=1, variant=variant, staging=staging,
                                  split_k=splits)
                assert torch.equal(got.float(), ref), (variant, splits)
    # 1x1 with a long K loop and few tiles (the shape split-K is for), fp16 data: bit-identical across repeats
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = _nhwc(torch.randn(8, 2048, 7, 7, device="cuda", generator=gen).half())
    w1 = (torch.randn(512, 2048, device="cuda", generator=gen) / 2048 ** 0.5).half()
    b1 = torch.randn(512, device="cuda", generator=gen).half()
    want = torch.nn.functional.conv2d(x.float(), w1.float().reshape(512, 2048, 1, 1), b1.float()).clamp_min(0)
    first = conv.conv1x1(x, w1, b1, None, relu=True, staging=staging, split_k=4)
    torch.testing.assert_close(first.float(), want, rtol=4e-3, atol=4e-3)
    for _ in range(5):
        assert torch.equal(conv.conv1x1(x, w1, b1, None, relu=True, staging=staging, split_k=4), first)
    auto = conv.conv1x1(x, w1, b1, None, relu=True, staging=staging)
    torch.testing.assert_close(auto.float(), want, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("shape", [
    # batch, H, W, Cin, Mid, Cout, stride, residual
    (2, 56, 56, 64, 64, 256, 1, True),
    (3, 28, 28, 128, 128, 512, 1, True),
    (2, 56, 56, 128, 128, 512, 2, False),
    (1, 15, 13, 64, 64, 128, 2, True),      # odd sizes, M = 56: one partial tile
    (5, 7, 9, 192, 128, 64, 1, False),      # Cin of three K steps per tap, a single output chunk
])
def test_conv3x3_pointwise_fused_matches_reference(torch, shape):
    """3x3 convolution + bias + ReLU + 1x1 convolution + bias + residual + ReLU in one kernel
    against the two-step fp32 computation (middle activation rounded to fp16, as the kernel
    keeps it), and against the two separate kernels."""
    from triton_client_amd.ops import conv

    b, h, w, cin, mid, cout, stride, with_res = shape
    gen = torch.Generator(device="cuda").manual_seed(cin + mid * 3 + cout * 7 + h)
    x = _nhwc(torch.randn(b, cin, h, w, device="cuda", generator=gen).half())
    w1 = _nhwc((torch.randn(mid, cin, 3, 3, device="cuda", generator=gen) / (9 * cin) ** 0.5).half())
    b1 = torch.randn(mid, device="cuda", generator=gen).half()
    w2 = (torch.randn(cout, mid, device="cuda", generator=gen) / mid ** 0.5).half()
    b2 = torch.randn(cout, device="cuda", generator=gen).half()
    t = torch.nn.functional.conv2d(x.float(), w1.float(), b1.float(), stride=stride, padding=1).clamp_min(0)
    ref = torch.nn.functional.conv2d(t.half().float(), w2.float().reshape(cout, mid, 1, 1), b2.float())
    res = _nhwc(torch.randn(tuple(ref.shape), device="cuda", generator=gen).half()) if with_res else None
    if res is not None:
        ref = ref + res.float()
    ref = ref.clamp_min(0)
    got = conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1)
    assert got.shape == ref.shape
    torch.testing.assert_close(got.float(), ref, rtol=4e-3, atol=4e-3)
    two = conv.conv1x1(conv.conv2d(x, w1, b1, None, relu=True, stride=stride, padding=1), w2, b2, res, relu=True)
    torch.testing.assert_close(got.float(), two.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("mid,n3", [(64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("shape", [(2, 56, 56, 256, 1), (1, 15, 13, 128, 2), (3, 9, 11, 64, 1), (44, 28, 28, 512, 1)])
def test_conv3x3_pointwise_next_conv_fused(torch, shape, mid, n3):
    """The fused block kernel with the next block's first 1x1 convolution attached: both
    results against the separate kernels and an fp32 reference."""
    from triton_client_amd.ops import conv

    b, h, w, cout, stride = shape
    gen = torch.Generator(device="cuda").manual_seed(cout + n3 + h + mid)
    x = _nhwc(torch.randn(b, 64, h, w, device="cuda", generator=gen).half())
    w1 = _nhwc((torch.randn(mid, 64, 3, 3, device="cuda", generator=gen) / 24.0).half())
    b1 = torch.randn(mid, device="cuda", generator=gen).half()
    w2 = (torch.randn(cout, mid, device="cuda", generator=gen) / mid ** 0.5).half()
    b2 = torch.randn(cout, device="cuda", generator=gen).half()
    w3 = (torch.randn(n3, cout, device="cuda", generator=gen) / cout ** 0.5).half()
    b3 = torch.randn(n3, device="cuda", generator=gen).half()
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = _nhwc(torch.randn(b, cout, oh, ow, device="cuda", generator=gen).half())
    y_ref = conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1)
    t_ref = torch.nn.functional.conv2d(y_ref.float(), w3.float().reshape(n3, cout, 1, 1), b3.float()).clamp_min(0)
    y, t = conv.conv2d_pointwise(x, w1, b1, w2, b2, res, relu=True, stride=stride, padding=1, next_weight=w3,
                                 next_bias=b3)
    assert torch.equal(y, y_ref)
    assert t.shape == t_ref.shape
    torch.testing.assert_close(t.float(), t_ref, rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(t.float(), conv.conv1x1(y_ref, w3, b3, None, relu=True).float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("attach", [False, True])
@pytest.mark.parametrize("shape", [(2, 56, 56, 256), (1, 15, 13, 128), (44, 28, 28, 256)])
def test_conv3x3_pointwise_projection_in_kernel(torch, shape, attach):
    """The projection shortcut as extra K of the second GEMM (block input as a second operand
    slab, weights side by side) against the separate computation in fp32."""
    from triton_client_amd.ops import conv

    b, h, w, cout = shape
    gen = torch.Generator(device="cuda").manual_seed(cout + h + int(attach))
    x0 = _nhwc(torch.randn(b, 64, h, w, device="cuda", generator=gen).half())  # block input
    t = _nhwc(torch.randn(b, 64, h, w, device="cuda", generator=gen).half())   # first 1x1's output
    w1 = _nhwc((torch.randn(64, 64, 3, 3, device="cuda", generator=gen) / 24.0).half())
    b1 = torch.randn(64, device="cuda", generator=gen).half()
    w2 = (torch.randn(cout, 64, device="cuda", generator=gen) / 8.0).half()
    b2 = torch.randn(cout, device="cuda", generator=gen).half()
    wd = (torch.randn(cout, 64, device="cuda", generator=gen) / 8.0).half()
    bd = torch.randn(cout, device="cuda", generator=gen).half()
    w3 = (torch.randn(64, cout, device="cuda", generator=gen) / cout ** 0.5).half()
    b3 = torch.randn(64, device="cuda", generator=gen).half()
    mid = torch.nn.functional.conv2d(t.float(), w1.float(), b1.float(), padding=1).clamp_min(0).half().float()
    ref = (torch.nn.functional.conv2d(mid, w2.float().reshape(cout, 64, 1, 1), b2.float())
           + torch.nn.functional.conv2d(x0.float(), wd.float().reshape(cout, 64, 1, 1), bd.float())).clamp_min(0)
    wcat = torch.cat([w2, wd], dim=1).contiguous()
    bcat = (b2.float() + bd.float()).half()
    got = conv.conv2d_pointwise(t, w1, b1, wcat, bcat, None, relu=True, padding=1, proj_input=x0,
                                next_weight=w3 if attach else None, next_bias=b3 if attach else None)
    if attach:
        got, nxt = got
        nref = torch.nn.functional.conv2d(got.float(), w3.float().reshape(64, cout, 1, 1), b3.float()).clamp_min(0)
        torch.testing.assert_close(nxt.float(), nref, rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(got.float(), ref, rtol=5e-3, atol=5e-3)


def test_conv3x3_pointwise_fused_exact_integers(torch):
    """Small integers keep every intermediate exact in fp16: a wrong slab, swizzle or chunk
    order in the second GEMM changes sums."""
    from triton_client_amd.ops import conv

    for mid, cout in ((64, 192), (128, 256)):
        x = _nhwc(torch.randint(-1, 2, (2, 64, 9, 11), device="cuda").half())
        w1 = torch.zeros(mid, 64, 3, 3, device="cuda")
        w1[torch.rand(mid, 64, 3, 3, device="cuda") < 0.05] = 1.0
        w1[3, 5, 0, 2] = -2.0
        w1 = _nhwc(w1.half())
        b1 = torch.randint(-2, 3, (mid,), device="cuda").half()
        w2 = torch.zeros(cout, mid, device="cuda")
        w2[torch.rand(cout, mid, device="cuda") < 0.1] = 1.0
        w2[7, mid - 1] = -3.0
        w2 = w2.half()
        b2 = torch.randint(-2, 3, (cout,), device="cuda").half()
        t = torch.nn.functional.conv2d(x.float(), w1.float(), b1.float(), padding=1).clamp_min(0)
        ref = torch.nn.functional.conv2d(t, w2.float().reshape(cout, mid, 1, 1), b2.float())
        assert t.abs().max() < 2048 and ref.abs().max() < 2048  # exactly representable
        got = conv.conv2d_pointwise(x, w1, b1, w2, b2, None, relu=False, padding=1)
        assert torch.equal(got.float(), ref), (mid, cout)


@pytest.mark.parametrize("batch,n_out", [(1, 1000), (8, 1000), (37, 1000), (128, 1000), (5, 16), (16, 7)])
def test_avgpool_linear_tail(torch, batch, n_out):
    """Global average pooling + fully connected layer + bias with fp32 output against fp32
    PyTorch (batches and class counts that are not multiples of the 16x16 MFMA tile included)."""
    from triton_client_amd.ops import conv

    gen = torch.Generator(device="cuda").manual_seed(batch * 31 + n_out)
    y = _nhwc(torch.randn(batch, 2048, 7, 7, device="cuda", generator=gen).half())
    w = (torch.randn(n_out, 2048, device="cuda", generator=gen) / 45.0).half()
    b = torch.randn(n_out, device="cuda", generator=gen).half()
    feats = y.float().mean(dim=(2, 3)).half().float()  # the kernel keeps the pooled features in fp16
    ref = feats @ w.float().t() + b.float()
    got = conv.avgpool_linear(y, w, b)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    out = torch.full((batch, n_out), float("nan"), device="cuda")
    assert conv.avgpool_linear(y, w, None, out) is out
    torch.testing.assert_close(out, ref - b.float(), rtol=2e-3, atol=2e-3)


def test_batched_copy_many_entries_and_alignments(torch):
    from triton_client_amd.ops.copy import batched_copy

    gen = torch.Generator(device="cuda").manual_seed(5)
    src = torch.randint(0, 255, (200, 70001), device="cuda", dtype=torch.uint8, generator=gen)
    dst = torch.zeros_like(src)
    sizes = [70001 if i % 3 else 4096 + i for i in range(150)]  # more than one launch (64 entries each)
    triples = []
    for i, n in enumerate(sizes):
        shift = i % 5  # odd source/destination offsets: the byte path
        triples.append((src[i].data_ptr() + shift, dst[i].data_ptr() + shift, n - shift))
    batched_copy(triples)
    torch.cuda.synchronize()
    for i, n in enumerate(sizes):
        shift = i % 5
        assert torch.equal(dst[i, shift:n], src[i, shift:n]), i
        assert int(dst[i, n:].sum()) == 0 and int(dst[i, :shift].sum()) == 0
    assert int(dst[150:].sum()) == 0
    batched_copy([])  # no-op
    # request-sized, 16-byte aligned ranges (whole 16 KiB chunks plus a ragged tail): the gather of a merged batch
    big_src = torch.randint(0, 255, (16, 1204224 + 48), device="cuda", dtype=torch.uint8, generator=gen)
    big_dst = torch.zeros(16 * (1204224 + 48) + 64, device="cuda", dtype=torch.uint8)
    base = (-big_dst.data_ptr()) % 16
    batched_copy([(big_src[i].data_ptr(), big_dst.data_ptr() + base + i * (1204224 + 48), 1204224 + 40)
                  for i in range(16)])
    torch.cuda.synchronize()
    view = big_dst[base: base + 16 * (1204224 + 48)].view(16, -1)
    assert torch.equal(view[:, :1204224 + 40], big_src[:, :1204224 + 40]) and int(view[:, 1204224 + 40:].sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 112, 112), (3, 16, 9, 7), (1, 8, 1, 1), (2, 24, 10, 13)])
def test_bias_relu_maxpool_matches_reference(torch, shape):
    from triton_client_amd.ops import conv

    gen = torch.Generator(device="cuda").manual_seed(5)
    x = _nhwc(torch.randn(*shape, device="cuda", generator=gen).half())
    bias = torch.randn(shape[1], device="cuda", generator=gen).half()
    got = conv.bias_relu_maxpool(x, bias)
    # the kernel adds the bias to the pooled maximum in fp32 and rounds once
    pooled = torch.nn.functional.max_pool2d(x.float(), 3, stride=2, padding=1)
    ref = (pooled + bias.float().view(1, -1, 1, 1)).clamp_min(0).half()
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 64, 96), (1, 37, 54), (2, 7, 10), (40, 224, 224)])
def test_stem_conv_pool_matches_fp32_reference(torch, shape):
    """conv 7x7/2 + bias + ReLU + max pool 3x3/2 in one kernel vs the fp32 chain,
    including image sizes that leave ragged pooled tiles."""
    from triton_client_amd.ops import conv

    n, h, w = shape
    gen = torch.Generator(device="cuda").manual_seed(h * w)
    x = torch.randn(n, 3, h, w, device="cuda", generator=gen).half()
    wt = (torch.randn(64, 3, 7, 7, device="cuda", generator=gen) / 147 ** 0.5).half()
    bias = torch.randn(64, device="cuda", generator=gen).half()
    got = conv.stem_conv_pool(x, conv.pack_stem_weight(wt), bias)
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=2, padding=3).clamp_min(0)
    ref = torch.nn.functional.max_pool2d(ref, 3, stride=2, padding=1)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(got.float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.gpu
def test_stem_conv_pool_exact_integer_taps(torch):
    """Integer data: a wrong tap / channel order or padding rule changes exact sums."""
    from triton_client_amd.ops import conv

    x = torch.randint(-2, 3, (2, 3, 40, 46), device="cuda").half()
    wt = torch.randint(-1, 2, (64, 3, 7, 7), device="cuda").half()
    wt[5, 2, 6, 0] = 3.0  # asymmetric in (c, kh, kw)
    wt[9, 0, 0, 6] = -3.0
    bias = torch.arange(-32, 32, device="cuda").half()
    got = conv.stem_conv_pool(x, conv.pack_stem_weight(wt), bias)
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), bias.float(), stride=2, padding=3).clamp_min(0)
    ref = torch.nn.functional.max_pool2d(ref, 3, stride=2, padding=1)
    assert torch.equal(got.float(), ref)


def _rows_block_operands(torch, b, h, w, n3, proj, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    t = _nhwc(torch.randn(b, 64, h, w, device="cuda", generator=gen).half())
    w1 = _nhwc((torch.randn(64, 64, 3, 3, device="cuda", generator=gen) / 24.0).half())
    b1 = torch.randn(64, device="cuda", generator=gen).half()
    w2 = (torch.randn(256, 64, device="cuda", generator=gen) / 8.0).half()
    b2 = torch.randn(256, device="cuda", generator=gen).half()
    w3 = (torch.randn(n3, 256, device="cuda", generator=gen) / 16.0).half()
    b3 = torch.randn(n3, device="cuda", generator=gen).half()
    x0 = wd = res = None
    if proj:
        x0 = _nhwc(torch.randn(b, 64, h, w, device="cuda", generator=gen).half())
        wd = (torch.randn(256, 64, device="cuda", generator=gen) / 8.0).half()
    else:
        res = _nhwc(torch.randn(b, 256, h, w, device="cuda", generator=gen).half())
    return t, w1, b1, w2, b2, w3, b3, x0, wd, res


@pytest.mark.parametrize("n3,proj", [(64, False), (128, False), (64, True)])
@pytest.mark.parametrize("shape", [(2, 56, 56), (1, 15, 13), (3, 9, 11), (5, 28, 28), (16, 56, 56)])
def test_bottleneck_rows_matches_reference(torch, shape, n3, proj):
    """The row-organised bottleneck kernel (3x3 + expansion (+ projection | residual) + next 1x1, no
    barrier behind the 3x3 part) against fp32 PyTorch and against the chunked fused kernel."""
    from triton_client_amd.ops import conv

    b, h, w = shape
    t, w1, b1, w2, b2, w3, b3, x0, wd, res = _rows_block_operands(torch, b, h, w, n3, proj, 17 * n3 + h + int(proj))
    F = torch.nn.functional
    mid = F.conv2d(t.float(), w1.float(), b1.float(), padding=1).clamp_min(0).half().float()
    ref = F.conv2d(mid, w2.float().reshape(256, 64, 1, 1), b2.float())
    if proj:
        ref = ref + F.conv2d(x0.float(), wd.float().reshape(256, 64, 1, 1))
        wcat = torch.cat([w2, wd], dim=1).contiguous()
    else:
        ref = ref + res.float()
        wcat = w2
    ref = ref.clamp_min(0)
    y, nxt = conv.bottleneck_rows(t, w1, b1, conv.permute_mid_k(wcat), b2, res, w3, b3, proj_input=x0)
    torch.testing.assert_close(y.float(), ref, rtol=5e-3, atol=5e-3)
    nref = F.conv2d(y.float(), w3.float().reshape(n3, 256, 1, 1), b3.float()).clamp_min(0)
    torch.testing.assert_close(nxt.float(), nref, rtol=4e-3, atol=4e-3)
    y_old, nxt_old = conv.conv2d_pointwise(t, w1, b1, wcat, b2, res, relu=True, padding=1, next_weight=w3,
                                           next_bias=b3, proj_input=x0)
    torch.testing.assert_close(y.float(), y_old.float(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(nxt.float(), nxt_old.float(), rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("n3,proj", [(64, False), (128, False), (64, True)])
def test_bottleneck_rows_exact_integers(torch, n3, proj):
    """Small integers keep every intermediate exact in fp16: a wrong K permutation of the second
    product, a wrong weight slab, swizzle, tap or phase changes sums.  Weights are asymmetric and
    sparse, every output channel and pixel is checked, M is not a multiple of the tile."""
    from triton_client_amd.ops import conv

    F = torch.nn.functional
    gen = torch.Generator(device="cuda").manual_seed(5 + n3 + int(proj))

    def sparse(shape, p, value=1.0):
        out = torch.zeros(shape, device="cuda")
        out[torch.rand(shape, device="cuda", generator=gen) < p] = value
        return out

    b, h, w = 3, 13, 11  # M = 429: three full tiles and a partial one
    t = _nhwc(torch.randint(-1, 2, (b, 64, h, w), device="cuda", generator=gen).half())
    w1 = sparse((64, 64, 3, 3), 0.05)
    w1[3, 5, 0, 2] = -2.0
    w1 = _nhwc(w1.half())
    b1 = torch.randint(-2, 3, (64,), device="cuda", generator=gen).half()
    w2 = sparse((256, 64), 0.1)
    w2[7, 63] = -3.0
    w2[200, 17] = 2.0
    b2 = torch.randint(-2, 3, (256,), device="cuda", generator=gen).half()
    w3 = sparse((n3, 256), 0.03)
    w3[n3 - 1, 255] = -1.0
    b3 = torch.randint(-2, 3, (n3,), device="cuda", generator=gen).half()
    mid = F.conv2d(t.float(), w1.float(), b1.float(), padding=1).clamp_min(0)
    ref = F.conv2d(mid, w2.reshape(256, 64, 1, 1), b2.float())
    x0 = res = None
    wcat = w2.half()
    if proj:
        x0 = _nhwc(torch.randint(-1, 2, (b, 64, h, w), device="cuda", generator=gen).half())
        wd = sparse((256, 64), 0.1)
        wd[100, 1] = -2.0
        ref = ref + F.conv2d(x0.float(), wd.reshape(256, 64, 1, 1))
        wcat = torch.cat([w2, wd], dim=1).half().contiguous()
    else:
        res = _nhwc(torch.randint(-3, 4, (b, 256, h, w), device="cuda", generator=gen).half())
        ref = ref + res.float()
    ref = ref.clamp_min(0)
    nref = F.conv2d(ref, w3.reshape(n3, 256, 1, 1), b3.float()).clamp_min(0)
    assert mid.abs().max() < 2048 and ref.abs().max() < 2048 and nref.abs().max() < 2048  # exactly representable
    y, nxt = conv.bottleneck_rows(t, w1, b1, conv.permute_mid_k(wcat), b2, res, w3.half(), b3, proj_input=x0)
    assert torch.equal(y.float(), ref)
    assert torch.equal(nxt.float(), nref)

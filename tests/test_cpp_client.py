"""Builds the native client (``make -C csrc``) and runs its unit tests, the
server-backed suite, the timeout / leak tools and the C++ examples against the
in-repo server."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "bin")


@pytest.fixture(scope="module")
def built():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host C++ toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "csrc"), "-j8", f"BUILD={os.path.join(ROOT, 'build')}",
                           "lib", "tests", "perf", "host_examples"], stdout=subprocess.DEVNULL)
    return BIN


def _run(cmd, timeout=180):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{' '.join(cmd)}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    return p.stdout


def test_cc_unit(built):
    out = _run([os.path.join(built, "cc_unit_test")])
    assert "0 failed" in out


def test_cc_client_suite(built, server):
    out = _run([os.path.join(built, "cc_client_test"), "-u", server.http_url])
    assert "0 failed" in out and "AsyncInferMulti" in out


def test_grpc_unit_and_client_suite(built, server):
    """HPACK / HTTP-2 / protobuf unit cases plus the gRPC client suite against
    the in-repo server (a grpcio server: interoperability with the stock gRPC
    implementation)."""
    out = _run([os.path.join(built, "grpc_client_test"), "-u", server.grpc_url])
    assert "0 failed" in out and "skipped" not in out and "GrpcDecoupledRepeat" in out


def test_cc_timeout_and_leak_tools(built, server):
    assert "timeouts reported" in _run([os.path.join(built, "client_timeout_test"), "-u", server.http_url, "-t",
                                        "100000"])
    assert "requests completed" in _run([os.path.join(built, "client_timeout_test"), "-u", server.http_url, "-t",
                                         "3000000"])
    assert "PASS" in _run([os.path.join(built, "memory_leak_test"), "-u", server.http_url, "-r", "200"])
    assert "PASS" in _run([os.path.join(built, "memory_leak_test"), "-u", server.http_url, "-r", "200", "-R"])


def test_cc_timeout_and_leak_tools_grpc(built, server):
    """The same tools over gRPC (``-i grpc``): sync, async and stream deadlines, every control call
    under a deadline (``-p``), and the growth check with a channel per client and a shared one."""
    tool = os.path.join(built, "client_timeout_test")
    assert "timeouts reported" in _run([tool, "-i", "grpc", "-u", server.grpc_url, "-t", "100000"])
    assert "requests completed" in _run([tool, "-i", "grpc", "-u", server.grpc_url, "-t", "3000000"])
    out = _run([tool, "-i", "grpc", "-u", server.grpc_url, "-p", "-t", "1"])
    assert "20 control calls" in out and "0 unexpected outcomes" in out and "PASS" in out
    assert "PASS" in _run([os.path.join(built, "memory_leak_test"), "-i", "grpc", "-u", server.grpc_url, "-r", "200"])
    assert "PASS" in _run([os.path.join(built, "memory_leak_test"), "-i", "grpc", "-u", server.grpc_url, "-r", "200",
                           "-R"])


def test_native_loadgen(built, server):
    import json

    out = _run([os.path.join(built, "tca_loadgen"), "-u", server.http_url, "-m", "simple", "-c", "8", "-d", "1.0",
                "--warmup", "0.2", "--shape", "INPUT0:INT32:1,16", "--shape", "INPUT1:INT32:1,16"])
    stats = json.loads(out.strip().splitlines()[-1])
    assert stats["errors"] == 0 and stats["requests"] > 100 and stats["p99_us"] >= stats["p50_us"] > 0


CPP_EXAMPLES = [
    "simple_http_infer_client", "simple_http_async_infer_client", "simple_http_string_infer_client",
    "simple_http_shm_client", "simple_http_sequence_sync_infer_client", "simple_http_health_metadata",
    "simple_http_model_control", "reuse_infer_objects_client", "image_client", "ensemble_image_client",
]


GRPC_CPP_EXAMPLES = [
    "simple_grpc_infer_client", "simple_grpc_async_infer_client", "simple_grpc_string_infer_client",
    "simple_grpc_shm_client", "simple_grpc_health_metadata", "simple_grpc_model_control",
    "simple_grpc_sequence_sync_infer_client", "simple_grpc_sequence_stream_infer_client",
    "simple_grpc_custom_repeat", "simple_grpc_keepalive_client", "simple_grpc_custom_args_client",
]


@pytest.mark.parametrize("name", GRPC_CPP_EXAMPLES)
def test_cpp_grpc_examples(built, server, name):
    out = _run([os.path.join(built, name), "-u", server.grpc_url])
    assert "PASS" in out


@pytest.mark.parametrize("name,extra", [("simple_grpc_infer_client", ["-C", "gzip"]),
                                        ("simple_grpc_sequence_sync_infer_client", ["-d"]),
                                        ("simple_grpc_sequence_stream_infer_client", ["-d"])])
def test_cpp_grpc_example_variants(built, server, name, extra):
    assert "PASS" in _run([os.path.join(built, name), "-u", server.grpc_url, *extra])


def test_native_loadgen_grpc_fixed_count(built, server, tmp_path):
    import csv
    import json

    trace = tmp_path / "trace.csv"
    out = _run([os.path.join(built, "tca_loadgen"), "-u", server.grpc_url, "-i", "grpc", "-m", "simple", "-c", "8",
                "--count", "500", "--warmup-count", "50", "--connections", "2", "--shape", "INPUT0:INT32:1,16",
                "--shape", "INPUT1:INT32:1,16", "--trace", str(trace)])
    stats = json.loads(out.strip().splitlines()[-1])
    assert stats["errors"] == 0 and stats["requests"] == 500 and stats["protocol"] == "grpc"
    # --trace: one row per measured request (the warm-up is not traced)
    with open(trace) as fh:
        rows = list(csv.DictReader(fh))
    assert len(rows) == 500 and min(int(r["start_ns"]) for r in rows) == 0
    assert all(r["ok"] == "1" and int(r["latency_ns"]) > 0 and 0 <= int(r["slot"]) < 8 for r in rows)


@pytest.mark.parametrize("name", CPP_EXAMPLES)
def test_cpp_examples(built, server, name):
    exe = os.path.join(built, name)
    if not os.path.exists(exe):
        pytest.skip(f"{name} not built")
    args = [exe, "-u", server.http_url]
    if name == "image_client":
        args += ["-m", "tiny_classifier", "-c", "3", "-s", "INCEPTION", "--synthetic", "2"]
    elif name == "ensemble_image_client":
        args += ["-c", "2", "--synthetic", "2"]
    out = _run(args)
    assert "PASS" in out


def _fnv1a(data):
    h = 1469598103934665603
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _write_adam7_png(path, pixels):
    """An interlaced 8-bit RGB PNG written by hand (Pillow does not write Adam7)."""
    import struct
    import zlib

    h, w, _ = pixels.shape
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2),
                           (0, 1, 1, 2)):
        sub = pixels[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) +
                 chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_cpp_png_reader_matches_pillow(built, tmp_path):
    """The zlib-based PNG reader of the C++ library against Pillow, over colour types, bit depths,
    filters (Pillow picks them adaptively for photographic content) and Adam7."""
    import numpy as np
    from PIL import Image

    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:37, 0:53]
    smooth = np.stack([(xx * 4 + yy) % 256, (yy * 6 + xx // 2) % 256, (xx + yy) * 2 % 256], axis=-1).astype(np.uint8)
    noisy = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    expected = {}

    def save(name, img, want):
        path = str(tmp_path / name)
        img.save(path)
        expected[path] = want

    for label, arr in (("smooth", smooth), ("noisy", noisy)):
        save(f"rgb_{label}.png", Image.fromarray(arr), arr)
        save(f"gray_{label}.png", Image.fromarray(arr[..., 0]), arr[..., 0])
        rgba = np.concatenate([arr, rng.integers(0, 256, size=(37, 53, 1), dtype=np.uint8)], axis=-1)
        save(f"rgba_{label}.png", Image.fromarray(rgba), arr)
        la = np.stack([arr[..., 1], rgba[..., 3]], axis=-1)
        save(f"la_{label}.png", Image.fromarray(la, "LA"), arr[..., 1])
    pal = Image.fromarray(smooth).quantize(colors=16)
    save("palette4.png", pal, np.asarray(pal.convert("RGB")))
    pal256 = Image.fromarray(noisy).quantize(colors=200)
    save("palette8.png", pal256, np.asarray(pal256.convert("RGB")))
    bilevel = Image.fromarray((smooth[..., 0] > 127).astype(np.uint8) * 255).convert("1")
    save("bilevel.png", bilevel, np.asarray(bilevel.convert("L")))
    deep = (rng.integers(0, 65536, size=(19, 23), dtype=np.uint16))
    save("gray16.png", Image.fromarray(deep), (deep >> 8).astype(np.uint8))
    inter = str(tmp_path / "adam7.png")
    _write_adam7_png(inter, noisy)
    expected[inter] = noisy
    ppm = str(tmp_path / "plain.ppm")
    with open(ppm, "wb") as fh:
        fh.write(b"P6\n# comment\n53 37\n255\n" + smooth.tobytes())
    expected[ppm] = smooth

    out = _run([os.path.join(built, "image_client"), "--decode-only", *expected])
    rows = {line.split()[0]: line.split()[1:] for line in out.strip().splitlines()}
    assert set(rows) == set(expected)
    for path, want in expected.items():
        h, w = want.shape[:2]
        c = 1 if want.ndim == 2 else 3
        assert rows[path] == [str(h), str(w), str(c), str(_fnv1a(want.tobytes()))], path

    # a flipped byte must be caught by the chunk checksum, not decoded
    bad = str(tmp_path / "bad.png")
    data = bytearray(open(str(tmp_path / "rgb_noisy.png"), "rb").read())
    data[len(data) // 2] ^= 0x40
    open(bad, "wb").write(bytes(data))
    p = subprocess.run([os.path.join(built, "image_client"), "--decode-only", bad], capture_output=True, text=True)
    assert p.returncode != 0 and "PNG" in (p.stdout + p.stderr)


def _jpeg_photo(h, w, seed=5):
    import numpy as np

    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([np.sin(x / 9.0) * 90 + 128 + y * 0.3, np.cos(y / 7.0 + x / 31.0) * 100 + 120, (x * y) % 251], -1)
    img = img + rng.normal(0, 12, img.shape)
    img[h // 4: h // 2, w // 3: w // 2] = [250, 10, 30]
    return np.clip(img, 0, 255).astype(np.uint8)


def test_cpp_jpeg_reader_matches_pillow(built, tmp_path):
    """The JPEG decoder of the C++ library against Pillow (libjpeg-turbo, default settings) over chroma
    subsampling modes, qualities, image sizes that are not multiples of the block size (down to 1x1), gray
    files, optimised Huffman tables, restart intervals and progressive files.  The decoder follows libjpeg's
    arithmetic, so the pixels are expected to be equal; the assertion allows a difference of 1 for Pillow
    builds on another libjpeg.  CMYK files are refused by name.  4:4:0 files (vertical-only subsampling) are
    not covered: Pillow cannot write them - parity unpinned."""
    import numpy as np
    from PIL import Image

    expected = {}
    for (h, w) in [(64, 64), (37, 53), (1, 1), (3, 2), (8, 17), (121, 159), (16, 5)]:
        im = Image.fromarray(_jpeg_photo(h, w))
        for ss in (0, 1, 2):
            for q in (30, 90):
                path = str(tmp_path / f"c_{h}x{w}_s{ss}_q{q}.jpg")
                im.save(path, quality=q, subsampling=ss)
                expected[path] = None
        path = str(tmp_path / f"g_{h}x{w}.jpg")
        im.convert("L").save(path, quality=80)
        expected[path] = None
    im = Image.fromarray(_jpeg_photo(200, 300))
    for name, kw in (("opt", {"optimize": True, "subsampling": 2}), ("rst", {"subsampling": 2, "restart_marker_blocks": 3}),
                     ("rstrow", {"subsampling": 1, "restart_marker_rows": 1}), ("q100", {"quality": 100, "subsampling": 0})):
        path = str(tmp_path / f"{name}.jpg")
        im.save(path, **{"quality": 85, **kw})
        expected[path] = None
    # progressive files (spectral selection + successive approximation), with restart intervals too
    for (h, w) in [(37, 53), (1, 1), (8, 17), (121, 159)]:
        pim = Image.fromarray(_jpeg_photo(h, w, seed=9))
        for ss in (0, 1, 2):
            path = str(tmp_path / f"p_{h}x{w}_s{ss}.jpg")
            pim.save(path, quality=80, subsampling=ss, progressive=True)
            expected[path] = None
        path = str(tmp_path / f"pg_{h}x{w}.jpg")
        pim.convert("L").save(path, quality=80, progressive=True)
        expected[path] = None
    path = str(tmp_path / "prst.jpg")
    im.save(path, quality=85, progressive=True, subsampling=2, restart_marker_blocks=3)
    expected[path] = None
    for path in expected:
        expected[path] = np.asarray(Image.open(path))
    dump = tmp_path / "dump"
    dump.mkdir()
    out = _run([os.path.join(built, "image_client"), "--decode-only", "--dump", str(dump), *expected])
    rows = {line.split()[0]: line.split()[1:] for line in out.strip().splitlines()}
    assert set(rows) == set(expected)
    exact = 0
    for path, want in expected.items():
        h, w = want.shape[:2]
        c = 1 if want.ndim == 2 else 3
        assert rows[path][:3] == [str(h), str(w), str(c)], path
        raw = open(str(dump / (os.path.basename(path) + (".pgm" if c == 1 else ".ppm"))), "rb").read()
        got = np.frombuffer(raw.split(b"\n", 3)[3], dtype=np.uint8).reshape(want.shape)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1, path
        exact += int(np.array_equal(got, want))
    from PIL import features

    if features.check_feature("libjpeg_turbo"):
        assert exact == len(expected)
    # a format nobody here reads is refused by name
    cmyk = str(tmp_path / "cmyk.jpg")
    im.convert("CMYK").save(cmyk, quality=85)
    p = subprocess.run([os.path.join(built, "image_client"), "--decode-only", cmyk], capture_output=True, text=True)
    assert p.returncode != 0 and "4-component" in (p.stdout + p.stderr)
    # damaged files end with a message, not a crash: truncated, and with bytes flipped all over the scan
    rng = np.random.default_rng(3)
    for k in range(48):
        data = open(str(tmp_path / ("opt.jpg" if k % 2 == 0 else "prst.jpg")), "rb").read()
        bad = bytearray(data[: len(data) - 1 - 20 * k] if k < 16 else data)
        if k >= 16:
            for at in rng.integers(2, len(bad) - 2, size=1 + k // 4):
                bad[int(at)] ^= int(rng.integers(1, 256))
        path = str(tmp_path / f"bad{k}.jpg")
        open(path, "wb").write(bytes(bad))
        p = subprocess.run([os.path.join(built, "image_client"), "--decode-only", path], capture_output=True, text=True)
        assert p.returncode in (0, 1), (k, p.returncode, p.stderr[-300:])


@pytest.mark.parametrize("protocol,mode", [("http", ()), ("grpc", ()), ("grpc", ("-a",))])
def test_cpp_image_client_jpeg_files(built, server, tmp_path, protocol, mode):
    """JPEG input over both protocols: the classes are the ones of the same pixels stored as PNG."""
    import numpy as np
    from PIL import Image

    jpgs, pngs = [], []
    for i in range(3):
        path = str(tmp_path / f"img{i}.jpg")
        Image.fromarray(_jpeg_photo(90 + 8 * i, 120, seed=20 + i)).save(path, quality=88, subsampling=2 - i)
        jpgs.append(path)
        png = str(tmp_path / f"img{i}.png")
        Image.fromarray(np.asarray(Image.open(path))).save(png)
        pngs.append(png)
    url = server.http_url if protocol == "http" else server.grpc_url
    common = ["-i", protocol, "-m", "tiny_classifier", "-b", "2", "-c", "2", "-s", "INCEPTION", *mode]
    out = _run([os.path.join(built, "image_client"), "-u", url, *common, *jpgs])
    ref = _run([os.path.join(built, "image_client"), "-u", url, *common, *pngs])
    pick = lambda text: [line for line in text.splitlines() if line.startswith("    ")]  # noqa: E731
    assert "PASS" in out and out.count("Image '") == 4 and pick(out) == pick(ref) and len(pick(out)) == 8


@pytest.mark.parametrize("protocol,mode", [("http", ()), ("grpc", ()), ("grpc", ("-a",))])
def test_cpp_image_client_png_files(built, server, tmp_path, protocol, mode):
    """PNG input over both protocols: same classes whatever the protocol / mode."""
    import numpy as np
    from PIL import Image

    rng = np.random.default_rng(11)
    files = []
    for i in range(3):
        path = str(tmp_path / f"img{i}.png")
        Image.fromarray(rng.integers(0, 256, size=(60 + 8 * i, 80, 3), dtype=np.uint8)).save(path)
        files.append(path)
    url = server.http_url if protocol == "http" else server.grpc_url
    out = _run([os.path.join(built, "image_client"), "-u", url, "-i", protocol, "-m", "tiny_classifier", "-b", "2",
                "-c", "2",
                "-s", "INCEPTION", *mode, *files])
    assert "PASS" in out and out.count("Image '") == 4
    ref = _run([os.path.join(built, "image_client"), "-u", server.http_url, "-m", "tiny_classifier", "-b", "2", "-c",
                "2",
                "-s", "INCEPTION", *files])
    pick = lambda text: [line for line in text.splitlines() if line.startswith("    ")]  # noqa: E731
    assert pick(out) == pick(ref)


def test_batch_image_decode_c_api(tmp_path):
    """``tca_decode_images`` behind ``ops.image_decode.decode_images``: a mixed batch (PNG of several colour
    types, JPEG, PNM) decoded on native threads equals Pillow; what the native readers refuse (a BMP file)
    falls back to Pillow; garbage is reported with the index of the file."""
    import io

    import numpy as np
    from PIL import Image

    from triton_client_amd.ops.image_decode import decode_images

    rgb = _jpeg_photo(61, 83)
    blobs, want = [], []
    for fmt, kw in (("PNG", {}), ("JPEG", {"quality": 90}), ("JPEG", {"quality": 70, "subsampling": 0}), ("PPM", {}),
                    ("JPEG", {"quality": 85, "progressive": True}), ("BMP", {})):  # BMP: read by Pillow
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, fmt, **kw)
        blobs.append(buf.getvalue())
        want.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    buf = io.BytesIO()
    Image.fromarray(rgb).convert("RGBA").save(buf, "PNG")
    blobs.append(buf.getvalue())
    want.append(rgb)
    buf = io.BytesIO()
    Image.fromarray(rgb[..., 0]).save(buf, "PNG")
    blobs.append(buf.getvalue())
    want.append(np.repeat(rgb[..., :1], 3, axis=2))
    for threads in (1, 4):
        got = decode_images(blobs, channels=3, threads=threads)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g.dtype == np.uint8 and g.shape == w.shape and np.abs(g.astype(int) - w.astype(int)).max() <= 1
    gray = decode_images(blobs[:1], channels=1)[0]
    assert gray.shape == (61, 83) and np.abs(gray.astype(int) - np.asarray(Image.fromarray(rgb).convert("L")).astype(int)).max() <= 1
    with pytest.raises(ValueError, match="image 1"):
        decode_images([blobs[0], b"not an image"], channels=3)
    assert decode_images([], channels=3) == []

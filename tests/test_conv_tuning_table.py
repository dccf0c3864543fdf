"""The tuned dispatch table of the convolution kernels (``tools/tune_conv.py``): well-formed
keys, known variant names, and the environment switch that turns it off."""
import importlib


def test_table_is_well_formed():
    from triton_client_amd.ops import conv

    table = conv.tuning_table()
    assert set(table) >= {"conv", "fused"}
    assert table["conv"], "the shipped table has entries"
    for key, variant in table["conv"].items():
        k, stride, h, w, cin, cout, res, batch = (int(v) for v in key.split(","))
        variant, _, staging = variant.partition("/")  # "tile" or "tile/lds2|lds3"
        assert variant in conv.VARIANTS and variant != "auto" and staging in ("", "lds2", "lds3")
        assert k in (1, 3) and stride in (1, 2) and res in (0, 1) and batch % 8 == 0 and 8 <= batch <= 128
        assert cin % 64 == 0 and cout % 64 == 0 and h == w
        if "x128" in variant:
            assert cout % 128 == 0
        if "x256" in variant:
            assert cout % 256 == 0
    for key, mode in table["fused"].items():
        h, w, cin, mid, cout, stride, n3, batch = (int(v) for v in key.split(","))
        assert mode in ("none", "64", "128") and mid in (64, 128) and n3 in (0, 64, 128)
    assert conv.tuned_block_mode(56, 56, 64, 64, 256, 1, 64, 128) in ("none", "64", "128")
    assert conv.tuned_block_mode(3, 3, 64, 64, 256, 1, 64, 128) is None  # not a measured shape


def test_table_can_be_switched_off(monkeypatch):
    from triton_client_amd.ops import conv

    monkeypatch.setenv("TCA_CONV_TUNING", "0")
    monkeypatch.setattr(conv, "_tuning", None)
    assert conv.tuning_table() == {"conv": {}, "fused": {}}
    monkeypatch.setattr(conv, "_tuning", None)
    monkeypatch.delenv("TCA_CONV_TUNING")
    importlib.reload(conv)
    assert conv.tuning_table()["conv"]

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
    for key, wins in table.get("rows", {}).items():
        n3, proj, batch = (int(v) for v in key.split(","))
        assert isinstance(wins, bool) and n3 in (64, 128) and proj in (0, 1) and 1 <= batch <= 128
        assert not (proj and n3 != 64)
    assert conv.tuned_block_mode(56, 56, 64, 64, 256, 1, 64, 128) in ("none", "64", "128")
    assert conv.tuned_block_mode(3, 3, 64, 64, 256, 1, 64, 128) is None  # not a measured shape


def test_rows_kernel_choice(monkeypatch):
    """The row-organised block kernel is chosen from the table, by rule for unmeasured batches, and by the switch."""
    from triton_client_amd.ops import conv

    monkeypatch.delenv("TCA_CONV_ROWS", raising=False)
    monkeypatch.setattr(conv, "_tuning", {"conv": {}, "fused": {}, "rows": {"64,0,128": True, "64,1,8": False}})
    assert conv.tuned_rows(64, False, 128) is True and conv.tuned_rows(64, True, 8) is False
    assert conv.tuned_rows(64, True, 5) is True          # rule: projection inside -> rows
    assert conv.tuned_rows(128, False, 100) is False     # rule: 128 next outputs only up to batch 32
    monkeypatch.setenv("TCA_CONV_ROWS", "0")
    assert conv.tuned_rows(64, False, 128) is False
    monkeypatch.setenv("TCA_CONV_ROWS", "1")
    assert conv.tuned_rows(128, False, 100) is True


def test_permute_mid_k_is_a_permutation():
    """The K order of the row-organised kernel's expansion weights: a permutation of the first 64 columns that
    follows the accumulator layout (lane quarter q, register r of 16-channel fragment f = channel 16 f + 4 q + r)."""
    import torch

    from triton_client_amd.ops import conv

    w = torch.arange(128, dtype=torch.float16).repeat(3, 1)
    p = conv.permute_mid_k(w)
    assert p.shape == w.shape and sorted(p[0, :64].tolist()) == list(range(64))
    assert p[0, 64:].tolist() == list(range(64, 128))
    for s in range(2):
        for q in range(4):
            for j in range(8):
                assert p[0, 32 * s + 8 * q + j].item() == 16 * (2 * s + j // 4) + 4 * q + j % 4


def test_table_can_be_switched_off(monkeypatch):
    from triton_client_amd.ops import conv

    monkeypatch.setenv("TCA_CONV_TUNING", "0")
    monkeypatch.setattr(conv, "_tuning", None)
    assert conv.tuning_table() == {"conv": {}, "fused": {}}
    monkeypatch.setattr(conv, "_tuning", None)
    monkeypatch.delenv("TCA_CONV_TUNING")
    importlib.reload(conv)
    assert conv.tuning_table()["conv"]

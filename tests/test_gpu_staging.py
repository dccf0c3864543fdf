"""Native wire stager (pinned ring + copy stream + device staging) against plain copies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def test_stager_places_every_tensor(torch):
    from triton_client_amd.ops.staging import WireStager

    rng = np.random.default_rng(3)
    row = 3 * 224 * 224 * 2
    dst = torch.zeros(64 * row, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    st = WireStager(0, entry_bytes=8 * row, dev_bytes=64 * row, entries=4)
    try:
        expect = np.zeros(64 * row, dtype=np.uint8)
        for step in range(12):  # many more bytes than the ring holds: entries are recycled
            rows = [int(r) for r in rng.integers(1, 9, size=6)]
            arrays, offsets, cursor = [], [], 0
            for r in rows:
                a = rng.integers(0, 256, size=r * row, dtype=np.uint8)
                arrays.append(a)
                offsets.append(cursor * row)
                expect[cursor * row: (cursor + r) * row] = a
                cursor += r + (1 if step % 2 else 0)  # odd steps leave gaps: ranges are not merged across them
            st.stage(arrays, offsets, dst.data_ptr(), stream)
            stream.synchronize()
            np.testing.assert_array_equal(dst.cpu().numpy(), expect)
        stats = st.stats()
        assert stats["tensors"] == 72 and stats["bytes"] > 4 * 8 * row
    finally:
        st.close()


def test_stager_grows_ring_and_device_buffers(torch):
    from triton_client_amd.ops.staging import WireStager

    st = WireStager(0, entry_bytes=4096, dev_bytes=8192, entries=2)
    try:
        dst = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.Stream()
        a = np.arange(1 << 18, dtype=np.uint8)  # larger than a ring entry and than the staging buffer
        b = np.full(1000, 7, dtype=np.uint8)
        st.stage([a, b], [0, 1 << 19], dst.data_ptr(), stream)
        stream.synchronize()
        got = dst.cpu().numpy()
        np.testing.assert_array_equal(got[: 1 << 18], a)
        np.testing.assert_array_equal(got[1 << 19: (1 << 19) + 1000], b)
        assert not got[(1 << 18): (1 << 19)].any()
    finally:
        st.close()

"""Classification extension on the GPU: top-k and the ``"value:index[:label]"``
BYTES payload (reference consumers: ``src/c++/examples/image_client.cc:256-276``,
``src/python/examples/grpc_image_client.py:227-243``)."""

from __future__ import annotations

import numpy as np

from . import _bind
from ._device import check, lib

_TORCH_CODE = None


def _code(t):
    import torch

    global _TORCH_CODE
    if _TORCH_CODE is None:
        _TORCH_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    return _TORCH_CODE.get(t.dtype)


def topk(logits, k):
    """``(values fp32 [rows,k], indices int32 [rows,k])`` of the k largest
    entries per row (ties -> lowest index), computed by the HIP kernel."""
    import torch

    x = logits.reshape(-1, logits.shape[-1]).contiguous()
    code = _code(x)
    rows, cols = x.shape
    k = min(int(k), cols)
    if code is None or cols > 12288:
        vals, idx = torch.topk(x.float(), k, dim=-1)
        return vals, idx.to(torch.int32)
    values = torch.empty((rows, k), dtype=torch.float32, device=x.device)
    indices = torch.empty((rows, k), dtype=torch.int32, device=x.device)
    rc = lib().tca_topk_launch(x.data_ptr(), code, rows, cols, k, values.data_ptr(), indices.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    check(rc, "tca_topk_launch")
    return values, indices


class LabelTable:
    """Class labels resident on the GPU (concatenated bytes + offsets)."""

    def __init__(self, labels, device):
        import torch

        blobs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in labels]
        offs = np.zeros(len(blobs) + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(b) for b in blobs])
        data = np.frombuffer(b"".join(blobs) or b"\0", dtype=np.uint8).copy()
        self.count = len(blobs)
        self.data = torch.from_numpy(data).to(device)
        self.offsets = torch.from_numpy(offs.view(np.int32)).to(device)
        self.max_len = max((len(b) for b in blobs), default=0)


def classification_bytes(values, indices, label_table=None, out=None):
    """Format ``values/indices`` ([rows,k]) into the BYTES wire payload on the GPU.

    Returns ``(buffer uint8, size int)``.  ``out`` may be a preallocated uint8
    tensor (e.g. a view of a hipshm output region).
    """
    import torch

    rows, k = values.shape
    entries = rows * k
    cap = entries * (4 + 64 + ((label_table.max_len + 1) if label_table is not None else 0))
    if out is None:
        out = torch.empty(cap, dtype=torch.uint8, device=values.device)
    size = torch.zeros(1, dtype=torch.int64, device=values.device)
    rc = lib().tca_classify_strings_launch(
        values.contiguous().data_ptr(), indices.contiguous().data_ptr(), rows, k,
        label_table.data.data_ptr() if label_table is not None else None,
        label_table.offsets.data_ptr() if label_table is not None else None,
        label_table.count if label_table is not None else 0, out.data_ptr(), out.numel(), size.data_ptr(),
        torch.cuda.current_stream().cuda_stream)
    check(rc, "tca_classify_strings_launch")
    n = int(size.item())
    if n > out.numel():
        raise ValueError(f"classification payload needs {n} bytes but the destination holds {out.numel()}")
    return out, n


def topk_strings(logits, k, labels=None):
    """Object ndarray ``[rows,k]`` (or ``[k]`` for a vector) of the classification strings.

    Top-k runs on the GPU; only ``rows*k`` (value, index) pairs cross PCIe.
    """
    vals, idx = topk(logits, k)
    v = vals.cpu().numpy()
    i = idx.cpu().numpy()
    rows, kk = v.shape
    out = np.empty((rows, kk), dtype=np.object_)
    for r in range(rows):
        for c in range(kk):
            j = int(i[r, c])
            text = "%f:%d" % (float(v[r, c]), j)
            if labels is not None and 0 <= j < len(labels):
                text += ":" + labels[j]
            out[r, c] = text.encode("utf-8")
    return out if logits.dim() > 1 else out.reshape(kk)

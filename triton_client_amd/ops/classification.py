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
    buf, size = _launch_strings(values, indices, label_table, out)
    n = int(size.item())
    if n > buf.numel():
        raise ValueError(f"classification payload needs {n} bytes but the destination holds {buf.numel()}")
    return buf, n


def _launch_strings(values, indices, label_table, out=None, size=None):
    import torch

    rows, k = values.shape
    entries = rows * k
    if out is None:
        cap = entries * (4 + 64 + ((label_table.max_len + 1) if label_table is not None else 0))
        out = torch.empty(cap, dtype=torch.uint8, device=values.device)
    if size is None:
        size = torch.zeros(1, dtype=torch.int64, device=values.device)
    rc = lib().tca_classify_strings_launch(
        values.contiguous().data_ptr(), indices.contiguous().data_ptr(), rows, k,
        label_table.data.data_ptr() if label_table is not None else None,
        label_table.offsets.data_ptr() if label_table is not None else None,
        label_table.count if label_table is not None else 0, out.data_ptr(), out.numel(), size.data_ptr(),
        torch.cuda.current_stream().cuda_stream)
    check(rc, "tca_classify_strings_launch")
    return out, size


def classification_payload_async(logits, k, label_table=None):
    """Enqueue top-k + string formatting + one device-to-host copy on the
    current stream.  Returns ``(pinned uint8 tensor, shape)``; after the stream
    (or an event recorded behind this call) completed, hand both to
    :func:`finish_payload`.  Payload size and payload share the buffer, so
    nothing needs a second round trip."""
    import torch

    vals, idx = topk(logits, k)
    rows, kk = vals.shape
    per = 4 + 64 + ((label_table.max_len + 1) if label_table is not None else 0)
    buf = torch.empty(8 + rows * kk * per, dtype=torch.uint8, device=vals.device)
    _launch_strings(vals, idx, label_table, buf[8:], buf[:8].view(torch.int64))
    host = torch.empty(buf.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(buf, non_blocking=True)
    # the caching allocator may hand vals/idx/buf to a later request once they
    # go out of scope; uses on the same stream stay ordered behind this copy
    return host, ([rows, kk] if logits.dim() > 1 else [kk])


def finish_payload(host, shape):
    """``(bytes, shape)`` of a completed :func:`classification_payload_async`."""
    arr = host.numpy()
    n = int(arr[:8].view(np.int64)[0])
    return arr[8: 8 + n].tobytes(), shape


def classification_payload(logits, k, label_table=None):
    """Synchronous form: the BYTES wire payload (host ``bytes``) and its shape."""
    import torch

    host, shape = classification_payload_async(logits, k, label_table)
    torch.cuda.current_stream().synchronize()
    return finish_payload(host, shape)


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

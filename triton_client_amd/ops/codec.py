"""Device-side tensor codec: BYTES pack / index / gather and BF16 conversion.

GPU counterparts of the host codec in :mod:`triton_client_amd.utils`
(reference per-element Python loops: ``tritonclient/utils/__init__.py:227-246``,
``:267-276``, ``:310-320``).  Operands are torch tensors on the target GPU
(``uint8`` for byte streams); every function enqueues on torch's current
stream and returns torch tensors.
"""

from __future__ import annotations

from ._device import check, lib


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def bytes_pack(payload, offsets):
    """CSR strings -> BYTES wire stream.

    payload : uint8 [total] concatenated element bytes
    offsets : int64/uint64 [n+1] element boundaries inside ``payload``
    Returns uint8 [total + 4*n].
    """
    import torch

    n = offsets.numel() - 1
    total = int(payload.numel()) + 4 * n
    out = torch.empty(total, dtype=torch.uint8, device=payload.device)
    if n > 0:
        offsets = offsets.to(torch.int64).contiguous()
        rc = lib().tca_bytes_pack_launch(payload.data_ptr(), offsets.data_ptr(), n, out.data_ptr(), total, _stream())
        check(rc, "tca_bytes_pack_launch")
    return out


def lengths_to_offsets(lengths, add_each=0):
    """Exclusive scan of uint32 lengths (+``add_each`` per element) -> int64 [n+1]."""
    import torch

    n = lengths.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=lengths.device)
    lens32 = lengths.to(torch.int32).contiguous()
    rc = lib().tca_exclusive_scan_u64_launch(lens32.data_ptr(), n, int(add_each), out.data_ptr(), _stream())
    check(rc, "tca_exclusive_scan_u64_launch")
    return out


def bytes_pack_from_lengths(payload, lengths):
    """Like :func:`bytes_pack` when only per-element lengths are known."""
    return bytes_pack(payload, lengths_to_offsets(lengths))


# streams longer than this are indexed by the chunk-parallel kernels (one workgroup per 4 KiB)
PARALLEL_INDEX_MIN_BYTES = 32 << 10


def bytes_index(stream_bytes, max_elems, parallel=None):
    """Walk a BYTES wire stream on the GPU.

    Returns ``(offsets int64 [count], lens int32 [count])`` of the payloads.
    Raises ``ValueError`` for a malformed stream.  At most ``max_elems``
    elements are indexed (pass ``prod(shape)``).  ``parallel`` forces the
    chunk-parallel (True) or the single-workgroup (False) kernel; by default
    the stream length decides.
    """
    import torch

    dev = stream_bytes.device
    size = stream_bytes.numel()
    offsets = torch.empty(max(1, max_elems), dtype=torch.int64, device=dev)
    lens = torch.empty(max(1, max_elems), dtype=torch.int32, device=dev)
    status = torch.zeros(3, dtype=torch.int64, device=dev)
    if parallel is None:
        parallel = size >= PARALLEL_INDEX_MIN_BYTES
    if parallel and size:
        need = int(lib().tca_bytes_index_workspace_bytes(size))
        work = torch.empty((need + 7) // 8, dtype=torch.int64, device=dev)
        rc = lib().tca_bytes_index_parallel_launch(stream_bytes.data_ptr(), size, max_elems, offsets.data_ptr(),
                                                   lens.data_ptr(), status.data_ptr(), work.data_ptr(),
                                                   work.numel() * 8,
                                                   _stream())
        check(rc, "tca_bytes_index_parallel_launch")
    else:
        rc = lib().tca_bytes_index_launch(stream_bytes.data_ptr(), size, max_elems, offsets.data_ptr(),
                                          lens.data_ptr(), status.data_ptr(), _stream())
        check(rc, "tca_bytes_index_launch")
    count, bad, cursor = (int(v) for v in status.tolist())
    if bad:
        raise ValueError(f"malformed BYTES tensor near offset {cursor}")
    return offsets[:count], lens[:count]


def bytes_gather(stream_bytes, offsets, lens, pitch):
    """Copy each payload into a zero padded row of ``pitch`` bytes: uint8 [n, pitch]."""
    import torch

    n = offsets.numel()
    out = torch.empty((n, pitch), dtype=torch.uint8, device=stream_bytes.device)
    if n:
        rc = lib().tca_bytes_gather_launch(stream_bytes.data_ptr(), offsets.contiguous().data_ptr(),
                                           lens.contiguous().data_ptr(), n, out.data_ptr(), int(pitch), _stream())
        check(rc, "tca_bytes_gather_launch")
    return out


def fp32_to_bf16(src, round_nearest=False):
    """fp32 tensor -> bf16 bit patterns (int16 tensor).  ``round_nearest=False``
    truncates, which is what the wire codec does."""
    import torch

    src = src.contiguous()
    out = torch.empty(src.shape, dtype=torch.int16, device=src.device)
    rc = lib().tca_fp32_to_bf16_launch(src.data_ptr(), out.data_ptr(), src.numel(), 1 if round_nearest else 0,
                                       _stream())
    check(rc, "tca_fp32_to_bf16_launch")
    return out


def bf16_to_fp32(src):
    """bf16 bit patterns (int16 / bfloat16 tensor) -> fp32 tensor."""
    import torch

    src = src.contiguous()
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    rc = lib().tca_bf16_to_fp32_launch(src.data_ptr(), out.data_ptr(), src.numel(), _stream())
    check(rc, "tca_bf16_to_fp32_launch")
    return out

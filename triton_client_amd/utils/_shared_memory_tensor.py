"""DLPack view of (a slice of) a shared-memory region.

Reference: ``tritonclient/utils/_shared_memory_tensor.py:34-87``.  The view is
non-owning; ``device_id == -1`` means host (system shm) memory, any other value
is an MI355X ordinal and the tensor is exported as ``kDLROCM`` so that
``torch.from_dlpack`` on PyTorch-ROCm aliases the region without a copy.
"""

from __future__ import annotations

from . import _dlpack


class SharedMemoryTensor:
    def __init__(self, dtype, shape, base_addr, offset, byte_size, device_id, owner=None):
        self._dl_dtype = _dlpack.triton_to_dlpack_dtype(dtype)
        self._shape = [int(s) for s in shape]
        need = _dlpack.get_byte_size(self._dl_dtype, len(self._shape), self._shape)
        if need > byte_size:
            raise ValueError(
                "shared memory region of {} bytes cannot hold a {} tensor of shape {} ({} bytes)".format(
                    byte_size, dtype, self._shape, need
                )
            )
        self._address = int(base_addr) + int(offset)
        self._device_id = int(device_id)
        self._owner = owner

    def __dlpack__(self, stream=None, **_ignored):
        # The region is synchronised by the shm module before it is handed out,
        # so no stream ordering is required from the consumer.
        dev_type, dev_id = self.__dlpack_device__()
        return _dlpack.make_capsule(
            self._address, self._dl_dtype, self._shape, dev_type, dev_id, owner=self._owner
        )

    def __dlpack_device__(self):
        if self._device_id == -1:
            return (_dlpack.DLDeviceType.kDLCPU, 0)
        return (_dlpack.DLDeviceType.kDLROCM, self._device_id)

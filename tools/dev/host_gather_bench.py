#!/usr/bin/env python3
"""How fast the batched copy kernel pulls request tensors out of page-locked host memory (what the native
serving path does for tensors that arrive inside messages), against hipMemcpyAsync of the same bytes, and
the way back for the output rows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from triton_client_amd.ops import _device  # noqa: E402
from triton_client_amd.ops.copy import batched_copy  # noqa: E402


def timed(fn, stream, iters=30):
    fn()
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    stream.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    row_in, row_out = 3 * 224 * 224 * 2, 4000
    arena = _device.PinnedBuffer(256 << 20)
    stream = torch.cuda.Stream()
    dev_in = torch.empty(128 * row_in, dtype=torch.uint8, device="cuda")
    dev_out = torch.empty(128 * row_out, dtype=torch.uint8, device="cuda")
    pinned_t = torch.empty(128 * row_in, dtype=torch.uint8).pin_memory()
    for requests, rows in ((8, 8), (16, 8), (64, 1), (128, 1), (4, 32)):
        nbytes = rows * row_in
        stride = (nbytes + 262143) // 262144 * 262144
        gather = [(arena.ptr + i * stride, dev_in.data_ptr() + i * nbytes, nbytes) for i in range(requests)]
        scatter = [(dev_out.data_ptr() + i * rows * row_out, arena.ptr + i * stride, rows * row_out) for i in range(requests)]
        total = requests * nbytes
        t_g = timed(lambda: batched_copy(gather, stream), stream)
        t_s = timed(lambda: batched_copy(scatter, stream), stream)

        def memcpy():
            with torch.cuda.stream(stream):
                for i in range(requests):
                    dev_in[i * nbytes:(i + 1) * nbytes].copy_(pinned_t[i * nbytes:(i + 1) * nbytes], non_blocking=True)

        t_m = timed(memcpy, stream)
        d2d = [(dev_in.data_ptr() + ((i + 1) % requests) * nbytes, dev_in.data_ptr() + i * nbytes, nbytes) for i in range(requests)]
        t_d = timed(lambda: batched_copy(d2d[:max(1, requests // 2)], stream), stream)
        print(f"{requests:4d} requests x {rows:3d} rows ({total / 1e6:6.1f} MB): kernel gather from host {t_g * 1e6:7.1f} us "
              f"({total / t_g / 1e9:5.1f} GB/s), hipMemcpyAsync x{requests} {t_m * 1e6:7.1f} us ({total / t_m / 1e9:5.1f} GB/s), "
              f"scatter of outputs to host {t_s * 1e6:6.1f} us, device-to-device half {t_d * 1e6:6.1f} us", flush=True)


if __name__ == "__main__":
    main()

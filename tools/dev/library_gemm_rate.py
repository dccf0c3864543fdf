#!/usr/bin/env python3
"""What the library GEMM (hipBLASLt through torch.matmul) reaches on the GEMM shapes of ResNet-50's stage-3/4
layers at batch 128 and on long-K versions of them: the yardstick for the convolution kernel's K loop."""
import time

import torch


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    shapes = [(25088, 1024, 1024), (25088, 1024, 4096), (25088, 256, 1024), (25088, 256, 2304), (25088, 256, 4096),
              (25088, 1024, 256), (6272, 512, 4608), (6272, 2048, 512), (6272, 512, 2048), (8192, 8192, 8192)]
    for (m, n, k) in shapes:
        a = torch.randn(m, k, device=dev).half()
        b = torch.randn(n, k, device=dev).half()
        out = torch.empty(m, n, device=dev, dtype=torch.float16)
        with torch.cuda.stream(st):
            for _ in range(3):
                torch.matmul(a, b.t(), out=out)
            st.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                torch.matmul(a, b.t(), out=out)
            st.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / reps
        print(f"M {m:6d} N {n:5d} K {k:5d}: {us:8.1f} us {2.0 * m * n * k / us / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()

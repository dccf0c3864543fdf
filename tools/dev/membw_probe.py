"""Practical memory roofs on this GPU for the tensor sizes of ResNet-50 stage 1 at batch 128
(205 MB fp16 tensors): copy, add + ReLU (2 reads, 1 write), and a strided variant that touches
128-byte segments of 512-byte rows the way a 64-channel chunk of a 256-channel NHWC tensor does."""
import torch

dev = "cuda"
n = 128 * 56 * 56 * 256
x = torch.randn(n, device=dev).half()
r = torch.randn(n, device=dev).half()
y = torch.empty_like(x)


def timed(fn, nbytes, label, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{label:46s} {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s")


timed(lambda: y.copy_(x), 2 * n * 2, "copy 205 MB -> 205 MB")
timed(lambda: torch.add(x, r, out=y), 3 * n * 2, "add: 2 reads + 1 write")
x2, r2, y2 = x.view(-1, 4, 64), r.view(-1, 4, 64), y.view(-1, 4, 64)
def chunks():
    for c in range(4):
        torch.add(x2[:, c], r2[:, c], out=y2[:, c])
timed(chunks, 3 * n * 2, "add in four 64-channel column chunks (4 launches)")

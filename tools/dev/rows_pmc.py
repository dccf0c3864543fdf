#!/usr/bin/env python3
"""A few launches of one stage-1 bottleneck tail (batch x 56 x 56) for counter collection:
``rows_pmc.py KIND BATCH N3 PROJ`` with KIND = rows | chunked."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from triton_client_amd.ops import conv

kind, batch, n3, proj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
h = w = 56


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


x = nhwc(torch.randn(batch, 64, h, w, device=dev).half())
w1 = nhwc((torch.randn(64, 64, 3, 3, device=dev) / 24.0).half())
b1 = torch.randn(64, device=dev).half()
w2 = (torch.randn(256, 64, device=dev) / 8.0).half()
b2 = torch.randn(256, device=dev).half()
w3 = (torch.randn(n3, 256, device=dev) / 16.0).half()
b3 = torch.randn(n3, device=dev).half()
x0 = nhwc(torch.randn(batch, 64, h, w, device=dev).half()) if proj else None
res = None if proj else nhwc(torch.randn(batch, 256, h, w, device=dev).half())
wcat = torch.cat([w2, (torch.randn(256, 64, device=dev) / 8.0).half()], dim=1).contiguous() if proj else w2
wperm = conv.permute_mid_k(wcat)
out = nhwc(torch.empty(batch, 256, h, w, device=dev, dtype=torch.float16))
for _ in range(5):
    if kind == "rows":
        conv.bottleneck_rows(x, w1, b1, wperm, b2, res, w3, b3, proj_input=x0, out=out)
    else:
        conv.conv2d_pointwise(x, w1, b1, wcat, b2, res, relu=True, padding=1, out=out, next_weight=w3, next_bias=b3,
                              proj_input=x0, tile_m=128)
torch.cuda.synchronize()
print(f"{kind} batch {batch} n3 {n3} proj {proj}: done")

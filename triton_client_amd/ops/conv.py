"""NHWC fp16 convolution pieces with fused epilogues (inference).

``conv1x1``   1x1 convolution as a hand-written MFMA GEMM whose epilogue adds
              the bias, optionally the residual branch, applies ReLU and writes
              the activation once (kernel: ``csrc/hip/conv_gemm.hip``).
``bias_act_`` the same epilogue as ONE in-place pass for convolutions computed
              by a library (3x3 / 7x7): replaces the separate bias-add,
              residual-add and ReLU passes of an eager forward.
``bias_relu_maxpool``  the stem's bias + ReLU folded into its 3x3/2 max pooling.

Tensors are torch CUDA tensors in channels-last memory format, i.e. logical
``[N, C, H, W]`` with strides of a ``[N, H, W, C]`` array.  The reference has no
model code (its models run in the server it talks to, e.g.
``src/python/examples/image_client.py:59-150`` only inspects metadata); this is
the serving side of that contract on MI355X.
"""

from __future__ import annotations

from ._device import check, lib

VARIANTS = {"auto": 0, "128x128": 1, "128x64": 2, "64x64": 3, "128x128w8": 4, "256x128w8": 5,
            "128x64w8": 6, "64x64w8": 7, "128x256w8": 8, "256x128w16": 9, "256x256w16": 10, "128x256w16": 11,
            "128x256halo": 12, "128x128halo": 13}
# operand staging of the GEMM kernel: "lds" = direct global->LDS DMA (default),
# "reg" = through registers + ds_write (kept for A/B measurements)
_REG_STAGING = 16


def _variant_code(variant, staging=None):
    """Kernel selector word: tile variant, staging ("lds" | "lds2" | "lds3" pin the
    number of LDS stages of the DMA pipeline, "reg" = register staging)."""
    import os

    code = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    staging = staging or os.environ.get("TCA_CONV_STAGING", "lds")
    if staging not in ("lds", "lds2", "lds3", "reg"):
        raise ValueError("staging must be 'lds', 'lds2', 'lds3' or 'reg'")
    if staging == "reg":
        return code | _REG_STAGING
    return code | (int(staging[3:]) << 5 if len(staging) > 3 else 0)


def gemm_pays(pixels, cout, k_total, split_k=None):
    """Whether the MFMA GEMM kernel beats library convolution + epilogue pass
    for a layer with ``pixels`` output pixels.  With split-K (the default) it
    always does on the layers measured (``tools/conv_bench.py``).  Without it,
    few 64x64 output tiles and a long K loop mean every CU runs at most one block
    and the kernel time is the K loop's latency chain (about a third of a
    microsecond per 64-wide K step), which is where the library's own split-K
    kernels win."""
    if SPLIT_K if split_k is None else split_k:
        return True
    tiles = ((pixels + 63) // 64) * (cout // 64)
    ksteps = k_total // 64
    return ksteps <= 40 or tiles >= 2.5 * ksteps


# Split-K (A/B switch): layers whose output tiles fill less than half of the CUs share
# their K loop out over several blocks per tile, summed inside the same launch.
SPLIT_K = __import__("os").environ.get("TCA_CONV_SPLITK", "1") != "0"
_SPLITK_PARTIAL_BYTES = 32 << 20   # 128 tiles x 8 runs x 64x64 fp32 = 16 MiB; room for 128x128 tiles
_SPLITK_ARRIVAL_SLOTS = 4096
_splitk_workspaces = {}


def _splitk_workspace(device):
    """Per-stream workspace of the split-K fix-up: partial tiles and one arrival
    counter per output tile (zeroed once; the kernel leaves it zeroed).  Launches on
    one stream are ordered, so they share it; another stream (another execution
    instance of a model, with its own captured graphs) gets its own."""
    import torch

    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _splitk_workspaces.get(key)
    if ws is None:
        # allocated outside any graph capture pool on purpose: the tensors live as long as the process
        partials = torch.empty(_SPLITK_PARTIAL_BYTES // 4, dtype=torch.float32, device=device)
        arrivals = torch.zeros(_SPLITK_ARRIVAL_SLOTS, dtype=torch.int32, device=device)
        ws = _splitk_workspaces[key] = (partials, arrivals)
    return ws


def release_splitk_workspaces(device, streams):
    """Drops the split-K workspaces of ``streams`` (torch streams of ``device``): a model that is unloaded takes
    its streams with it, and their workspaces (32 MiB each) would otherwise stay for the life of the process."""
    for st in streams:
        _splitk_workspaces.pop((device.index, st.cuda_stream), None)


_tuning = None


def tuning_table():
    """The tuned dispatch table measured by ``tools/tune_conv.py`` on an MI355X
    (``conv_tuning_gfx950.json`` next to this module; ``TCA_CONV_TUNING=0`` ignores it):
    ``{"conv": {"k,stride,h,w,cin,cout,residual,batch": variant}, "fused": {...: "none|64|128"}}``."""
    global _tuning
    if _tuning is None:
        import json
        import os

        # TCA_CONV_TUNING_FILE: another table (A/B runs of a freshly tuned one)
        path = os.environ.get("TCA_CONV_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                      "conv_tuning_gfx950.json")
        table = {"conv": {}, "fused": {}}
        if os.environ.get("TCA_CONV_TUNING", "1") != "0" and os.path.exists(path):
            with open(path) as fh:
                table = json.load(fh)
        _tuning = table
    return _tuning


def tuned_block_mode(h, w, cin, mid, cout, stride, n3, batch):
    """"none" | "64" | "128" for a bottleneck block's fused kernel, or None when not measured."""
    return tuning_table()["fused"].get(f"{h},{w},{cin},{mid},{cout},{stride},{n3},{batch}")


def tuned_rows(n3, proj, batch):
    """Whether the row-organised bottleneck kernel serves a stage-1 block (``n3`` outputs of the attached next
    convolution, ``proj`` = projection shortcut inside) at this batch: the measured table (``tools/tune_rows.py``)
    where it has the batch, else the rule those measurements suggest - with the projection inside always; a
    residual and 64 next outputs up to batch 64; 128 next outputs up to batch 32."""
    import os

    forced = os.environ.get("TCA_CONV_ROWS")  # A/B switch: 0 = never, 1 = wherever the kernel applies
    if forced in ("0", "1"):
        return forced == "1"
    hit = tuning_table().get("rows", {}).get(f"{n3},{int(bool(proj))},{batch}")
    if hit is not None:
        return bool(hit)
    if proj:
        return True
    return batch <= (64 if n3 == 64 else 32)


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def _is_nhwc(t):
    import torch

    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def _launch(x, weight, bias, residual, out, n, h, w, r, stride, padding, cin, cout, relu, variant, staging, split_k,
            tuned=True):
    """``split_k``: None = automatic when enabled (``SPLIT_K``), 1 = off, n > 1 = exactly n."""
    if tuned and variant in (0, "auto"):
        # a measured winner for exactly this layer and batch beats the kernel's threshold rule
        variant = tuning_table()["conv"].get(f"{r},{stride},{h},{w},{cin},{cout},{int(residual is not None)},{n}", 0)
    if isinstance(variant, str) and "/" in variant:
        # "tile/staging" (a table entry that also pins the number of LDS stages); an explicit argument wins
        variant, tuned_staging = variant.split("/", 1)
        staging = staging or tuned_staging
    code = _variant_code(variant, staging)
    splits = (0 if SPLIT_K else 1) if split_k is None else int(split_k)
    if splits == 1:
        partials = arrivals = None
    else:
        partials, arrivals = _splitk_workspace(x.device)
    rc = lib().tca_conv_gemm_splitk_launch(
        x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
        residual.data_ptr() if residual is not None else None, out.data_ptr(), n, h, w, int(r), int(stride),
        int(padding), cin, cout, 1 if relu else 0, code, splits,
        partials.data_ptr() if partials is not None else None, partials.numel() * 4 if partials is not None else 0,
        arrivals.data_ptr() if arrivals is not None else None, arrivals.numel() if arrivals is not None else 0,
        _stream())
    check(rc, "tca_conv_gemm_splitk_launch")


def conv1x1(x, weight, bias=None, residual=None, relu=False, stride=1, variant=0, out=None, staging=None,
            split_k=None, tuned=True):
    """``act(conv1x1(x, weight) + bias (+ residual))``.

    x        fp16 ``[N, Cin, H, W]`` channels-last
    weight   fp16 ``[Cout, Cin]`` (or ``[Cout, Cin, 1, 1]``), contiguous
    bias     fp16 ``[Cout]`` or None
    residual fp16 ``[N, Cout, OH, OW]`` channels-last or None
    Cin must be a multiple of 64 and Cout of 64 (128 for the 128x128 tile).
    """
    import torch

    if x.dtype != torch.float16 or weight.dtype != torch.float16:
        raise TypeError("conv1x1 takes fp16 tensors")
    if not _is_nhwc(x):
        raise ValueError("x must be a channels-last [N, C, H, W] tensor")
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    if weight.numel() != cout * cin or not weight.is_contiguous():
        raise ValueError(f"weight must be a contiguous [{cout}, {cin}] matrix")
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    if out is None:
        out = torch.empty((n, cout, oh, ow), dtype=torch.float16, device=x.device,
                          memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, cout, oh, ow) or not _is_nhwc(out):
        raise ValueError("out must be a channels-last tensor of the output shape")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not _is_nhwc(residual)
                                 or residual.dtype != torch.float16):
        raise ValueError("residual must match the output (fp16, channels-last)")
    if bias is not None and (bias.numel() != cout or bias.dtype != torch.float16 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous fp16 vector of Cout elements")
    _launch(x, weight, bias, residual, out, n, h, w, 1, stride, 0, cin, cout, relu, variant, staging, split_k, tuned)
    return out


def conv2d(x, weight, bias=None, residual=None, relu=False, stride=1, padding=0, variant=0, out=None,
           staging=None, split_k=None, tuned=True):
    """``act(conv2d(x, weight) + bias (+ residual))`` as an implicit GEMM on the
    MFMA kernel (square kernels, e.g. 3x3 with padding 1).

    weight   fp16 ``[Cout, Cin, R, R]`` in channels-last memory format, i.e.
             stored ``[Cout][R][R][Cin]`` - the GEMM's K order
    Cin must be a multiple of 64 (every K step of 64 lies inside one tap) and
    Cout a multiple of 64.
    """
    import torch

    if x.dtype != torch.float16 or weight.dtype != torch.float16:
        raise TypeError("conv2d takes fp16 tensors")
    if not _is_nhwc(x):
        raise ValueError("x must be a channels-last [N, C, H, W] tensor")
    n, cin, h, w = x.shape
    cout, wcin, r, s = weight.shape
    if wcin != cin or r != s:
        raise ValueError("weight must be [Cout, Cin, R, R] with Cin matching x")
    if not weight.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("weight must be in channels-last memory format")
    oh, ow = (h + 2 * padding - r) // stride + 1, (w + 2 * padding - r) // stride + 1
    if out is None:
        out = torch.empty((n, cout, oh, ow), dtype=torch.float16, device=x.device,
                          memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, cout, oh, ow) or not _is_nhwc(out):
        raise ValueError("out must be a channels-last tensor of the output shape")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not _is_nhwc(residual)
                                 or residual.dtype != torch.float16):
        raise ValueError("residual must match the output (fp16, channels-last)")
    if bias is not None and (bias.numel() != cout or bias.dtype != torch.float16 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous fp16 vector of Cout elements")
    _launch(x, weight, bias, residual, out, n, h, w, r, stride, padding, cin, cout, relu, variant, staging, split_k,
            tuned)
    return out


def conv2d_pointwise(x, weight, bias, weight2, bias2=None, residual=None, relu=True, stride=1, padding=0, out=None,
                     next_weight=None, next_bias=None, tile_m=0, proj_input=None):
    """``act(conv1x1(relu(conv2d(x, weight) + bias), weight2) + bias2 (+ residual))`` in one
    kernel: a bottleneck block's 3x3 convolution and the 1x1 expansion that follows it.
    The middle activation stays in LDS as the second GEMM's operand - it is never written
    to memory nor read back (at batch 128 that is 2 x 51 MB per block of ResNet-50's first
    stage, all of it on layers that run at the memory roofline).

    weight   fp16 ``[Mid, Cin, R, R]`` channels-last, Mid = 64 or 128 (one tile holds every
             middle channel), Cin a multiple of 64
    weight2  fp16 ``[Cout, Mid]`` contiguous, Cout a multiple of 64

    With ``next_weight`` (fp16 ``[N3, Cout]``, N3 = 64 or 128, with Mid = 128 only 128; ``relu`` on) the
    kernel also computes ``relu(conv1x1(result, next_weight) + next_bias)`` - the first
    convolution of the NEXT bottleneck block - from the output chunks while they are still on
    the CU, and the call returns ``(result, next_result)``: that convolution's separate pass
    over the (memory-bound) block output disappears.  ``tile_m`` = 64 or 128 pins the pixels per
    tile (0: the tuned table, then a rule on the number of tiles).

    ``proj_input`` (Mid = 64, stride 1, no ``residual``; N3 = 64 if any): the block's INPUT, a
    channels-last fp16 ``[N, 64, OH, OW]`` tensor.  ``weight2`` is then ``[Cout, Mid + 64]`` - the
    expansion's and the projection shortcut's weights side by side - and ``bias2`` the sum of their
    biases: the shortcut is accumulated in the same fp32 accumulator (the second GEMM simply gets a
    longer K) instead of being computed by its own kernel, written, and read back as a residual.
    """
    import torch

    if x.dtype != torch.float16 or weight.dtype != torch.float16 or weight2.dtype != torch.float16:
        raise TypeError("conv2d_pointwise takes fp16 tensors")
    if not _is_nhwc(x):
        raise ValueError("x must be a channels-last [N, C, H, W] tensor")
    n, cin, h, w = x.shape
    mid, wcin, r, s = weight.shape
    if wcin != cin or r != s or not weight.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("weight must be a channels-last [Mid, Cin, R, R] tensor with Cin matching x")
    cout = weight2.shape[0]
    k2 = mid + (64 if proj_input is not None else 0)
    if weight2.numel() != cout * k2 or not weight2.is_contiguous():
        raise ValueError(f"weight2 must be a contiguous [{cout}, {k2}] matrix")
    oh, ow = (h + 2 * padding - r) // stride + 1, (w + 2 * padding - r) // stride + 1
    if out is None:
        out = torch.empty((n, cout, oh, ow), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, cout, oh, ow) or not _is_nhwc(out):
        raise ValueError("out must be a channels-last tensor of the output shape")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not _is_nhwc(residual)
                                 or residual.dtype != torch.float16):
        raise ValueError("residual must match the output (fp16, channels-last)")
    for b, c in ((bias, mid), (bias2, cout)):
        if b is not None and (b.numel() != c or b.dtype != torch.float16 or not b.is_contiguous()):
            raise ValueError("biases must be contiguous fp16 vectors of the matching channel count")
    if proj_input is not None:
        if (mid != 64 or stride != 1 or residual is not None or proj_input.dtype != torch.float16
                or tuple(proj_input.shape) != (n, 64, oh, ow) or not _is_nhwc(proj_input)):
            raise ValueError("proj_input needs Mid = 64, stride 1, no residual and a channels-last fp16 "
                             "[N, 64, OH, OW] tensor")
    out3 = None
    n3 = 0
    if next_weight is not None:
        n3 = next_weight.shape[0]
        if (next_weight.dtype != torch.float16 or next_weight.numel() != n3 * cout or not next_weight.is_contiguous()
                or n3 not in (64, 128) or (mid == 128 and n3 != 128) or not relu):
            raise ValueError("next_weight must be a contiguous fp16 [64 | 128, Cout] matrix (128 rows with Mid = 128; "
                             "relu on)")
        if next_bias is not None and (next_bias.numel() != n3 or next_bias.dtype != torch.float16
                                      or not next_bias.is_contiguous()):
            raise ValueError("next_bias must be a contiguous fp16 vector of N3 elements")
        out3 = torch.empty((n, n3, oh, ow), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    rc = lib().tca_conv_pointwise_launch(
        x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, weight2.data_ptr(),
        bias2.data_ptr() if bias2 is not None else None, residual.data_ptr() if residual is not None else None,
        out.data_ptr(), n, h, w, int(r), int(stride), int(padding), cin, mid, cout, 1 if relu else 0,
        next_weight.data_ptr() if next_weight is not None else None,
        next_bias.data_ptr() if next_bias is not None else None,
        out3.data_ptr() if out3 is not None else None, n3, int(tile_m),
        proj_input.data_ptr() if proj_input is not None else None, _stream())
    check(rc, "tca_conv_pointwise_launch")
    return out if next_weight is None else (out, out3)


def permute_mid_k(weight2):
    """``[Cout, 64 (+ 64)]`` expansion weights with the first 64 K columns reordered to the order in which
    the row-organised bottleneck kernel holds the middle activation - straight out of the 3x3 product's
    accumulator registers: position ``32 s + 8 q + j`` takes middle channel ``16 (2 s + j // 4) + 4 q + j % 4``.
    Columns past 64 (the projection shortcut's) keep their order."""
    import torch

    order = [16 * (2 * s + j // 4) + 4 * q + j % 4 for s in range(2) for q in range(4) for j in range(8)]
    index = torch.tensor(order + list(range(64, weight2.shape[1])), device=weight2.device)
    return weight2.index_select(1, index).contiguous()


def bottleneck_rows(x, weight, bias, weight2_permuted, bias2, residual, next_weight, next_bias=None, proj_input=None,
                    out=None):
    """``y = relu(conv1x1(relu(conv3x3(x) + bias), W2) (+ conv1x1(proj_input, Wd)) + bias2 (+ residual))`` and
    ``t = relu(conv1x1(y, next_weight) + next_bias)`` in one kernel whose wavefronts own pixel rows for all
    channels (``csrc/hip/bottleneck_rows.hip``): no workgroup barrier after the 3x3 part.  Shapes: 64 input and
    middle channels, 256 output channels, stride 1, padding 1, width <= 62, ``next_weight`` ``[64 | 128, 256]``
    (64 with ``proj_input``).  ``weight2_permuted`` comes from :func:`permute_mid_k`.  Returns ``(y, t)``."""
    import torch

    if not _is_nhwc(x) or x.dtype != torch.float16 or x.shape[1] != 64:
        raise ValueError("x must be a channels-last fp16 [N, 64, H, W] tensor")
    n, _, h, w = x.shape
    k2 = 64 + (64 if proj_input is not None else 0)
    n3 = next_weight.shape[0]
    if (tuple(weight.shape) != (64, 64, 3, 3) or not weight.is_contiguous(memory_format=torch.channels_last)
            or weight.dtype != torch.float16):
        raise ValueError("weight must be a channels-last fp16 [64, 64, 3, 3] tensor")
    if (tuple(weight2_permuted.shape) != (256, k2) or not weight2_permuted.is_contiguous()
            or weight2_permuted.dtype != torch.float16):
        raise ValueError(f"weight2_permuted must be a contiguous fp16 [256, {k2}] matrix")
    if (n3 not in (64, 128) or tuple(next_weight.shape) != (n3, 256) or not next_weight.is_contiguous()
            or next_weight.dtype != torch.float16 or (proj_input is not None and n3 != 64)):
        raise ValueError("next_weight must be a contiguous fp16 [64 | 128, 256] matrix (64 rows with proj_input)")
    for b, c in ((bias, 64), (bias2, 256), (next_bias, n3)):
        if b is not None and (b.numel() != c or b.dtype != torch.float16 or not b.is_contiguous()):
            raise ValueError("biases must be contiguous fp16 vectors of the matching channel count")
    if bias is None:
        raise ValueError("the 3x3 convolution's bias is required")
    if residual is not None and (proj_input is not None or tuple(residual.shape) != (n, 256, h, w)
                                 or not _is_nhwc(residual) or residual.dtype != torch.float16):
        raise ValueError("residual must be a channels-last fp16 [N, 256, H, W] tensor (and excludes proj_input)")
    if proj_input is not None and (tuple(proj_input.shape) != (n, 64, h, w) or not _is_nhwc(proj_input)
                                   or proj_input.dtype != torch.float16):
        raise ValueError("proj_input must be a channels-last fp16 [N, 64, H, W] tensor")
    if out is None:
        out = torch.empty((n, 256, h, w), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, 256, h, w) or not _is_nhwc(out):
        raise ValueError("out must be a channels-last tensor of the output shape")
    out3 = torch.empty((n, n3, h, w), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    rc = lib().tca_bottleneck_rows_launch(
        x.data_ptr(), weight.data_ptr(), bias.data_ptr(), weight2_permuted.data_ptr(),
        bias2.data_ptr() if bias2 is not None else None, residual.data_ptr() if residual is not None else None,
        out.data_ptr(), n, h, w, 256, next_weight.data_ptr(), next_bias.data_ptr() if next_bias is not None else None,
        out3.data_ptr(), n3, proj_input.data_ptr() if proj_input is not None else None, _stream())
    check(rc, "tca_bottleneck_rows_launch")
    return out, out3


def bias_act_(y, bias, residual=None, relu=False):
    """In place ``y = act(y + bias (+ residual))`` on a channels-last fp16 tensor."""
    import torch

    if y.dtype != torch.float16 or not _is_nhwc(y):
        raise ValueError("y must be a channels-last fp16 [N, C, H, W] tensor")
    n, c, h, w = y.shape
    if bias.numel() != c or bias.dtype != torch.float16 or not bias.is_contiguous():
        raise ValueError("bias must be a contiguous fp16 vector of C elements")
    if residual is not None and (tuple(residual.shape) != tuple(y.shape) or not _is_nhwc(residual)
                                 or residual.dtype != torch.float16):
        raise ValueError("residual must match y (fp16, channels-last)")
    rc = lib().tca_bias_act_launch(y.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                                   n * h * w, c, 1 if relu else 0, _stream())
    check(rc, "tca_bias_act_launch")
    return y


def bias_relu_maxpool(x, bias, out=None):
    """``max_pool2d(relu(x + bias), 3, stride=2, padding=1)`` in one pass over a
    channels-last fp16 tensor (bias and ReLU commute with the window maximum, so
    they are applied to the pooled values)."""
    import torch

    if x.dtype != torch.float16 or not _is_nhwc(x):
        raise ValueError("x must be a channels-last fp16 [N, C, H, W] tensor")
    n, c, h, w = x.shape
    if c % 8 != 0:
        raise ValueError("C must be a multiple of 8")
    if bias.numel() != c or bias.dtype != torch.float16 or not bias.is_contiguous():
        raise ValueError("bias must be a contiguous fp16 vector of C elements")
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        out = torch.empty((n, c, oh, ow), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, c, oh, ow) or out.dtype != torch.float16 or not _is_nhwc(out):
        raise ValueError("out must be a channels-last fp16 tensor of the pooled shape")
    rc = lib().tca_bias_relu_maxpool_launch(x.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w, c, _stream())
    check(rc, "tca_bias_relu_maxpool_launch")
    return out


def avgpool_linear(y, weight, bias=None, out=None):
    """``linear(mean(y, dim=(2, 3)), weight, bias)`` as fp32: the classifier tail.  ``y`` is a
    channels-last fp16 ``[N, C, H, W]`` tensor (C a multiple of 512), ``weight`` fp16 ``[Out, C]``
    contiguous, ``out`` an optional fp32 ``[N, Out]`` contiguous destination (e.g. a view of a
    client's shared-memory region)."""
    import torch

    if y.dtype != torch.float16 or not _is_nhwc(y):
        raise ValueError("y must be a channels-last fp16 [N, C, H, W] tensor")
    n, c, h, w = y.shape
    n_out = weight.shape[0]
    if weight.dtype != torch.float16 or weight.numel() != n_out * c or not weight.is_contiguous() or c % 512:
        raise ValueError("weight must be a contiguous fp16 [Out, C] matrix with C a multiple of 512")
    if bias is not None and (bias.numel() != n_out or bias.dtype != torch.float16 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous fp16 vector of Out elements")
    if out is None:
        out = torch.empty((n, n_out), dtype=torch.float32, device=y.device)
    elif tuple(out.shape) != (n, n_out) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 [N, Out] tensor")
    feats = torch.empty((n, c), dtype=torch.float16, device=y.device)
    rc = lib().tca_avgpool_launch(y.data_ptr(), feats.data_ptr(), n, h * w, c, _stream())
    check(rc, "tca_avgpool_launch")
    rc = lib().tca_linear_launch(feats.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 out.data_ptr(), n, n_out, c, _stream())
    check(rc, "tca_linear_launch")
    return out


def pack_stem_weight(weight):
    """``[64, 3, 7, 7]`` convolution weight -> the ``[64, 192]`` fp16 matrix the
    fused stem kernel reads: K ordered (c, kh, kw) with kw padded from 7 to 8 and
    the total padded from 168 to 192, zeros in the padding.  The kw padding sits in
    FRONT (slot kw + 1), which makes every 8-pixel run the kernel reads start at an
    even input column."""
    import torch

    if tuple(weight.shape) != (64, 3, 7, 7):
        raise ValueError("the fused stem expects a [64, 3, 7, 7] weight")
    packed = torch.zeros(64, 192, dtype=torch.float16, device=weight.device)
    padded = torch.zeros(64, 3, 7, 8, dtype=torch.float16, device=weight.device)
    padded[:, :, :, 1:] = weight.to(torch.float16)
    packed[:, :168] = padded.reshape(64, 168)
    return packed.contiguous()


def stem_conv_pool(x, packed_weight, bias, out=None):
    """``max_pool2d(relu(conv2d(x, w, stride=2, padding=3) + bias), 3, 2, 1)`` for the
    3-channel 7x7 network stem in one kernel.  ``x`` is a plain contiguous NCHW fp16
    tensor (what clients send); the result is channels-last ``[N, 64, PH, PW]``."""
    import torch

    if x.dtype != torch.float16 or x.dim() != 4 or x.shape[1] != 3 or not x.is_contiguous():
        raise ValueError("x must be a contiguous NCHW fp16 tensor with 3 channels")
    if tuple(packed_weight.shape) != (64,
                                      192) or packed_weight.dtype != torch.float16 or not packed_weight.is_contiguous():
        raise ValueError("packed_weight must come from pack_stem_weight()")
    if bias.numel() != 64 or bias.dtype != torch.float16 or not bias.is_contiguous():
        raise ValueError("bias must be a contiguous fp16 vector of 64 elements")
    n, _, h, w = x.shape
    if w % 2:
        raise ValueError("the fused stem reads rows as aligned dwords: the image width must be even")
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    if out is None:
        out = torch.empty((n, 64, ph, pw), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    elif tuple(out.shape) != (n, 64, ph, pw) or out.dtype != torch.float16 or not _is_nhwc(out):
        raise ValueError("out must be a channels-last fp16 tensor of the pooled shape")
    rc = lib().tca_stem_conv_pool_launch(x.data_ptr(), packed_weight.data_ptr(), bias.data_ptr(), out.data_ptr(), n,
                                         h, w,
                                         _stream())
    check(rc, "tca_stem_conv_pool_launch")
    return out
